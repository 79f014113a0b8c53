#!/bin/bash
# Runs ON the GPU box: are host <-> device copies shader kernels whatever the environment asks for?  rocprofv3 kernel stats of the BAM bench under
# HSA_ENABLE_SDMA / GPU_FORCE_BLIT_COPY_SIZE settings (the program itself follows `--`; `env` only sets variables for rocprofv3, before anything touches the GPU).
cd /tmp && export TMPDIR=/tmp
for v in "" "HSA_ENABLE_SDMA=1" "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0"; do
  rm -rf /tmp/sdma_prof
  env $v timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/sdma_prof -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_bam.py --copies 12000 > /tmp/sdma_out.txt 2>&1
  F=$(find /tmp/sdma_prof -name "*kernel_stats.csv" | head -1)
  echo "[$v] $(grep -c . $F) kernels; copyBuffer: $(grep rocclr_copyBuffer $F | cut -d, -f2,3 | head -2 | tr '\n' ' ')  value: $(grep -o '"value": [0-9.]*' /tmp/sdma_out.txt | head -1)"
done
