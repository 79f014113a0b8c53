#!/bin/bash
# Runs ON the GPU box: instruction counters of inflate kernel builds under build/ab/<name>/ (tools/ab_inflate.sh builds them) on the
# 1 GB check input, one rocprofv3 --pmc pass per set and build (the program itself follows `--`).   tools/pmc_inflate_variants.sh serial wide2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_variants
mkdir -p "$OUT"
BAM=/dev/shm/xm_pmc_1g.bam
python3 -c "
import sys; sys.path.insert(0, '$ROOT/tools'); import bench_bam
bench_bam.tiled_bam('$ROOT/tests/golden/ref_data/paired_end_testdata_human.bam', '$BAM', 8400)" || exit 1
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  export LD_LIBRARY_PATH=$ROOT/build/ab/$n
  N=1
  for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    timeout -k 10 200 rocprofv3 --pmc $SET -d "$OUT/$n$N" -o inflate --output-format csv -- "$ROOT/build/inflate_gpu_check" --reps 1 $BAM > /dev/null 2> "$OUT/$n$N.err" || { echo "pmc $n $N failed"; tail -5 "$OUT/$n$N.err"; rm -f $BAM; exit 1; }
    N=$((N + 1))
  done
done
rm -f $BAM
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
csv.field_size_limit(1 << 30)
for n in sys.argv[2:]:
    acc = defaultdict(list)
    for path in glob.glob(out + "/" + n + "[12]/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "inflate_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(n, {k: "%.3g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
