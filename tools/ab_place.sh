#!/bin/bash
# Runs ON the GPU box: same-box A/B of the three-launch step (xm_classify_compact*_dev) against the single-pass kernel
# (xm_classify_place*_dev, XM_BENCH_PLACE=1), interleaved ROUNDS times.   tools/ab_place.sh "<bench args>" [rounds]
ARGS=${1:---workload cfg2}; ROUNDS=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in $(seq 1 $ROUNDS); do
  for v in 0 1; do
    XM_BENCH_PLACE=$v python3 "$ROOT/bench.py" $ARGS --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-workloads 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s ms_per_step %.4f  classify %.4f  kernels %s  ok=%s' % ('place' if $v else 'three-launch', d['ms_per_step'], d['roofline']['kernel_ms'], d['kernel_ms'], d['verified_vs_oracle']))"
  done
done
