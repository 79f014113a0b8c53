#!/bin/bash
# Runs ON the GPU box: the flat step (xm_classify_compact_dev, three launches) against the segmented-lists step
# (xm_classify_runs_dev, one launch; XM_BENCH_RUNS=1) on the same box, interleaved ROUNDS times per workload.
#   tools/ab_runs.sh [rounds] [workload ...]
ROUNDS=${1:-3}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=${*:-cfg2 cfg5 se}
for w in $WL; do
  echo "== $w"
  for r in $(seq 1 $ROUNDS); do
    for form in flat runs; do
      if [ $form = runs ]; then export XM_BENCH_RUNS=1; else unset XM_BENCH_RUNS; fi
      python3 "$ROOT/bench.py" --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-workloads 2>/dev/null |
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-5s ms_per_step %.4f median %.4f  kernels %s  B/unit %.2f  frac_by_ms_per_step %.3f ok=%s' % ('$form', d['ms_per_step'], d['ms_per_step_median'], d['kernel_ms'], d['roofline_step']['bytes_per_unit'], d['roofline_step']['frac_by_ms_per_step'], d['verified_vs_oracle']))"
    done
  done
done
unset XM_BENCH_RUNS
