#!/bin/bash
# Runs ON the GPU box: bench.py once per tuning build under build/ab/ (tools/ab_build.sh), interleaved ROUNDS times,
# one short line per run.   tools/ab_run.sh "<bench args>" [rounds] [name ...]
ARGS=${1:---workload cfg3}; ROUNDS=${2:-2}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAMES=${*:-$(cd "$ROOT/build/ab" && ls *.so | sed 's/\.so$//')}
for r in $(seq 1 $ROUNDS); do
  for nm in $NAMES; do
    XENOMAPPER_HIP_LIB=$ROOT/build/ab/$nm.so python3 "$ROOT/bench.py" $ARGS --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-workloads 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s ms_per_step %.4f  classify %.4f  kernels %s  frac_step %.3f ok=%s' % ('$nm', d['ms_per_step'], d['roofline']['kernel_ms'], d['kernel_ms'], d['roofline_step']['frac_by_ms_per_step'], d['verified_vs_oracle']))"
  done
done
