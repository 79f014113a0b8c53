#!/bin/bash
# Runs ON the GPU box: the three-launch step against the single-kernel experiment (tuning builds with -DXM_ONEPASS=1 under
# build/ab/), interleaved.   tools/ab_onepass.sh "<bench args>" rounds name [name ...]
ARGS=${1:---workload cfg2}; ROUNDS=${2:-2}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {  # label place-mode lib
  XM_BENCH_PLACE=$2 XENOMAPPER_HIP_LIB=$3 python3 "$ROOT/bench.py" $ARGS --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-workloads 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s ms_per_step %.4f median %.4f  kernels %s  ok=%s' % ('$1', d['ms_per_step'], d['ms_per_step_median'], d['kernel_ms'], d['verified_vs_oracle']))"
}
for r in $(seq 1 $ROUNDS); do
  run three-launch 0 "$ROOT/xenomapper_amd/libxenomapper_hip.so"
  for nm in "$@"; do run "$nm" 2 "$ROOT/build/ab/$nm.so"; done
done
