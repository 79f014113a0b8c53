#!/bin/bash
# Runs ON the GPU box: SQ instruction / activity counters of the xm:: kernels for one workload (two --pmc passes).
#   tools/collect_sq.sh r02b cfg2
set -u
TAG=${1:?tag}; W=${2:-cfg2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify --no-extra-workloads"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/sq1_$W" -o "$W" --output-format csv -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/sq1_$W.err" || { echo "sq1 failed"; tail -5 "$OUT/sq1_$W.err"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d "$OUT/sq2_$W" -o "$W" --output-format csv -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/sq2_$W.err" || { echo "sq2 failed"; tail -5 "$OUT/sq2_$W.err"; exit 1; }
python3 "$ROOT/tools/pmc_table.py" "$OUT/sq1_$W/${W}_counter_collection.csv" "$OUT/sq2_$W/${W}_counter_collection.csv"
