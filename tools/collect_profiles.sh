#!/bin/bash
# Runs ON the GPU box (through gpurun): bench lines + rocprofv3 kernel stats + PMC traffic passes for one round tag.
#   tools/collect_profiles.sh r02a [workload ...]      (default workloads: cfg2 cfg3)
# Output under gpurun_out/prof_<tag>/; condense afterwards with tools/summarize_prof.py into profiles/.
# rocprofv3: --pmc runs are separate from the --kernel-trace --stats run, the program itself follows `--`.
set -u
TAG=${1:?round tag}
shift
WORKLOADS=${*:-cfg2 cfg3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
  # "runs": the segmented-lists form of configs[1] (xm_classify_runs_dev, one launch) -- bench.py --workload cfg2 with XM_BENCH_RUNS=1
  if [ "$W" = runs ]; then export XM_BENCH_RUNS=1; BW=cfg2; else unset XM_BENCH_RUNS; BW=$W; fi
  ARGS="--workload $BW --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-verify --no-extra-workloads"
  echo "== $W: kernel trace"; date
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$W" -o "$W" --output-format csv -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace_$W.json" 2> "$OUT/trace_$W.err" || { echo "trace $W failed"; tail -5 "$OUT/trace_$W.err"; exit 1; }
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== $W: pmc $C"; date
    timeout -k 10 300 rocprofv3 --pmc $C -d "$OUT/pmc_${C}_$W" -o "$W" --output-format csv -- python3 "$ROOT/bench.py" $ARGS --steps 5 > /dev/null 2> "$OUT/pmc_${C}_$W.err" || { echo "pmc $C $W failed"; tail -5 "$OUT/pmc_${C}_$W.err"; exit 1; }
  done
done
unset XM_BENCH_RUNS
# condensed summaries (what gets copied into profiles/): kernel stats + per-launch HBM bytes, and the traffic entry bench.py replays
for W in $WORKLOADS; do
  python3 "$ROOT/tools/summarize_prof.py" --stats "$OUT/stats_$W/${W}_kernel_stats.csv" \
      --fetch "$OUT/pmc_FETCH_SIZE_$W/${W}_counter_collection.csv" --write "$OUT/pmc_WRITE_SIZE_$W/${W}_counter_collection.csv" \
      --out "$OUT/${TAG}_$W" $( [ "$W" = runs ] || echo --traffic-json "$OUT/pmc_traffic.json" --workload "$W" ) > "$OUT/summary_$W.txt" 2>&1 || { echo "summarize $W failed"; tail -5 "$OUT/summary_$W.txt"; }
  head -12 "$OUT/summary_$W.txt"
done
echo done
