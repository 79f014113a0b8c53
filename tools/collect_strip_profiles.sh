#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 kernel stats of the GPU stripper's kernels for one round tag -- the file path
# on a paired (plain walk) and a single-end (skipping walk) pair of SAM files, 2 M units each.
#   tools/collect_strip_profiles.sh r04        -> gpurun_out/prof_<tag>/<tag>_strip_{pe,se}_kernel_stats.csv  (copy into profiles/)
set -u
TAG=${1:?round tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for M in liberal:pe se:se; do
  MODE=${M%%:*}; NAME=${M#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/strip_$NAME" -o strip --output-format csv -- python3 "$ROOT/tools/bench_e2e.py" --pairs 2000000 --mode "$MODE" \
      > "$OUT/bench_e2e_$NAME.json" 2> "$OUT/strip_$NAME.err" || { echo "strip $NAME failed"; tail -5 "$OUT/strip_$NAME.err"; exit 1; }
  F=$(find "$OUT/strip_$NAME" -name "*kernel_stats.csv" | head -1)
  cp "$F" "$OUT/${TAG}_strip_${NAME}_kernel_stats.csv"
  echo "== $NAME"; grep -i "anonymous\|copyBuffer" "$OUT/${TAG}_strip_${NAME}_kernel_stats.csv" | cut -c1-120
done
echo done
