#!/bin/bash
# ON the GPU box: the BAM path (tools/bench_bam.py --copies 48000) under rocprofv3 --kernel-trace for several environments,
# the timed pass's per-kernel means side by side (tools/trace_gaps.py).   tools/ab_bam_env_trace.sh "name:VAR=1 VAR2=3" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  OUT=$ROOT/gpurun_out/prof_env_$name
  rm -rf "$OUT"; mkdir -p "$OUT"
  ( for kv in $envs; do export "$kv"; done
    timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/bam" -o bam --output-format csv -- python3 "$ROOT/tools/bench_bam.py" --copies ${COPIES:-48000} > "$OUT/bench.json" 2> "$OUT/err.txt" ) || { echo "$name failed"; tail -3 "$OUT/err.txt"; continue; }
  echo "== $name ($envs): $(python3 -c "import json,sys; d=json.loads(open('$OUT/bench.json').readline()); print('%.2f M pairs/s' % (d['value']/1e6))")"
  T=$(find "$OUT/bam" -name "*kernel_trace.csv" | head -1)
  python3 "$ROOT/tools/trace_gaps.py" "$T" --second-pass --top 7 | sed -n 1,9p
  rm -f "$T"
done
