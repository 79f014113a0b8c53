#!/bin/bash
# ON the GPU box: rocprofv3 kernel stats of the GPU BAM path (tools/bench_bam.py --copies 16000) for build/ab/<name>.so builds given as arguments.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  OUT=$ROOT/gpurun_out/prof_ab_bam_$n
  mkdir -p "$OUT"
  export XENOMAPPER_HIP_LIB=$ROOT/build/ab/$n.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/bam" -o bam --output-format csv -- python3 "$ROOT/tools/bench_bam.py" --copies 16000 > "$OUT/bench.json" 2> "$OUT/err.txt" || { echo "$n failed"; tail -3 "$OUT/err.txt"; continue; }
  echo "== $n"
  python3 - "$(find "$OUT/bam" -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'(?:\(anonymous namespace\)::|xm::)(\w+)(<[^>]*>)?', r['Name'])
    if m and float(r['AverageNs']) > 2e5: print("  %-28s calls %3s avg %9.1f us" % ((m.group(1) + (m.group(2) or ''))[:28], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
