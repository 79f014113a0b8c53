#!/bin/bash
# Runs ON the GPU box: BAM in -> six outputs with the records' SAM text printed on the device (default) against printed by the host
# threads from the packed records (XENOMAPPER_GPU_BAM_TEXT=0), three rounds in rotation.   tools/ab_bam_text.sh [copies] [bench_bam flags]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
C=${1:-48000}
shift
for round in 1 2 3; do
  for t in 1 0; do
    echo -n "round $round text on device $t: "
    XENOMAPPER_GPU_BAM_TEXT=$t timeout -k 10 300 python3 tools/bench_bam.py --copies $C "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['phases']
print('%.2f M/s, %.3f s; window %.3f strip %.3f classify %.3f | parse %.3f (wait %.3f print %.3f) emit %.3f' % (d['value'] / 1e6, d['seconds'], p.get('window', 0), p.get('strip', 0), p.get('classify', 0), p.get('parse', 0), p.get('bam_wait_raw', 0), p.get('bam_print', 0), p.get('emit', 0)))" || exit 1
  done
done
