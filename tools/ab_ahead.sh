#!/bin/bash
# Runs ON the GPU box: the read-ahead behind the run's quarter and half windows sized for the window that follows (twice as
# large; default) against sized like the window just staged (XENOMAPPER_BAM_AHEAD_GROW=0: stage() then indexes and reads
# windows 1 and 2 itself), alternating on one box.
cd "$(dirname "$0")/.."
for v in 1 0 1 0 1 0; do
  XENOMAPPER_BAM_AHEAD_GROW=$v python3 tools/bench_bam.py --copies 48000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('ahead sized for the next window = $v: %.2f M pairs/s %.3f s window %.3f strip %.3f wait_raw %.3f hits %d misses %d' % (d['value']/1e6, d['seconds'], p.get('window',0), p.get('strip',0), p.get('bam_wait_raw',0), p.get('bam_ahead_hits',0), p.get('bam_ahead_misses',0)))"
done
