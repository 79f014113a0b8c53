#!/bin/bash
# Runs ON the GPU box: the output copy's waves (XM_BAMDEV_COPY_WAVES) x pieces of 16 bytes per lane in flight (XM_BAMDEV_COPY_DEEP: 4 or 8),
# BAM in -> six outputs on /dev/null, configurations in rotation.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for cfg in "8 4" "8 8" "6 8" "4 8" "6 4" "10 4"; do
  set -- $cfg
  XM_BAMDEV_COPY_WAVES=$1 XM_BAMDEV_COPY_DEEP=$2 python3 tools/bench_bam.py --copies 48000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('copy waves $1, pieces in flight $2: %.2f M pairs/s %.3f s strip %.3f wait_raw %.3f' % (d['value']/1e6, d['seconds'], p.get('strip',0), p.get('bam_wait_raw',0)))"
done
done
