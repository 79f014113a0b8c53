#!/bin/bash
# Runs ON the GPU box: the next window's inflate launch behind the printer of the window in front (default) or beside it
# (XM_BAMDEV_SERIAL_FILL=0), alternating on one box.
cd "$(dirname "$0")/.."
for v in 1 0 1 0 1 0; do
  XM_BAMDEV_SERIAL_FILL=$v python3 tools/bench_bam.py --copies 48000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('inflate behind the printer = $v: %.2f M pairs/s %.3f s strip %.3f wait_raw %.3f' % (d['value']/1e6, d['seconds'], p.get('strip',0), p.get('bam_wait_raw',0)))"
done
