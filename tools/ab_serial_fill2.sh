#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for cfg in "1 8" "0 8" "1 10" "0 10" "0 12"; do
  set -- $cfg
  XM_BAMDEV_SERIAL_FILL=$1 XM_BAMDEV_COPY_WAVES=$2 python3 tools/bench_bam.py --copies 48000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('serial fill $1, copy waves $2: %.2f M pairs/s %.3f s strip %.3f wait_raw %.3f' % (d['value']/1e6, d['seconds'], p.get('strip',0), p.get('bam_wait_raw',0)))"
done
done
