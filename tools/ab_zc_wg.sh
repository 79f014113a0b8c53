#!/bin/bash
cd "$(dirname "$0")/.."
for round in 1 2; do
for cfg in "1 2" "0 2" "0 4" "0 8" "0 16"; do
  set -- $cfg
  XM_BAMDEV_ZEROCOPY=$1 XM_BAMDEV_COPY_WG=$2 python3 tools/bench_bam.py --copies 48000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('zero copy $1, copy workgroups $2 x 4: %.2f M pairs/s %.3f s strip %.3f wait_raw %.3f' % (d['value']/1e6, d['seconds'], p.get('strip',0), p.get('bam_wait_raw',0)))"
done
done
