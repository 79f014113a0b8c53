#!/bin/bash
# Runs ON the GPU box: the workgroups (x 4 one-wave groups) of the kernel that sends the gathered outputs home (XM_BAMDEV_COPY_WG),
# swept twice in rotation on one box.   tools/ab_copy_wg.sh [values...]
cd "$(dirname "$0")/.."
VALS=${@:-1 2 4 8 16}
for round in 1 2; do
for v in $VALS; do
  XM_BAMDEV_COPY_WG=$v python3 tools/bench_bam.py --copies 48000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('copy workgroups $v x 4: %.2f M pairs/s %.3f s strip %.3f wait_raw %.3f' % (d['value']/1e6, d['seconds'], p.get('strip',0), p.get('bam_wait_raw',0)))"
done
done
