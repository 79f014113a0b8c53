#!/bin/bash
# Runs ON the GPU box: SAM text in -> six outputs with the workgroups (x 4 one-wave groups) of the kernel that sends the gathered
# outputs home swept (XM_BAMDEV_COPY_WG, shared with the BAM path), to /dev/null and to files.   tools/ab_sam_copy_wg.sh [pairs]
PAIRS=${1:-4000000}
cd "$(dirname "$0")/.."
for EXTRA in "" "--out-dir /dev/shm"; do
  echo "== outputs: ${EXTRA:-/dev/null}"
  for rep in 1 2; do
  for v in 1 2 4 8 16; do
    XM_BAMDEV_COPY_WG=$v timeout -k 10 240 python3 tools/bench_e2e.py --pairs $PAIRS $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('copy workgroups %2d x 4: %6.2f M pairs/s  %.3f s | stage %.3f upload_ms %.0f strip %.3f wait_out %.3f emit %.3f' % ($v, d['value']/1e6, d['seconds'], p.get('stage',0), p.get('strip_upload_ms',0), p.get('strip',0), p.get('sam_wait_out',0), p.get('emit',0)))" || echo "$v failed"
  done
  done
done
