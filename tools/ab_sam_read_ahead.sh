#!/bin/bash
# Runs ON the GPU box: SAM text in -> six outputs with each window's bytes read while the window in front is stripped
# (XENOMAPPER_SAM_READ_AHEAD=1, xm_strip_begin_behind / xm_strip_set_lead) against read when that one has said where it stopped (0),
# alternating; to /dev/null and to files on tmpfs.   tools/ab_sam_read_ahead.sh [reader threads ...]
cd "$(dirname "$0")/.."
for EXTRA in "" "--out-dir /dev/shm"; do
  echo "== outputs: ${EXTRA:-/dev/null}"
  for rep in 1 2 3; do
  for v in 1 0; do
    XENOMAPPER_SAM_READ_AHEAD=$v timeout -k 10 120 python3 tools/bench_e2e.py --pairs 4000000 $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('read ahead $v: %.2f M pairs/s %.3f s | window %.3f stage %.3f strip %.3f classify %.3f wait_out %.3f emit %.3f other %.3f; read-ahead windows %d of %d' % (d['value']/1e6, d['seconds'], p.get('window',0), p.get('stage',0), p.get('strip',0), p.get('classify',0), p.get('sam_wait_out',0), p.get('emit',0), p.get('other',0), p.get('sam_windows_read_ahead',0), p.get('sam_windows',0)))"
  done
  done
done
