#!/bin/bash
# Runs ON the GPU box: the reader pool of the SAM path's staging reads (XENOMAPPER_PREAD_THREADS; 0 = the slot's own 16 threads),
# three rounds in rotation, to /dev/null and to files on tmpfs.   tools/ab_pread_threads.sh [thread counts ...]
cd "$(dirname "$0")/.."
for EXTRA in "" "--out-dir /dev/shm"; do
  echo "== outputs: ${EXTRA:-/dev/null}"
  for rep in 1 2 3; do
  for t in ${@:-0 8 12 24 32}; do
    XENOMAPPER_PREAD_THREADS=$t timeout -k 10 120 python3 tools/bench_e2e.py --pairs 4000000 $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('reader threads %2d: %.2f M pairs/s %.3f s | stage %.3f strip %.3f wait_out %.3f emit %.3f' % ($t, d['value']/1e6, d['seconds'], p.get('stage',0), p.get('strip',0), p.get('sam_wait_out',0), p.get('emit',0)))"
  done
  done
done
