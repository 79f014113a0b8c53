#!/bin/bash
# Runs ON the GPU box: how far the output files are extended ahead of the writer (XENOMAPPER_AHEAD x the bytes of the last call), SAM
# text in -> six files on tmpfs, configurations in rotation.   tools/ab_ahead_threads.sh [factors...]
cd "$(dirname "$0")/.."
for rep in 1 2 3 4 5; do
for f in ${@:-0.5 1 1.5 2}; do
  XENOMAPPER_AHEAD=$f timeout -k 10 240 python3 tools/bench_e2e.py --pairs 4000000 --out-dir /dev/shm 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('SAM -> files, ahead x $f: %6.2f M pairs/s  %.3f s | stage %.3f emit %.3f (extend %.3f fill %.3f)' % (d['value']/1e6, d['seconds'], p.get('stage',0), p.get('emit',0), p.get('emit_extend',0), p.get('emit_fill',0)))"
done
done
