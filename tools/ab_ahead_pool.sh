#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for t in 2 4 6; do
  XENOMAPPER_AHEAD_THREADS=$t timeout -k 10 240 python3 tools/bench_e2e.py --pairs 4000000 --out-dir /dev/shm 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('SAM -> files, ahead threads $t: %6.2f M pairs/s  %.3f s | stage %.3f emit %.3f (extend %.3f, fill %.3f)' % (d['value']/1e6, d['seconds'], p.get('stage',0), p.get('emit',0), p.get('emit_extend',0), p.get('emit_fill',0)))"
done
done
for rep in 1 2; do
for t in 2 6; do
  XENOMAPPER_AHEAD_THREADS=$t timeout -k 10 240 python3 tools/bench_bam.py --copies 48000 --files 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('BAM -> files, ahead threads $t: %6.2f M pairs/s  %.3f s | emit %.3f (extend %.3f, fill %.3f)' % (d['value']/1e6, d['seconds'], p.get('emit',0), p.get('emit_extend',0), p.get('emit_fill',0)))"
done
done
