#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for f in 1 2 0.5; do
  XENOMAPPER_AHEAD=$f timeout -k 10 240 python3 tools/bench_bam.py --copies 48000 --files 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('BAM -> files, ahead x $f: %6.2f M pairs/s  %.3f s | emit %.3f (extend %.3f fill %.3f) close %.3f' % (d['value']/1e6, d['seconds'], p.get('emit',0), p.get('emit_extend',0), p.get('emit_fill',0), p.get('close',0)))"
done
done
