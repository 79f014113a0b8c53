#!/bin/bash
# Runs ON the GPU box: the writer's one long mapping per output file (XENOMAPPER_MAP_AHEAD_MB=1024, default) against a mapping per
# bin and window (0: as until round 6), alternating; BAM and SAM text in, six files on tmpfs out.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for v in 1024 0; do
  XENOMAPPER_MAP_AHEAD_MB=$v timeout -k 10 240 python3 tools/bench_bam.py --copies 48000 --files 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('BAM -> files, mapping ahead $v MB: %6.2f M pairs/s  %.3f s | emit %.3f (extend %.3f of which mmap %.3f, fill %.3f) close %.3f' % (d['value']/1e6, d['seconds'], p.get('emit',0), p.get('emit_extend',0), p.get('emit_map',0), p.get('emit_fill',0), p.get('close',0)))"
done
done
for rep in 1 2 3; do
for v in 1024 0; do
  XENOMAPPER_MAP_AHEAD_MB=$v timeout -k 10 240 python3 tools/bench_e2e.py --pairs 4000000 --out-dir /dev/shm 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('SAM -> files, mapping ahead $v MB: %6.2f M pairs/s  %.3f s | stage %.3f emit %.3f (extend %.3f of which mmap %.3f, fill %.3f) close %.3f' % (d['value']/1e6, d['seconds'], p.get('stage',0), p.get('emit',0), p.get('emit_extend',0), p.get('emit_map',0), p.get('emit_fill',0), p.get('close',0)))"
done
done
