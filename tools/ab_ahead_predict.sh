#!/bin/bash
# Runs ON the GPU box: the output files extended towards the size the run predicts, in 32 MB pieces (default), against twice the
# last call's bytes ahead (XENOMAPPER_AHEAD_PREDICT=0, the policy until round 6), alternating; SAM text and BAM in, six files on tmpfs out.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for v in 1 0; do
  XENOMAPPER_AHEAD_PREDICT=$v timeout -k 10 240 python3 tools/bench_e2e.py --pairs 4000000 --out-dir /dev/shm 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('SAM -> files, predict $v: %6.2f M pairs/s  %.3f s | stage %.3f strip %.3f wait_out %.3f emit %.3f (extend %.3f fill %.3f) close %.3f' % (d['value']/1e6, d['seconds'], p.get('stage',0), p.get('strip',0), p.get('sam_wait_out',0), p.get('emit',0), p.get('emit_extend',0), p.get('emit_fill',0), p.get('close',0)))"
done
done
for rep in 1 2; do
for v in 1 0; do
  XENOMAPPER_AHEAD_PREDICT=$v timeout -k 10 240 python3 tools/bench_bam.py --copies 48000 --files 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('BAM -> files, predict $v: %6.2f M pairs/s  %.3f s | strip %.3f wait_raw %.3f emit %.3f (extend %.3f fill %.3f) close %.3f' % (d['value']/1e6, d['seconds'], p.get('strip',0), p.get('bam_wait_raw',0), p.get('emit',0), p.get('emit_extend',0), p.get('emit_fill',0), p.get('close',0)))"
done
done
