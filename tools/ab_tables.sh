#!/bin/bash
# Runs ON the GPU box: the BAM front end's block / walk / segment tables read in place by the kernels (default) against copied up first
# (XM_BAMDEV_ZEROCOPY_TABLES=0), alternating on one box.
cd "$(dirname "$0")/.."
for v in 1 0 1 0 1 0; do
  XM_BAMDEV_ZEROCOPY_TABLES=$v python3 tools/bench_bam.py --copies 48000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('tables read in place = $v: %.2f M pairs/s %.3f s strip %.3f wait_raw %.3f' % (d['value']/1e6, d['seconds'], p.get('strip',0), p.get('bam_wait_raw',0)))"
done
