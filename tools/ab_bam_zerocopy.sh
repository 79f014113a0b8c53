#!/bin/bash
# Runs ON the GPU box: BAM in -> six outputs with the inflate launch reading the compressed blocks in the host's page-locked staging buffers
# (the default: no upload) against a copy in HBM (XM_BAMDEV_ZEROCOPY=0), three rounds in rotation.   tools/ab_bam_zerocopy.sh [copies]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
C=${1:-48000}
for round in 1 2 3; do
  for z in 0 1; do
    echo -n "round $round zero copy $z: "
    export XM_BAMDEV_ZEROCOPY=$z
    timeout -k 10 300 python3 tools/bench_bam.py --copies $C 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['phases']
print('%.2f M/s, %.3f s; window %.3f strip %.3f (device: inflate+crc+h2d %.0f ms, record kernels %.0f ms) classify %.3f | wait %.3f emit %.3f' % (d['value'] / 1e6, d['seconds'], p.get('window', 0), p.get('strip', 0), p.get('strip_upload_ms', 0), p.get('strip_kernels_ms', 0), p.get('classify', 0), p.get('bam_wait_raw', 0), p.get('emit', 0)))" || exit 1
  done
done
