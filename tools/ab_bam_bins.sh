#!/bin/bash
# Runs ON the GPU box: BAM in -> six outputs with the outputs gathered on the device (xm_bamdev_fetch_bins) against round 5's form
# (text printed on the device, lines gathered by the host), and the copy kernel's workgroup count (0 = the runtime's blit).
#   tools/ab_bam_bins.sh [copies] > gpurun_out/r6/ab_bam_bins.txt
COPIES=${1:-48000}
cd "$(dirname "$0")/.."
one() {   # label, env...
  local label=$1; shift
  for rep in 1 2; do
    env "$@" timeout -k 10 240 python3 tools/bench_bam.py --copies $COPIES $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('%-26s %6.2f M pairs/s  %.3f s | strip %.3f inflate_ms %.0f wait_raw %.3f emit %.3f write %.3f bins %s' % ('$label', d['value']/1e6, d['seconds'], p.get('strip',0), p.get('strip_upload_ms',0), p.get('bam_wait_raw',0), p.get('emit',0), p.get('write',0), p.get('bam_windows_device_bins',0)))" || echo "$label failed"
  done
}
for EXTRA in "" "--files"; do
  echo "== outputs: ${EXTRA:-/dev/null}"
  one "host gathers (r5)" XENOMAPPER_GPU_BAM_BINS=0
  one "device bins, wg 64" XENOMAPPER_GPU_BAM_BINS=1 XM_BAMDEV_COPY_WG=64
  one "device bins, wg 16" XENOMAPPER_GPU_BAM_BINS=1 XM_BAMDEV_COPY_WG=16
  one "device bins, wg 8" XENOMAPPER_GPU_BAM_BINS=1 XM_BAMDEV_COPY_WG=8
  one "device bins, wg 4" XENOMAPPER_GPU_BAM_BINS=1 XM_BAMDEV_COPY_WG=4
  one "device bins, wg 2" XENOMAPPER_GPU_BAM_BINS=1 XM_BAMDEV_COPY_WG=2
  one "device bins, wg 32" XENOMAPPER_GPU_BAM_BINS=1 XM_BAMDEV_COPY_WG=32
  one "device bins, blit" XENOMAPPER_GPU_BAM_BINS=1 XM_BAMDEV_COPY_WG=0
done
