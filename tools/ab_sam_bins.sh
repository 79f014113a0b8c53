#!/bin/bash
# Runs ON the GPU box: SAM text in -> six outputs, the outputs gathered on the device (xm_strip_fetch_bins) against the host writer.
#   tools/ab_sam_bins.sh [pairs] > gpurun_out/r6/ab_sam_bins.txt
PAIRS=${1:-4000000}
cd "$(dirname "$0")/.."
one() {
  local label=$1; shift
  for rep in 1 2; do
    env "$@" timeout -k 10 240 python3 tools/bench_e2e.py --pairs $PAIRS $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['phases']
print('%-22s %6.2f M pairs/s  %.3f s | stage %.3f upload_ms %.0f (%.1f GB/s) strip %.3f classify %.3f wait_out %.3f emit %.3f (extend %.3f fill %.3f) write %.3f bins %s/%s' % ('$label', d['value']/1e6, d['seconds'], p.get('stage',0), p.get('strip_upload_ms',0), (d['input_bytes']/1e6/max(p.get('strip_upload_ms',1),1e-9)), p.get('strip',0), p.get('classify',0), p.get('sam_wait_out',0), p.get('emit',0), p.get('emit_extend',0), p.get('emit_fill',0), p.get('write',0), p.get('sam_windows_device_bins',0), p.get('sam_windows',0)))" || echo "$label failed"
  done
}
for EXTRA in "" "--out-dir /dev/shm"; do
  echo "== outputs: ${EXTRA:-/dev/null}"
  one "host gathers" XENOMAPPER_GPU_SAM_BINS=0 XM_STRIP_ZEROCOPY=0
  one "device bins" XENOMAPPER_GPU_SAM_BINS=1 XM_STRIP_ZEROCOPY=0
  one "bins + zero-copy S1" XENOMAPPER_GPU_SAM_BINS=1 XM_STRIP_ZEROCOPY=1
  one "bins + zc, 256 MB" XENOMAPPER_GPU_SAM_BINS=1 XM_STRIP_ZEROCOPY=1 XENOMAPPER_WINDOW_MB=256
  one "bins + zc, 64 MB" XENOMAPPER_GPU_SAM_BINS=1 XM_STRIP_ZEROCOPY=1 XENOMAPPER_WINDOW_MB=64



done
