#!/bin/bash
# Runs ON the GPU box: what the far window reads cost the inflate launch with the wide token loop.  Two
# builds under build/ab (tools/ab_build.sh base: nofar:-DXMI_TIMING_NO_FAR_READS): "nofar" takes EVERY source byte from the
# 1 KiB ring -- wrong bytes, same control flow, timing only (its `verified` is false by construction).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for r in 1 2 3; do
  for nm in base nofar; do
    XENOMAPPER_HIP_LIB=$ROOT/build/ab/$nm.so python3 tools/bench_inflate.py --out-gb 1.0 --reps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-6s inflate %.2f ms = %.1f GB/s  (all: %s)  crc %.2f ms  verified %s' % ('$nm', d['ms'], d['value'], d['ms_all'], d['crc_ms'], d['verified']))"
  done
done
