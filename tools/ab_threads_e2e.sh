#!/bin/bash
# Same-box A/B of the SAM file path's thread split (ON the GPU box): writer / parser threads (XENOMAPPER_THREADS) against a reader pool
# of its own for the staging preads (XENOMAPPER_PREAD_THREADS), 4 M pairs, outputs on /dev/null and on tmpfs files, two rounds.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for round in 1 2; do
  for combo in "0 0" "8 8" "10 6" "12 8" "16 8" "8 0"; do
    set -- $combo
    for out in devnull files; do
      if [ $out = files ]; then EXTRA="--out-dir /dev/shm/xm_ab_out"; mkdir -p /dev/shm/xm_ab_out; else EXTRA=""; fi
      echo -n "round $round threads $1 readers $2 $out: "
      XENOMAPPER_THREADS=$1 XENOMAPPER_PREAD_THREADS=$2 timeout -k 10 120 python3 tools/bench_e2e.py --pairs 4000000 $EXTRA 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
ph=d['phases']
up=d['input_bytes']/1e9/ph['stage'] if ph.get('stage') else 0
print('%.2f M pairs/s  upload %.1f GB/s  phases %s' % (d['value']/1e6, up, {k:ph[k] for k in ('stage','emit','total') if k in ph}))"
      rm -rf /dev/shm/xm_ab_out
    done
  done
done
