#!/bin/bash
# Runs ON the GPU box: tools/probe_queue_collision.py for 0..9 streams made before the front end's own, with the copy stream tried
# against the compute streams (default) and taken untested (XM_COPY_STREAM_PROBE=0).
cd "$(dirname "$0")/.."
for env in ${ENVS:-"XM_COPY_STREAM_PROBE=1" "XM_COPY_STREAM_PROBE=0"}; do
  echo "== $env"
  for n in ${@:-0 1 2 3 4 5 6 7 8 9}; do
    env $env python3 tools/probe_queue_collision.py $n 2>/dev/null | tail -1
  done
done
