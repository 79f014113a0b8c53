#!/bin/bash
# Runs ON the GPU box: the single-end step (K1 + K2b + staged K2c) of the round-4 tree (build/r04tree: `mkdir -p build/r04tree && git archive 1631e56 | tar -x -C build/r04tree && (cd build/r04tree && python -m xenomapper_amd.build)`, built in
# place) against the current tree, alternating on one box -- whether anything is left of round 5's K2c regression.
cd "$(dirname "$0")/.."
ARGS="--workload se --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-verify --no-extra-workloads"
for i in 1 2 3; do
  for tree in build/r04tree .; do
    ( cd $tree && python3 bench.py $ARGS 2>/dev/null | tail -1 > /tmp/ab_line.json; python3 - "$tree" <<'PY'
import json, sys, os
tree = sys.argv[1]
line = json.load(open("/tmp/ab_line.json"))
full = {}
p = os.path.join("gpurun_out", "bench_full_1gpu_se.json")
if os.path.exists(p):
    full = json.load(open(p))
print("%-14s ms_per_step %.4f median %.4f kernels %s" % ("r04" if "r04" in tree else "current", line["ms_per_step"], line.get("ms_per_step_median", 0), full.get("kernel_ms")))
PY
    )
  done
done
