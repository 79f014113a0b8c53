#!/bin/bash
# Runs ON the GPU box: a fingerprint of this box for the K1p question (why classify_cigp_kernel moves +-7 % box to box):
# the kernel's mean duration and the streaming probe from a plain bench run, then SQ / GRBM counters of the same kernel, then the clocks.
# One block of text per call; calls on different boxes are put side by side in profiles/rNN_k1p_boxes.txt.
#   tools/k1p_box.sh r05
set -u
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_k1p_$(date +%H%M%S)
mkdir -p "$OUT"
ARGS="--workload cfg3 --no-cpu-baseline --no-e2e --no-verify --no-extra-workloads"
echo "== box $(hostname) $(date -u +%FT%TZ)"
(rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4) || true
python3 "$ROOT/bench.py" $ARGS --steps 30 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -3 "$OUT/bench.err"; exit 1; }
python3 - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("plain run: K1p %.1f us (frac %.3f), probe ceiling %.0f GB/s, memcpy d2d %.0f GB/s, ms_per_step %.4f, kernels %s" % (
    1e3 * r["kernel_ms"], r["frac"], r.get("copy_ceiling_GBps") or 0, r.get("memcpy_d2d_GBps") or 0, d["ms_per_step"], d["kernel_ms"]))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d "$OUT/sq" -o k1p --output-format csv -- python3 "$ROOT/bench.py" $ARGS --steps 5 --warmup 2 > /dev/null 2> "$OUT/sq.err" || { echo "pmc failed"; tail -3 "$OUT/sq.err"; exit 1; }
python3 - "$OUT/sq" <<'PY'
import csv, glob, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
acc = defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "classify_cigp_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
dur = []
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "classify_cigp_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if dur and "GRBM_GUI_ACTIVE" in m:
    dm = sum(dur) / len(dur)
    print("under --pmc (launches apart): K1p %.1f us, shader clock = GRBM_GUI_ACTIVE / 8 / duration = %.3f GHz" % (dm / 1e3, m["GRBM_GUI_ACTIVE"] / 8 / dm))
print("K1p counters per launch: " + "  ".join("%s %.4g" % (k, m[k]) for k in sorted(m)))
if "GRBM_GUI_ACTIVE" in m and "SQ_WAVE_CYCLES" in m:
    print("  GRBM_GUI_ACTIVE / 8 XCDs = %.0f cycles; wave quad-cycles per wave %.0f; waiting %.2f of wave time" % (
        m["GRBM_GUI_ACTIVE"] / 8, m["SQ_WAVE_CYCLES"] / max(m.get("SQ_WAVES", 1), 1), m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]))
PY
