#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 kernel stats of the GPU BAM path (xm_bamdev: inflate, CRC, record walk, strip,
# pair, fused pass) for one round tag -- the file path on the tiled BAM fixtures (tools/bench_bam.py, 48 000 copies = 11.4 M pairs unless COPIES says otherwise).
#   tools/collect_bam_profiles.sh r05        -> gpurun_out/prof_<tag>/<tag>_bam_kernel_stats.csv  (copy into profiles/)
set -u
TAG=${1:?round tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/bam" -o bam --output-format csv -- python3 "$ROOT/tools/bench_bam.py" --copies ${COPIES:-48000} \
    > "$OUT/bench_bam.json" 2> "$OUT/bam.err" || { echo "bam trace failed"; tail -5 "$OUT/bam.err"; exit 1; }
F=$(find "$OUT/bam" -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$OUT/${TAG}_bam_kernel_stats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as fh:                 # kernel names without their argument lists (they are hundreds of characters long)
    w = csv.writer(fh)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        name = re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", r["Name"])).split("(")[0]
        if not name.startswith("at::"):
            w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
cat "$OUT/${TAG}_bam_kernel_stats.csv"
tail -1 "$OUT/bench_bam.json" | cut -c1-300
# the timed (second) pass as the GPU saw it: busy time, share per kernel, overlap, idle gaps (tools/trace_gaps.py)
T=$(find "$OUT/bam" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/trace_gaps.py" "$T" --second-pass --top 14 --timeline 90 > "$OUT/${TAG}_bam_timeline.txt" 2>&1
cat "$OUT/${TAG}_bam_timeline.txt"
rm -f "$T"
echo done
