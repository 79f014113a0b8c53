#!/bin/bash
# Runs ON the GPU box: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes, --pmc alone) of the BAM path's kernels.
#   tools/pmc_bam_kernels.sh [copies] > gpurun_out/r6/pmc_bam.txt
COPIES=${1:-24000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_bam
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C -d "$OUT/$C" -o bam --output-format csv -- python3 "$ROOT/tools/bench_bam.py" --copies $COPIES > /dev/null 2> "$OUT/$C.err" || { echo "pmc $C failed"; tail -3 "$OUT/$C.err"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            name = re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])).split("(")[0]
            d = tot.setdefault(name, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            d[c][0] += float(r["Counter_Value"])
            d[c][1] += 1
print("%-28s %8s %14s %14s   (KiB counters x 1024 / launches; FETCH_SIZE raw, not doubled)" % ("kernel", "launches", "read MB/launch", "write MB/launch"))
for name, d in sorted(tot.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"][0] + kv[1]["WRITE_SIZE"][0])):
    n = max(d["FETCH_SIZE"][1], d["WRITE_SIZE"][1], 1)
    if name.startswith("at::"):
        continue
    print("%-28s %8d %14.1f %14.1f" % (name[:28], n, d["FETCH_SIZE"][0] * 1024 / 1e6 / max(d["FETCH_SIZE"][1], 1), d["WRITE_SIZE"][0] * 1024 / 1e6 / max(d["WRITE_SIZE"][1], 1)))
PY
