#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-4}
OUT=$ROOT/gpurun_out/prof_qc$N
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/t" -o t --output-format csv -- python3 "$ROOT/tools/probe_queue_collision.py" $N > "$OUT/out.txt" 2> "$OUT/err.txt"
tail -1 "$OUT/out.txt"
T=$(find "$OUT/t" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/trace_gaps.py" "$T" --second-pass --top 6 --timeline 60 > "$OUT/timeline.txt" 2>&1
python3 - "$T" > "$OUT/queues.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
q = collections.defaultdict(collections.Counter)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-40:]
    q[r["Queue_Id"]][name] += 1
for k, c in q.items():
    print("queue", k, dict(c.most_common(8)))
PY
rm -f "$T"
