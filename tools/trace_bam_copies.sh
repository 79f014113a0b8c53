cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6/memcopy
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o bam --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_bam.py --copies 48000 > $OUT/bench.json 2> $OUT/err.txt
ls $OUT
F=$(find $OUT -name "*memory_copy_trace.csv" | head -1)
head -2 $F
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "copies")
big = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:25]
for r in big:
    print({k: r[k] for k in r if k in ("Direction", "Bytes", "Source_Agent_Id", "Destination_Agent_Id")}, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms")
from collections import Counter
c = Counter()
for r in rows:
    c[(r.get("Direction"), int(r.get("Bytes", 0)) // (1 << 20))] += 1
print(sorted(c.items(), key=lambda kv: -kv[1])[:30])
PY
rm -f $(find $OUT -name "*kernel_trace.csv")
