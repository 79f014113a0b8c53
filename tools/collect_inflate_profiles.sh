#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 kernel stats + PMC passes of the GPU BGZF decoder on ~1 GB of inflated BAM, with the
# stand-alone checker (tools/inflate_gpu_check.hip -> build/inflate_gpu_check: every block against zlib; the program itself follows `--`).
#   tools/collect_inflate_profiles.sh r05      -> gpurun_out/prof_<tag>/inflate_*  (copy the summaries into profiles/)
set -u
TAG=${1:?round tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export LD_LIBRARY_PATH=$ROOT/xenomapper_amd:${LD_LIBRARY_PATH:-}
BAM=/dev/shm/xm_prof_1g.bam
python3 -c "
import sys; sys.path.insert(0, '$ROOT/tools'); import bench_bam
bench_bam.tiled_bam('$ROOT/tests/golden/ref_data/paired_end_testdata_human.bam', '$BAM', 8400)" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/inflate_stats" -o inflate --output-format csv -- "$ROOT/build/inflate_gpu_check" --reps 5 $BAM > "$OUT/inflate_check.txt" 2> "$OUT/inflate_stats.err" || { echo "stats failed"; tail -5 "$OUT/inflate_stats.err"; exit 1; }
cat "$OUT/inflate_check.txt" | cut -c1-300
F=$(find "$OUT/inflate_stats" -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/${TAG}_inflate_kernel_stats.csv"
grep -i "inflate\|crc32" "$OUT/${TAG}_inflate_kernel_stats.csv" | cut -c1-200
N=1
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 200 rocprofv3 --pmc $SET -d "$OUT/inflate_pmc$N" -o inflate --output-format csv -- "$ROOT/build/inflate_gpu_check" --reps 2 $BAM > /dev/null 2> "$OUT/inflate_pmc$N.err" || { echo "pmc $N failed"; tail -5 "$OUT/inflate_pmc$N.err"; exit 1; }
  N=$((N + 1))
done
rm -f $BAM
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(out + "/inflate_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        name = "inflate_kernel" if "inflate_kernel" in k else "crc32_kernel" if "crc32_kernel" in k else None
        if name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open("%s/%s_inflate_pmc.json" % (out, tag), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
echo done
