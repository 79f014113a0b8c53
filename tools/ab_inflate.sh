#!/bin/bash
# Same-box A/B of inflate kernel builds (ON the GPU box): build/ab/<name>/libxenomapper_hip.so for every name given, the stand-alone
# checker (every block against zlib) on 1 GB of tiled BAM, three rounds in rotation.   tools/ab_inflate.sh glob flat
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
BAM=/dev/shm/xm_ab_1g.bam
python3 -c "
import sys; sys.path.insert(0, 'tools'); import bench_bam
bench_bam.tiled_bam('tests/golden/ref_data/paired_end_testdata_human.bam', '$BAM', 8400)" || exit 1
for round in 1 2 3; do
  for n in "$@"; do
    echo -n "round $round $n: "
    LD_LIBRARY_PATH=$ROOT/build/ab/$n:${LD_LIBRARY_PATH:-} timeout -k 10 120 build/inflate_gpu_check --reps 3 $BAM | cut -c1-230 || { echo "failed"; rm -f $BAM; exit 1; }
  done
done
rm -f $BAM
