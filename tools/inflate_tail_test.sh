#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export LD_LIBRARY_PATH=$ROOT/xenomapper_amd:$LD_LIBRARY_PATH
for c in 4090 4100 8180 8190 8400 12280 12290 16380; do
python3 -c "
import sys; sys.path.insert(0, 'tools'); import bench_bam
bench_bam.tiled_bam('tests/golden/ref_data/paired_end_testdata_human.bam', '/dev/shm/xm_tail.bam', $c)" || exit 1
echo -n "copies $c: "; timeout -k 10 120 build/inflate_gpu_check --reps 3 /dev/shm/xm_tail.bam | cut -c24-160
done
rm -f /dev/shm/xm_tail.bam
