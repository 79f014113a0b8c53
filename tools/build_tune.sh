#!/bin/bash
# builds /tmp/tune (or $1): the A/B harness = current kernels + the frozen round-1 kernels
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/tune}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -Dxm=xm_r01 -I include -c tools/legacy/xm_kernels_r01.hip -o /tmp/xm_legacy_r01.o
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -I include -I xenomapper_amd/csrc $XM_TUNE_FLAGS -c tools/tune_kernels.hip -o /tmp/xm_tune_main.o
$HIPCC --offload-arch=gfx950 /tmp/xm_tune_main.o /tmp/xm_legacy_r01.o -o "$OUT"
echo "built $OUT"
