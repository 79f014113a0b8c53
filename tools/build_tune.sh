#!/bin/bash
# builds /tmp/tune (or $1): the on-box A/B harness of the kernels (tools/tune_kernels.hip)
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/tune}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -I include -I xenomapper_amd/csrc $XM_TUNE_FLAGS tools/tune_kernels.hip -o "$OUT"
echo "built $OUT"
