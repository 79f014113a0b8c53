#!/bin/bash
# Tuning builds of libxenomapper_hip.so for same-box A/B runs: one library per "name:flags" argument under build/ab/
# (in-tree, so it travels with gpurun; git-ignored).  Run with XENOMAPPER_HIP_LIB=build/ab/<name>.so python bench.py ...
#   tools/ab_build.sh base: wpe8:-DXM_CIGP_WPE=8
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  $HIPCC -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I include $flags xenomapper_amd/csrc/xm_kernels.hip xenomapper_amd/csrc/xm_api.hip xenomapper_amd/csrc/xm_strip.hip xenomapper_amd/csrc/xm_inflate.hip xenomapper_amd/csrc/xm_bamdev.hip -o build/ab/$name.so &
done
wait
ls -la build/ab
