#!/bin/bash
# Builds an EXPERIMENT variant of the HIP library into tools/ab/ (git-ignored, travels to the GPU box with the tree):
#   bash tools/build_variant.sh <name> <extra hipcc flags...>       e.g.  bash tools/build_variant.sh glds -DB2_RANGEW1K_GLDS
# Select it at run time with BLAH2HIP_LIBRARY=$PWD/tools/ab/libblah2hip_<name>.so (blah2_amd/_lib.py); the product library
# (blah2_amd/libblah2hip.so) is never touched.
set -eu
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -fno-slp-vectorize "$@" \
  -I include -I blah2_amd/csrc blah2_amd/csrc/capi.hip blah2_amd/csrc/clutter.hip blah2_amd/csrc/spectrum.hip \
  -o tools/ab/libblah2hip_${name}.so
echo "tools/ab/libblah2hip_${name}.so"
