#!/bin/bash
# Builds the instrumented variants of the library the phase tables of DESIGN.md section 4 come from
# (s_memtime buckets per wave, printed to stderr by the 8th launch) into tools/ab/, which travels
# to the GPU box with the tree:
#   bash tools/build_trace.sh
#   gpurun -- 'BLAH2HIP_LIBRARY=$PWD/tools/ab/lib_rangew_trace.so python bench.py --steps 10 --warmup 3 --no-parity --no-cpu-baseline 2>&1 | grep trace'
#   gpurun -- 'BLAH2HIP_LIBRARY=$PWD/tools/ab/lib_dopw_trace.so python bench.py --config cfg3 --steps 10 --warmup 3 --no-parity --no-cpu-baseline 2>&1 | grep trace'
# The product library never contains this code (it is compiled out without the macros).
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/ab
for v in ${TRACE_VARIANTS:-rangew:RANGEW_TRACE dopw:DOPW_TRACE c2t:C2T_TRACE sla:SLA_TRACE}; do
  n=${v%%:*}; m=${v##*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -fno-slp-vectorize -D$m \
    -I include -I blah2_amd/csrc blah2_amd/csrc/capi.hip blah2_amd/csrc/clutter.hip blah2_amd/csrc/spectrum.hip \
    -o tools/ab/lib_${n}_trace.so
  echo "tools/ab/lib_${n}_trace.so"
done
