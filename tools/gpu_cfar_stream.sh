#!/bin/bash
# GPU box: parity of the stream 2-D CFAR kernel, then its time beside the tile kernel, for the C2S_V builds in tools/ab/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cfar_gpu.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_full_chain_gpu.py -x -q -k "cfar2d or dev" 2>&1 | tail -5
for cfg in cfg3 cfg2; do
  timeout 300 python tools/gpu_cfar_diag.py $cfg 64 stream,tile 0 2>&1 | tail -4
  for v in 1 4; do
    BLAH2HIP_LIBRARY=$PWD/tools/ab/lib_c2s_v$v.so timeout 300 python tools/gpu_cfar_diag.py $cfg 64 stream 0 2>&1 | tail -2 | sed "s/^/V=$v /"
  done
done
timeout 300 python tools/gpu_cfar_diag.py cfg3 1 stream,tile 0 2>&1 | tail -4
