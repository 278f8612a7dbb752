#!/bin/bash
# Same-box interleaved A/B of the LDS-DMA range kernel experiment (tools/build_variant.sh glds -DB2_RANGEW1K_GLDS):
#   gpurun -- 'bash tools/gpu_ab_glds.sh'
set -u
V=$PWD/tools/ab/libblah2hip_glds.so
echo "== parity of the variant (the persistent / timed-kernel tests that run rangew1k_kernel)"
BLAH2HIP_LIBRARY=$V python -m pytest tests/test_timed_kernels_gpu.py tests/test_persistent_kernels_gpu.py tests/test_ambiguity_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for lib in default glds; do
    if [ $lib = glds ]; then export BLAH2HIP_LIBRARY=$V; else unset BLAH2HIP_LIBRARY; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-parity --long-s 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); rl=r['roofline']
print('$lib', 'us/CPI %.3f' % r['us_per_cpi'], 'range us/launch %.1f' % rl['avg_launch_us'], 'frac %.4f' % rl['frac'], 'kernel', rl['kernel'])"
  done
done
