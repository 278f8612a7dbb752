#!/bin/bash
# headline chain with the Doppler stage on doppler_tile1k_kernel<16> (auto) and <8> (tile8k), interleaved, same box
set -u
python -m pytest tests/test_persistent_kernels_gpu.py tests/test_timed_kernels_gpu.py -q -k "tile8k or tile_kernels or every_doppler" 2>&1 | tail -3
P='import sys,json; j=json.loads(sys.stdin.read().strip().split("\n")[-1]); B=j["config"]["batch_cpis_per_step"]; print(sys.argv[1], round(j["value"]), {k:round(v/B,3) for k,v in j["roofline"]["kernel_us_per_step"].items()}, j["roofline"]["chain_frac"], (j.get("parity") or {}).get("pass"))'
for rep in 1 2 3; do
  for k in auto tile8k; do
    python bench.py --no-cpu-baseline --no-configs --steps 40 --doppler-kernel $k 2>&1 | python -c "$P" $k
  done
done
python bench.py --no-cpu-baseline --no-configs --steps 40 --batch 64 --doppler-kernel tile8k 2>&1 | python -c "$P" tile8k_b64
python bench.py --no-cpu-baseline --no-configs --steps 40 --batch 64 2>&1 | python -c "$P" auto_b64
