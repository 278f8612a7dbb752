#!/bin/bash
# doppler_tilew4_kernel: parity (persistent-kernel test), then configs[4] with the Doppler stage on tilew2 (auto) and tilew4, interleaved
set -u
timeout 600 python -m pytest tests/test_persistent_kernels_gpu.py -q -x -k "one_wave_4096" 2>&1 | tail -15
P='import sys,json; j=json.loads(sys.stdin.read().strip().split("\n")[-1]); B=j["config"]["batch_cpis_per_step"]; print(sys.argv[1], round(j["value"]), {k:round(v/B,3) for k,v in j["roofline"]["kernel_us_per_step"].items()}, round(j["roofline"]["chain_frac"],4), (j.get("parity") or {}).get("pass"), (j.get("parity") or {}).get("peak_rel"))'
for rep in 1 2; do
  for k in auto tilew4; do
    timeout 300 python bench.py --no-cpu-baseline --no-configs --config cfg5 --fmt f16 --steps 20 --warmup 3 --doppler-kernel $k 2>&1 | python -c "$P" $k
  done
done
timeout 300 python bench.py --no-cpu-baseline --no-configs --config cfg5 --fmt f16 --batch 32 --steps 8 --warmup 2 --doppler-kernel tilew4 2>&1 | python -c "$P" tilew4_b32
