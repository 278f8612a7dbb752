#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
for s in 1 2 3 1 2; do
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity --streams $s > $OUT/bench_streams_$s.log 2>&1
  python - <<PY
import json
j=json.loads(open("$OUT/bench_streams_$s.log").read().strip().split("\n")[-1])
print("streams $s: %.0f CPIs/s  %.2f us/CPI"%(j["value"], j["us_per_cpi"]), j["roofline"]["kernel_us_per_step"])
PY
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
