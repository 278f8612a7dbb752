#!/bin/bash
# shader/memory clocks and power while the headline bench runs (run through gpurun)
set -u
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
rocm-smi --showclocks --showpower --showperflevel --showmaxpower > $OUT/smi_idle.log 2>&1
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT/smi_run.log 2>&1 &
SMI=$!
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-parity ${@} > $OUT/bench_clock.log 2>&1
kill $SMI 2>/dev/null
grep -E "sclk|Max|Perf" $OUT/smi_idle.log | head -8
sort $OUT/smi_run.log | uniq -c | sort -rn | head -8
python - <<PY
import json
j=json.loads(open("$OUT/bench_clock.log").read().strip().split("\n")[-1]); print(round(j["value"]), j["roofline"]["kernel_us_per_step"])
PY
