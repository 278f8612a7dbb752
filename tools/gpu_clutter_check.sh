#!/bin/bash
# clutter-filter parity tests + Toeplitz-solve timing + full-chain benches (run through gpurun)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_full_chain_gpu.py tests/test_clutter_gpu.py tests/test_replay_gpu.py -m gpu -q -s > $OUT/pytest_c.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_c.log
grep -E "^\[clutter|^\[coloured|passed|failed|rc=" $OUT/pytest_c.log | tail -n 12
python tools/gpu_solve_diag.py > $OUT/solve_diag.log 2>&1; grep -v "K=4\|error" $OUT/solve_diag.log | tail -n 9
python bench.py --config cfg3 --chain full --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_c3f.log 2>&1
python - <<PY
import json
j=json.loads(open("$OUT/bench_c3f.log").read().strip().split("\n")[-1])
print("cfg3 full: %.0f CPIs/s %.1f us/CPI"%(j["value"], j["us_per_cpi"]), {k["kernel"]:round(k["us_per_cpi"],1) for k in j["roofline"]["kernels"]}, j["parity"]["pass"], j["parity"].get("chain_err_over_direct_path"))
PY
