#!/bin/bash
# clutter-filter parity tests + Toeplitz-solve timing + full-chain benches (run through gpurun)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_full_chain_gpu.py tests/test_clutter_gpu.py tests/test_replay_gpu.py -m gpu -q -s > $OUT/pytest_c.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_c.log
grep -E "^\[clutter|^\[coloured|passed|failed|rc=" $OUT/pytest_c.log | tail -n 12
python tools/gpu_solve_diag.py > $OUT/solve_diag.log 2>&1; grep -v "K=4\|error" $OUT/solve_diag.log | tail -n 9
