#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_full_chain_gpu.py tests/test_clutter_gpu.py tests/test_replay_gpu.py -m gpu -q -s > $OUT/pytest_c.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_c.log
grep -E "^\[|passed|failed|rc=" $OUT/pytest_c.log | tail -n 30
python tools/gpu_solve_diag.py > $OUT/solve_diag.log 2>&1; tail -n 12 $OUT/solve_diag.log
python bench.py --chain full --batch 64 --no-cpu-baseline > $OUT/bench_r2_full.log 2>&1
python bench.py --config cfg3 --chain full --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_r2_cfg3_full.log 2>&1
python bench.py --config cfg3 --chain full --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --no-parity > $OUT/bench_r2_cfg3_full64.log 2>&1
for f in $OUT/bench_r2_full.log $OUT/bench_r2_cfg3_full*.log; do echo "== $f"; tail -c 700 $f; echo; done
