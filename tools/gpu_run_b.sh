#!/bin/bash
# GPU session B (round 2): full GPU suite (all failures listed), Toeplitz-solve variants, full-chain benches.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT/cal
rocm-smi --showclocks --showpower > $OUT/smi_before.log 2>&1
python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 8 $OUT/pytest.log
python tools/gpu_solve_diag.py > $OUT/solve_diag.log 2>&1; tail -n 12 $OUT/solve_diag.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_r2_long.log 2>&1
rocm-smi --showclocks --showpower > $OUT/smi_after.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r2.log 2>&1
python bench.py --chain full --batch 64 --no-cpu-baseline > $OUT/bench_r2_full.log 2>&1
python bench.py --config cfg3 --chain full --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_r2_cfg3_full.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/cal/fetch -o cal --output-format csv -- $REPO/tools/membench/pmccal > $OUT/cal/fetch.log 2>&1
cd $REPO
for f in $OUT/bench_r2*.log; do echo "== $f"; tail -c 300 $f; echo; done
