#!/bin/bash
# GPU session A (round 2): full GPU test suite, the bench lines, the F=1024 range kernel A/B and the
# PMC calibration.  Run from the repo root through gpurun; everything lands under gpurun_out/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT/cal
python -m pytest tests -m gpu -q -x -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 5 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_r2.log 2>$OUT/bench_r2.err; echo "bench rc=$?"
python bench.py --chain full --batch 64 --no-cpu-baseline > $OUT/bench_r2_full.log 2>&1
python bench.py --config cfg3 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_r2_cfg3.log 2>&1
python bench.py --config cfg3 --chain full --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_r2_cfg3_full.log 2>&1
python bench.py --config cfg5 --fmt f16 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_r2_cfg5.log 2>&1
python bench.py --config cfg5 --fmt f16 --steps 10 --warmup 2 --no-cpu-baseline --no-parity --doppler-kernel column > $OUT/bench_r2_cfg5_column.log 2>&1
python bench.py --config small --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r2_small_w4.log 2>&1
BLAH2HIP_LIBRARY=$REPO/tools/ab/libblah2hip_r8w3.so python bench.py --config small --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r2_small_w3.log 2>&1
python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2.log 2>&1; echo "gpus2 rc=$? (expected non-zero on a 1-GPU box)" >> $OUT/bench_gpus2.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/cal/fetch -o cal --output-format csv -- $REPO/tools/membench/pmccal > $OUT/cal/fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/cal/write -o cal --output-format csv -- $REPO/tools/membench/pmccal > $OUT/cal/write.log 2>&1
cd $REPO
for f in $OUT/bench_r2*.log; do echo "== $f"; tail -c 600 $f; echo; done
