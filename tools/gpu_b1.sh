#!/bin/bash
# lone-CPI chain: same-box A/B of the tree against tools/ab/*.so, then rocprofv3 kernel stats of the tree
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT/prof/b1
bash tools/gpu_1k.sh --batch 1 --steps 2000 --warmup 50 --no-configs ${B1_ARGS:-} 2>&1 | tee $OUT/ab_b1.log
unset BLAH2HIP_LIBRARY
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/b1/trace -o bench --output-format csv -- python $REPO/bench.py --batch 1 --steps 500 --warmup 20 --no-cpu-baseline --no-parity --no-configs > $OUT/prof/b1/trace.log 2>&1)
find $OUT/prof/b1 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -3
