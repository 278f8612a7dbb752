#!/bin/bash
# GPU box: why the cfg 5 range kernel is slower per CPI at batch 32 than at batch 8 -- the same counters for both launches
# (each counter set in its own run, only --kernel-trace beside --pmc).  Output: gpurun_out/cfg5_batch/<batch>/<set>/..., and
# the list of available counters once (names differ between ROCm releases).
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/cfg5_batch
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep "Counter_Name" | awk '{print $NF}' | sort -u > $OUT/avail.txt
for B in 8 32; do
  mkdir -p $OUT/$B
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "TCC_TAG_STALL_sum TCC_BUSY_sum" "TCC_EA_RDREQ_DRAM_sum TCC_EA_RD_UNCACHED_32B_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE GRBM_EA_BUSY GRBM_TC_BUSY"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/$B/$tag -o bench --output-format csv -- python $REPO/bench.py --config cfg5 --fmt f16 --batch $B --steps 4 --warmup 1 --no-cpu-baseline --no-parity > $OUT/$B/$tag.log 2>&1 || echo "pass failed: B=$B $set"
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$B/trace -o bench --output-format csv -- python $REPO/bench.py --config cfg5 --fmt f16 --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-parity > $OUT/$B/trace.log 2>&1
done
ls $OUT/8 $OUT/32 | head -40
