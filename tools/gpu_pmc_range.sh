#!/bin/bash
# Wave-time decomposition and memory-path stall counters of the range kernel (bench default command).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof/amb2; rm -rf $OUT; mkdir -p $OUT
echo '{"config": "cfg2", "batch": 128, "fmt": "c32", "chain": "amb", "note": "packed-arithmetic core"}' > $OUT/bench_config.json
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- $B > $OUT/trace.log 2>&1
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT/pmc_$i -o bench --output-format csv -- $B > $OUT/pmc_$i.log 2>&1 || echo "pass $pass failed"
done
cd $REPO; python tools/summarize_prof.py r02 | head -3
grep range_kernel profiles/r02_amb2_pmc.csv | cut -d, -f2-4
