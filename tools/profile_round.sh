#!/bin/bash
# Runs on the GPU box (through gpurun) from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh'
# then, back in the container:  python tools/summarize_prof.py r01
# Leaves under gpurun_out/: bench logs, rocprofv3 kernel-trace stats and the
# separate PMC passes (never combined with a trace domain) for the bench command.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
BATCH=${BATCH:-128}
mkdir -p $OUT/prof
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5 --batch $BATCH --no-cpu-baseline"
echo "{\"config\": \"cfg2\", \"batch\": $BATCH, \"fmt\": \"c32\"}" > $OUT/prof/bench_config.json
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/trace -o bench --output-format csv -- $B > $OUT/prof/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT/prof/pmc_$tag -o bench --output-format csv -- $B > $OUT/prof/pmc_$tag.log 2>&1 || echo "pmc pass $pass failed"
done
cd $REPO
python bench.py > $OUT/bench_r1.log 2>&1
python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_r1_b1.log 2>&1
python bench.py --fmt i16 --no-cpu-baseline > $OUT/bench_r1_i16.log 2>&1
python bench.py --chain full --batch 64 --no-cpu-baseline > $OUT/bench_r1_full.log 2>&1
python bench.py --config cfg3 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_r1_cfg3.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r1_torchrun.log 2>&1
tail -n 1 $OUT/bench_r1*.log
