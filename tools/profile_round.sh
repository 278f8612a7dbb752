#!/bin/bash
# Runs on the GPU box (through gpurun) from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh'
# then, back in the container:  python tools/summarize_prof.py r06 && python tools/collect_bench.py r06
# For each profiled bench command (tag -> arguments below) leaves under gpurun_out/prof/<tag>/:
# the rocprofv3 --kernel-trace --stats run and the separate PMC passes (never combined with a
# trace domain other than --kernel-trace), plus the plain bench logs.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd /tmp && export TMPDIR=/tmp
prune() { # dir: keep what tools/summarize_prof.py reads (the stats table; the counter rows of OUR kernels) -- gpurun copies back 64 MiB at most
  find "$1" -name "*kernel_trace.csv" -delete 2>/dev/null
  for f in $(find "$1" -name "*_counter_collection.csv" 2>/dev/null); do
    { head -1 "$f"; grep -E 'blah2|anonymous namespace|sla::|rd<|wr<|cal_' "$f" | grep -v '^"Correlation_Id"'; } > "$f.tmp" && mv "$f.tmp" "$f"
  done
}
profile() { # tag, json description, bench args...
  local tag=$1; local desc=$2; shift 2
  local B="python $REPO/bench.py $* --no-cpu-baseline --no-parity --no-configs --long-s 0"
  mkdir -p $OUT/prof/$tag
  echo "$desc" > $OUT/prof/$tag/bench_config.json
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof/$tag/trace -o bench --output-format csv -- $B > $OUT/prof/$tag/trace.log 2>&1
  for pass in "FETCH_SIZE" "WRITE_SIZE" ${PMC_EXTRA:+"SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"}; do
    local ptag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    timeout 400 rocprofv3 --kernel-trace --pmc $pass -d $OUT/prof/$tag/pmc_$ptag -o bench --output-format csv -- $B > $OUT/prof/$tag/pmc_$ptag.log 2>&1 || echo "pmc pass $pass failed ($tag)"
  done
  prune $OUT/prof/$tag
}
PMC_EXTRA=1 profile amb '{"config": "cfg2", "batch": 256, "fmt": "c32", "chain": "amb"}' --steps 12 --warmup 3
PMC_EXTRA= profile full '{"config": "cfg2", "batch": 256, "fmt": "c32", "chain": "full"}' --chain full --steps 6 --warmup 2
PMC_EXTRA= profile cfg3 '{"config": "cfg3", "batch": 32, "fmt": "c32", "chain": "amb"}' --config cfg3 --steps 10 --warmup 2 --prewarm-s 0.3
PMC_EXTRA= profile cfg3_full '{"config": "cfg3", "batch": 32, "fmt": "c32", "chain": "full", "fir": "fused"}' --config cfg3 --chain full --cfar 2d --batch 32 --streams 2 --steps 10 --warmup 2 --prewarm-s 0.3
PMC_EXTRA= profile cfg3_full_twostage '{"config": "cfg3", "batch": 32, "fmt": "c32", "chain": "full-two-stage", "fir": "two-stage"}' --config cfg3 --chain full --cfar 2d --batch 32 --streams 2 --fir two-stage --steps 10 --warmup 2 --prewarm-s 0.3
PMC_EXTRA= profile cfg5 '{"config": "cfg5", "batch": 8, "fmt": "f16", "chain": "amb"}' --config cfg5 --fmt f16 --steps 10 --warmup 2
mkdir -p $OUT/cal
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/cal/fetch -o cal --output-format csv -- $REPO/tools/membench/pmccal > $OUT/cal/fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/cal/write -o cal --output-format csv -- $REPO/tools/membench/pmccal > $OUT/cal/write.log 2>&1
prune $OUT/cal
cd $REPO
rm -f $OUT/bench_r6*.log
python bench.py > $OUT/bench_r6.log 2>&1                      # the default line: headline + configs[] legs + cpu_baseline + e2e_host
python bench.py --fmt i16 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_i16.log 2>&1
python bench.py --chain full --steps 20 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_full.log 2>&1
python bench.py --config cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_cfg3.log 2>&1
python bench.py --config cfg3 --chain full --cfar 2d --batch 32 --streams 2 --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_cfg3_full.log 2>&1
python bench.py --config cfg3 --chain full --cfar 2d --batch 32 --streams 2 --fir two-stage --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_cfg3_full_twostage.log 2>&1
python bench.py --config cfg5 --fmt f16 --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_cfg5.log 2>&1
python bench.py --batch 1 --steps 2000 --warmup 50 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_b1.log 2>&1
python bench.py --chain full --cfar 1d --batch 1 --steps 500 --warmup 20 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_full_b1.log 2>&1
python bench.py --config small --steps 40 --warmup 3 --no-cpu-baseline --no-configs --no-replay > $OUT/bench_r6_small.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r6_torchrun.log 2>&1
# the Toeplitz solve on its own: HIP-event time per launch by taps / batch / form, and rocprofv3's kernel durations of the same
python tools/gpu_solve.py --json $OUT/solve_timing.json > $OUT/solve_timing.log 2>&1
mkdir -p $OUT/prof/solve
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/solve/trace -o bench --output-format csv -- python $REPO/tools/gpu_solve.py --quick > $OUT/prof/solve/trace.log 2>&1)
# the lone-CPI chain (small-launch kernels) under the profiler
mkdir -p $OUT/prof/b1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/b1/trace -o bench --output-format csv -- python $REPO/bench.py --batch 1 --steps 500 --warmup 20 --no-cpu-baseline --no-parity --no-configs > $OUT/prof/b1/trace.log 2>&1)
mkdir -p $OUT/prof/b1full
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/b1full/trace -o bench --output-format csv -- python $REPO/bench.py --chain full --cfar 1d --batch 1 --steps 500 --warmup 20 --no-cpu-baseline --no-parity --no-configs --no-replay --long-s 0 > $OUT/prof/b1full/trace.log 2>&1)
prune $OUT/prof/solve; prune $OUT/prof/b1; prune $OUT/prof/b1full
# the fused FIR + range kernel beside the two kernels it replaces (configs[2], 16 CPIs per launch)
bash $REPO/tools/prof_fused_ab.sh 16 10 > $OUT/fused_fir_ab.txt 2>&1
# the spread of the full chain's gates over 16 CPIs of configs[2] and configs[1]
(python $REPO/tools/gpu_chain_gate_stats.py cfg3 16 2d; python $REPO/tools/gpu_chain_gate_stats.py cfg2 16 1d) 2>&1 | grep -v amdgpu > $OUT/chain_gate_stats.txt
du -sh $OUT
tail -qn 1 $OUT/bench_r6*.log | cut -c1-200
