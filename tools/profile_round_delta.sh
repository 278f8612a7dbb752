#!/bin/bash
# The part of tools/profile_round.sh that the second half of round 4 changed (the stream 2-D CFAR kernel, the workgroup
# transform's exchange reads, the FIR's carried overlap, --streams for the full chain, the replay read modes): re-profiles
# `full`, `cfg3` and `cfg3_full`, re-runs their bench lines and the replay figures.
#   gpurun --timeout 1500 -- 'bash tools/profile_round_delta.sh'
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd /tmp && export TMPDIR=/tmp
profile() { # tag, json description, bench args...
  local tag=$1; local desc=$2; shift 2
  local B="python $REPO/bench.py $* --no-cpu-baseline --no-parity"
  rm -rf $OUT/prof/$tag; mkdir -p $OUT/prof/$tag
  echo "$desc" > $OUT/prof/$tag/bench_config.json
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof/$tag/trace -o bench --output-format csv -- $B > $OUT/prof/$tag/trace.log 2>&1
  for pass in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout 400 rocprofv3 --kernel-trace --pmc $pass -d $OUT/prof/$tag/pmc_$pass -o bench --output-format csv -- $B > $OUT/prof/$tag/pmc_$pass.log 2>&1 || echo "pmc pass $pass failed ($tag)"
  done
}
profile full '{"config": "cfg2", "batch": 256, "fmt": "c32", "chain": "full"}' --chain full --steps 6 --warmup 2
profile cfg3 '{"config": "cfg3", "batch": 32, "fmt": "c32", "chain": "amb"}' --config cfg3 --steps 10 --warmup 2 --prewarm-s 0.3
profile cfg3_full '{"config": "cfg3", "batch": 256, "fmt": "c32", "chain": "full"}' --config cfg3 --chain full --steps 2 --warmup 1 --prewarm-s 0.3
cd $REPO
python bench.py --chain full --steps 20 --no-cpu-baseline > $OUT/bench_r4_full.log 2>&1
python bench.py --chain full --batch 64 --streams 2 --steps 40 --no-cpu-baseline > $OUT/bench_r4_full_b64_s2.log 2>&1
python bench.py --config cfg3 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_r4_cfg3.log 2>&1
python bench.py --config cfg3 --chain full --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_r4_cfg3_full.log 2>&1
python bench.py --config cfg3 --chain full --batch 64 --streams 2 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_r4_cfg3_full_b64_s2.log 2>&1
python bench.py --config cfg3 --chain full --streams 2 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_r4_cfg3_full_s2.log 2>&1
python bench.py --config cfg3 --chain full --batch 32 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_r4_cfg3_full_b32.log 2>&1
python bench.py --config cfg3 --chain full --batch 32 --streams 2 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_r4_cfg3_full_b32_s2.log 2>&1
python bench.py --chain full --batch 1 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_r4_full_b1.log 2>&1
python tools/replay_bench.py --out $OUT/replay.json > $OUT/replay_bench.log 2>&1
tail -qn 1 $OUT/bench_r4_full*.log $OUT/bench_r4_cfg3*.log | cut -c1-160
grep -c chain $OUT/replay.json
