#!/bin/bash
# PMC traffic of the configs[2] full chain with the B2_FIR_CARRY variant (tools/ab/libblah2hip_carry.so)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; tag=cfg3_full_carry
export BLAH2HIP_LIBRARY=$REPO/tools/ab/libblah2hip_carry.so
mkdir -p $OUT/prof/$tag; cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --config cfg3 --chain full --cfar 2d --batch 32 --streams 2 --steps 10 --warmup 2 --prewarm-s 0.3 --no-cpu-baseline --no-parity --no-configs --long-s 0"
echo '{"config": "cfg3", "batch": 32, "fmt": "c32", "chain": "full-carry", "fir": "fused"}' > $OUT/prof/$tag/bench_config.json
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof/$tag/trace -o bench --output-format csv -- $B > $OUT/prof/$tag/trace.log 2>&1
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $pass -d $OUT/prof/$tag/pmc_$pass -o bench --output-format csv -- $B > $OUT/prof/$tag/pmc_$pass.log 2>&1
done
find $OUT/prof/$tag -name "*kernel_trace.csv" -delete
for f in $(find $OUT/prof/$tag -name "*_counter_collection.csv"); do { head -1 "$f"; grep -E 'blah2|anonymous namespace|sla::' "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"; done
du -sh $OUT/prof/$tag
