#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; rm -rf $OUT/prof_e; mkdir -p $OUT/prof_e
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_e/fetch -o bench --output-format csv -- python $REPO/bench.py --config cfg3 --chain full --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $OUT/prof_e/fetch.log 2>&1
python - <<PY
import csv,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/prof_e/fetch/bench_counter_collection.csv")):
    agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "clutter" in k or "range" in k: print(k, len(v), "%.1f MB x2 = %.1f MB"%(sum(v)/len(v)*1024/1e6, 2*sum(v)/len(v)*1024/1e6))
PY
