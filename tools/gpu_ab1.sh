#!/bin/bash
# A/B of library builds on ONE bench command, interleaved twice (run through gpurun):
#   bash tools/gpu_ab1.sh <bench args...>
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
for rep in 1 2; do
for lib in "" $(ls tools/ab/*.so 2>/dev/null); do
  t=tree; [ -n "$lib" ] && t=$(basename $lib .so)
  if [ -n "$lib" ]; then export BLAH2HIP_LIBRARY=$REPO/$lib; else unset BLAH2HIP_LIBRARY; fi
  python bench.py --no-cpu-baseline "$@" > $OUT/ab1_$t.log 2> $OUT/ab1_$t.err
  python - <<PY
import json
try:
    j=json.loads(open("$OUT/ab1_$t.log").read().strip().split("\n")[-1])
    k=j["roofline"]["kernel_us_per_step"]; B=j["config"]["batch_cpis_per_step"]
    print("$t: %.0f CPIs/s  "%j["value"] + " ".join("%s %.2f"%(n,v/B) for n,v in k.items()) + "  parity %s"%((j["parity"] or {}).get("pass")))
except Exception as e:
    print("$t: FAILED", e); print(open("$OUT/ab1_$t.err").read()[-400:])
PY
done
done
