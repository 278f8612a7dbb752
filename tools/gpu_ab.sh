#!/bin/bash
# A/B of library builds on the benches that cover every FFT kernel (run through gpurun).
#   tools/ab/*.so are the alternatives; the in-tree library is "tree"
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
run() { # tag lib args...
  if [ -n "$2" ]; then export BLAH2HIP_LIBRARY=$2; else unset BLAH2HIP_LIBRARY; fi
  python bench.py --no-cpu-baseline ${@:3} > $OUT/ab_$1.log 2> $OUT/ab_$1.err
  python - <<PY
import json
try:
    j=json.loads(open("$OUT/ab_$1.log").read().strip().split("\n")[-1])
    k=j["roofline"]["kernel_us_per_step"]; B=j["config"]["batch_cpis_per_step"]
    print("$1: %.0f CPIs/s  "%j["value"] + " ".join("%s %.2f"%(n,v/B) for n,v in k.items()) + "  parity %s"%((j["parity"] or {}).get("pass")))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/ab_$1.err").read()[-400:])
PY
}
for lib in "" $(ls tools/ab/*.so 2>/dev/null); do
  t=tree; [ -n "$lib" ] && t=$(basename $lib .so) && lib=$REPO/$lib
  run ${t}_cfg2 "$lib" --steps 100 --warmup 5
  run ${t}_cfg2e16 "$lib" --steps 100 --warmup 5 --range-kernel e16
  run ${t}_cfg3 "$lib" --config cfg3 --steps 20 --warmup 3 --no-parity
  run ${t}_cfg2full "$lib" --chain full --batch 64 --steps 40 --warmup 3
  run ${t}_small "$lib" --config small --steps 40 --warmup 3
done
