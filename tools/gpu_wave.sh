#!/bin/bash
# one-wave range kernel: parity tests + headline bench per variant (run through gpurun)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
if [ "${SKIPTEST:-0}" = 0 ]; then python -m pytest tests/test_timed_kernels_gpu.py -m gpu -q -k "one_wave" > $OUT/pytest_wave.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_wave.log; tail -n 3 $OUT/pytest_wave.log; fi
run() { # tag, env lib, extra args
  if [ -n "$2" ]; then export BLAH2HIP_LIBRARY=$2; else unset BLAH2HIP_LIBRARY; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${@:3} > $OUT/bw_$1.log 2> $OUT/bw_$1.err
  python - <<PY
import json
try:
    j=json.loads(open("$OUT/bw_$1.log").read().strip().split("\n")[-1])
    k=j["roofline"]["kernel_us_per_step"]
    print("$1: %.0f CPIs/s range %.1f us/launch (%.2f us/CPI) frac %.3f parity %s"%(j["value"], k["range"], k["range"]/j["config"]["batch_cpis_per_step"], j["roofline"]["frac"], (j["parity"] or {}).get("pass")))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/bw_$1.err").read()[-600:])
PY
}
run base "" 

for f in tools/ab/lib_w*.so; do
  t=$(basename $f .so)
  run $t $REPO/$f --range-kernel wave; grep trace $OUT/bw_$t.err | tail -n 2
done
