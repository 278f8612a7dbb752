#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
show() { python - "$1" "$2" <<PY
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], "%.0f CPIs/s  %.2f us/CPI"%(j["value"], j["us_per_cpi"]), {k:round(v,1) for k,v in j["roofline"]["kernel_us_per_step"].items()}, "parity", (j.get("parity") or {}).get("pass"))
except Exception as e:
    print(sys.argv[2], "ERR", open(sys.argv[1]).read()[-400:])
PY
}
export BLAH2HIP_LIBRARY=$REPO/tools/ab/libblah2hip_r8w3.so
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --range-kernel e8 > $OUT/bg_cfg2.log 2>&1; show $OUT/bg_cfg2.log "cfg2 e8 w3"
python bench.py --config cfg3 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --range-kernel e8 > $OUT/bg_cfg3.log 2>&1; show $OUT/bg_cfg3.log "cfg3 e8 w3"
python bench.py --config cfg5 --fmt f16 --steps 10 --warmup 2 --no-cpu-baseline --range-kernel e8 > $OUT/bg_cfg5.log 2>&1; show $OUT/bg_cfg5.log "cfg5 e8 w3"
python bench.py --config small --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bg_small.log 2>&1; show $OUT/bg_small.log "small e8 w3"
unset BLAH2HIP_LIBRARY
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --range-kernel e16 > $OUT/bg_cfg2_e16.log 2>&1; show $OUT/bg_cfg2_e16.log "cfg2 e16"
python bench.py --config small --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bg_small4.log 2>&1; show $OUT/bg_small4.log "small e8 w4"
