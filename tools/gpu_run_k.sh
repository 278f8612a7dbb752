#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
show() { python - "$1" "$2" <<PY
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], "%.0f CPIs/s  %.2f us/CPI frac %.3f"%(j["value"], j["us_per_cpi"], j["roofline"]["frac"]), {k:round(v,1) for k,v in j["roofline"]["kernel_us_per_step"].items()}, "parity", (j.get("parity") or {}).get("pass"), (j.get("parity") or {}).get("peak_rel"))
except Exception as e:
    print(sys.argv[2], "ERR", open(sys.argv[1]).read()[-600:])
PY
}
for v in 0 1 0 1; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --range-variant $v > $OUT/bk_cfg2_$v.log 2>&1; show $OUT/bk_cfg2_$v.log "cfg2 prefetch $v"
done
for v in 0 1; do
python bench.py --fmt i16 --steps 40 --warmup 5 --no-cpu-baseline --range-variant $v > $OUT/bk_i16_$v.log 2>&1; show $OUT/bk_i16_$v.log "cfg2 i16 prefetch $v"
python bench.py --config cfg5 --fmt f16 --steps 10 --warmup 2 --no-cpu-baseline --range-variant $v > $OUT/bk_cfg5_$v.log 2>&1; show $OUT/bk_cfg5_$v.log "cfg5 prefetch $v"
python bench.py --config cfg3 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --range-variant $v > $OUT/bk_cfg3_$v.log 2>&1; show $OUT/bk_cfg3_$v.log "cfg3 prefetch $v"
done
