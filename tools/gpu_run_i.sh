#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
python -m pytest tests -m gpu -q -x > $OUT/pytest_i.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_i.log
tail -n 6 $OUT/pytest_i.log
show() { python - "$1" "$2" <<PY
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], "F=%d x%d"%(j["config"]["fft_len"], j["config"]["n_seg"]), "%.0f CPIs/s  %.2f us/CPI frac %.3f"%(j["value"], j["us_per_cpi"], j["roofline"]["frac"]), {k:round(v,1) for k,v in j["roofline"]["kernel_us_per_step"].items()}, "parity", (j.get("parity") or {}).get("pass"), (j.get("parity") or {}).get("peak_rel"))
except Exception as e:
    print(sys.argv[2], "ERR", open(sys.argv[1]).read()[-600:])
PY
}
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bi_cfg2.log 2>&1; show $OUT/bi_cfg2.log "cfg2"
python bench.py --fmt i16 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bi_cfg2_i16.log 2>&1; show $OUT/bi_cfg2_i16.log "cfg2 i16"
python bench.py --chain full --batch 64 --no-cpu-baseline > $OUT/bi_full.log 2>&1; show $OUT/bi_full.log "cfg2 full"
python bench.py --config cfg3 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bi_cfg3.log 2>&1; show $OUT/bi_cfg3.log "cfg3"
python bench.py --config cfg3 --chain full --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bi_cfg3_full.log 2>&1; show $OUT/bi_cfg3_full.log "cfg3 full"
python bench.py --config cfg5 --fmt f16 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bi_cfg5.log 2>&1; show $OUT/bi_cfg5.log "cfg5"
python bench.py --config small --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bi_small.log 2>&1; show $OUT/bi_small.log "small"
python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bi_b1.log 2>&1; show $OUT/bi_b1.log "cfg2 b1"
