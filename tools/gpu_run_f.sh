#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_timed_kernels_gpu.py tests/test_ambiguity_gpu.py -m gpu -q -x > $OUT/pytest_f.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_f.log
tail -n 6 $OUT/pytest_f.log
show() { python - "$1" "$2" <<PY
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], "%.0f CPIs/s  %.2f us/CPI"%(j["value"], j["us_per_cpi"]), {k:round(v,1) for k,v in j["roofline"]["kernel_us_per_step"].items()}, "parity", (j.get("parity") or {}).get("pass"))
except Exception as e:
    print(sys.argv[2], "ERR", open(sys.argv[1]).read()[-400:])
PY
}
for rk in e16 e8; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --range-kernel $rk > $OUT/bf_cfg2_$rk.log 2>&1; show $OUT/bf_cfg2_$rk.log "cfg2 $rk"
  python bench.py --config cfg3 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --range-kernel $rk > $OUT/bf_cfg3_$rk.log 2>&1; show $OUT/bf_cfg3_$rk.log "cfg3 $rk"
  python bench.py --config cfg5 --fmt f16 --steps 10 --warmup 2 --no-cpu-baseline --range-kernel $rk > $OUT/bf_cfg5_$rk.log 2>&1; show $OUT/bf_cfg5_$rk.log "cfg5 $rk"
done
python bench.py --config small --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bf_small.log 2>&1; show $OUT/bf_small.log "small e8"
