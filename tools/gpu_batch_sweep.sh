#!/bin/bash
# batch-size sweep of the ambiguity chain at cfg3 / cfg5 (tail of the last round of pulses)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
show() { python - "$1" "$2" <<PY
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], "%.0f CPIs/s  %.2f us/CPI"%(j["value"], j["us_per_cpi"]), {k["kernel"]:round(k["us_per_cpi"],2) for k in j["roofline"]["kernels"]}, "parity", (j.get("parity") or {}).get("pass"))
except Exception as e:
    print(sys.argv[2], "ERR", open(sys.argv[1]).read()[-400:])
PY
}
for B in 4 8 16; do python bench.py --config cfg5 --fmt f16 --batch $B --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bs_cfg5_$B.log 2>&1; show $OUT/bs_cfg5_$B.log "cfg5 f16 batch $B"; done
for B in 8 16 32; do python bench.py --config cfg3 --batch $B --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bs_cfg3_$B.log 2>&1; show $OUT/bs_cfg3_$B.log "cfg3 batch $B"; done
