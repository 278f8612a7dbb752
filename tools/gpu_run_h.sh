#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
show() { python - "$1" "$2" <<PY
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], "F=%d x%d segs"%(j["config"]["fft_len"], j["config"]["n_seg"]), "%.0f CPIs/s  %.2f us/CPI"%(j["value"], j["us_per_cpi"]), {k:round(v,1) for k,v in j["roofline"]["kernel_us_per_step"].items()}, "parity", (j.get("parity") or {}).get("pass"))
except Exception as e:
    print(sys.argv[2], "ERR", open(sys.argv[1]).read()[-400:])
PY
}
python -m pytest tests/test_timed_kernels_gpu.py -m gpu -q -x > $OUT/pytest_h.log 2>&1; tail -n 3 $OUT/pytest_h.log
for cfg in yml test cfg2; do for F in 1024 2048; do
  BLAH2HIP_FFT_LEN=$F python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bh_${cfg}_$F.log 2>&1; show $OUT/bh_${cfg}_$F.log "$cfg forced F=$F"
done; done
