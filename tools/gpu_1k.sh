#!/bin/bash
# same-box A/B of the planner's choice for large launches at cfg 2 (F = 1024, one-wave kernel) against F = 2048 (run through gpurun)
P='import sys,json; j=json.loads(sys.stdin.read().strip().split("\n")[-1]); B=j["config"]["batch_cpis_per_step"]; print(sys.argv[1], round(j["value"]), j["roofline"]["kernel"], j["config"]["fft_len"], {k:round(v/B,3) for k,v in j["roofline"]["kernel_us_per_step"].items()}, (j.get("parity") or {}).get("pass"))'
for rep in 1 2; do
  for a in "" "--fmt i16" "--chain full --batch 64" "--batch 32" "--batch 256"; do
    python bench.py --no-cpu-baseline $a 2>&1 | python -c "$P" "plan[$a]"
    python bench.py --no-cpu-baseline --fft-len 2048 $a 2>&1 | python -c "$P" "2048[$a]"
  done
done
