#!/bin/bash
# A/B of the range kernels (one-wave 2048 / 1024), tree against the libraries of tools/ab (run through gpurun)
P='import sys,json; j=json.loads(sys.stdin.read().strip().split("\n")[-1]); B=j["config"]["batch_cpis_per_step"]; print(sys.argv[1], round(j["value"]), j["roofline"]["kernel"], j["config"]["seg_len"], {k:round(v/B,3) for k,v in j["roofline"]["kernel_us_per_step"].items()}, (j.get("parity") or {}).get("pass"))'
for rep in 1 2; do
for lib in "" $(ls tools/ab/*.so 2>/dev/null); do
  t=tree; if [ -n "$lib" ]; then t=$(basename $lib .so); export BLAH2HIP_LIBRARY=$PWD/$lib; else unset BLAH2HIP_LIBRARY; fi
  python bench.py --no-cpu-baseline 2>&1 | python -c "$P" $t
  for b in 96 128; do
  python bench.py --fft-len 1024 --range-kernel wave1k --no-cpu-baseline --batch $b 2>&1 | python -c "$P" $t-1k-b$b
  done
  python bench.py --fft-len 1024 --range-kernel wave1k --no-cpu-baseline --fmt i16 2>&1 | python -c "$P" $t-1k-i16
done
done
