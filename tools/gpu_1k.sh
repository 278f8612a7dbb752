#!/bin/bash
# same-box A/B: tree against the libraries of tools/ab (run through gpurun); arguments: bench arguments
P='import sys,json; j=json.loads(sys.stdin.read().strip().split("\n")[-1]); B=j["config"]["batch_cpis_per_step"]; print(sys.argv[1], round(j["value"]), {k:round(v/B,3) for k,v in j["roofline"]["kernel_us_per_step"].items()}, (j.get("parity") or {}).get("pass"))'
for rep in 1 2; do
for lib in "" $(ls tools/ab/*.so 2>/dev/null); do
  t=tree; if [ -n "$lib" ]; then t=$(basename $lib .so); export BLAH2HIP_LIBRARY=$PWD/$lib; else unset BLAH2HIP_LIBRARY; fi
  python bench.py --no-cpu-baseline "$@" 2>&1 | python -c "$P" $t
done
done
