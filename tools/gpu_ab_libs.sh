#!/bin/bash
# GPU box: the bench lines of the product library beside variant builds in tools/ab/ (BLAH2HIP_LIBRARY), interleaved
cd "$(dirname "$0")/.."
show() { tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$1:', round(r['value'], 1), 'CPIs/s,', round(1e6 / r['value'], 2), 'us/CPI, parity', (r.get('parity') or {}).get('pass'), {k['kernel']: round(k['us_per_cpi'], 2) for k in r['roofline']['kernels'] if k['us_per_cpi'] > 0.2})"; }
for rep in 1 2; do
for lib in "" $@; do
  name=${lib:-product}
  export BLAH2HIP_LIBRARY=${lib:+$PWD/tools/ab/$lib}
  [ -z "$lib" ] && unset BLAH2HIP_LIBRARY
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | show "headline $name"
done
done
for lib in "" $@; do
  name=${lib:-product}
  export BLAH2HIP_LIBRARY=${lib:+$PWD/tools/ab/$lib}
  [ -z "$lib" ] && unset BLAH2HIP_LIBRARY
  timeout 300 python bench.py --chain full --steps 12 --no-cpu-baseline 2>&1 | show "cfg2 full $name"
  timeout 300 python bench.py --config cfg3 --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | show "cfg3 $name"
  timeout 300 python bench.py --config cfg5 --fmt f16 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | show "cfg5 $name"
  timeout 400 python bench.py --config cfg3 --chain full --batch 32 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | show "cfg3 full b32 $name"
done
