#!/bin/bash
# GPU box: the cfg 3 full chain (batch 32) on the product library and on variant builds in tools/ab/, interleaved twice
cd "$(dirname "$0")/.."
show() { tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$1:', round(1e6 / r['value'], 2), 'us/CPI, parity', (r.get('parity') or {}).get('pass'), {k['kernel']: round(k['us_per_cpi'], 2) for k in r['roofline']['kernels'] if k['us_per_cpi'] > 0.5})"; }
for rep in 1 2; do
for lib in "" $@; do
  name=${lib:-product}
  if [ -z "$lib" ]; then unset BLAH2HIP_LIBRARY; else export BLAH2HIP_LIBRARY=$PWD/tools/ab/$lib; fi
  timeout 400 python bench.py --config cfg3 --chain full --batch 32 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | show "cfg3 full b32 $name"
done
done
