#!/bin/bash
# GPU box: the full chain at cfg 3 with one and two batches in flight (bench.py --streams), batch 32 and 256
cd "$(dirname "$0")/.."
show() { tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); B = r['config'].get('batch_per_gpu') or r['config'].get('batch') or 1
print('$1:', round(r['value'], 1), 'CPIs/s,', round(1e6 / r['value'], 1), 'us/CPI, parity', (r.get('parity') or {}).get('pass'))"; }
for s in 1 2; do
  timeout 400 python bench.py --config cfg3 --chain full --batch 32 --steps 8 --warmup 2 --no-cpu-baseline --streams $s 2>&1 | show "cfg3 full b32 streams $s"
done
timeout 600 python bench.py --config cfg3 --chain full --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | show "cfg3 full b256 streams 1"
timeout 300 python bench.py --chain full --steps 20 --no-cpu-baseline 2>&1 | show "cfg2 full b256 streams 1"
timeout 300 python bench.py --chain full --batch 64 --steps 40 --no-cpu-baseline --streams 2 2>&1 | show "cfg2 full b64 streams 2"
timeout 300 python bench.py --chain full --batch 64 --steps 40 --no-cpu-baseline --streams 1 2>&1 | show "cfg2 full b64 streams 1"
