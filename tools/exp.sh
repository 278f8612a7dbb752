P='import json,sys; d=json.loads(sys.stdin.readline()); print(d["config"]["batch_cpis_per_step"], d["config"]["streams_per_gpu"], round(d["value"]), round(d["roofline"]["frac"],3), d["roofline"]["kernel_us_per_step"])'
for s in 1 2 3; do python bench.py --no-cpu-baseline --streams $s | python -c "$P"; done
for s in 2 4; do python bench.py --no-cpu-baseline --streams $s --batch 64 | python -c "$P"; done
