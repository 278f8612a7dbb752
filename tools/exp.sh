P='import json,sys; d=json.loads(sys.stdin.readline()); print(d["config"]["batch_cpis_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["kernel_us_per_step"])'
for b in 16 32 33 63 64 95 127 128; do python bench.py --no-cpu-baseline --batch $b --steps 20 | python -c "$P"; done
