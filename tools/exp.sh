P='import json,sys; d=json.loads(sys.stdin.readline()); print(d["config"]["batch_cpis_per_step"], d["config"]["fmt"], d["config"]["fft_len"], round(d["value"]), round(d["roofline"]["frac"],3), d["roofline"]["kernel_us_per_step"])'
python bench.py --no-cpu-baseline | python -c "$P"
BLAH2HIP_RANGE_DEFER=1 python bench.py --no-cpu-baseline | python -c "$P"
python bench.py --no-cpu-baseline | python -c "$P"
BLAH2HIP_RANGE_DEFER=1 python bench.py --no-cpu-baseline | python -c "$P"
BLAH2HIP_RANGE_DEFER=1 timeout 900 python -m pytest tests/test_ambiguity_gpu.py tests/test_edge_cases_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
