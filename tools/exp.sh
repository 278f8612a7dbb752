P='import json,sys; d=json.loads(sys.stdin.readline()); print(d["config"]["batch_cpis_per_step"], d["config"]["chain"], round(d["value"]), round(d["us_per_cpi"],1), d["roofline"]["kernel_us_per_step"])'
timeout 600 python -m pytest tests/test_clutter_gpu.py tests/test_baseline_configs_gpu.py tests/test_host_cpp_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --chain full --batch 16 | python -c "$P"
python bench.py --no-cpu-baseline --chain full --batch 64 | python -c "$P"
