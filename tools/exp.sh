P='import json,sys; d=json.loads(sys.stdin.readline()); print(d["config"]["batch_cpis_per_step"], d["config"]["fmt"], d["config"]["fft_len"], round(d["value"]), round(d["roofline"]["frac"],3), d["roofline"]["kernel_us_per_step"])'
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline | python -c "$P"
python bench.py --no-cpu-baseline --fmt i16 | python -c "$P"
BLAH2HIP_FFT_LEN=1024 python bench.py --no-cpu-baseline | python -c "$P"
BLAH2HIP_FFT_LEN=1024 BLAH2HIP_RANGE_BUF=0 python bench.py --no-cpu-baseline | python -c "$P"
python bench.py --no-cpu-baseline --config cfg3 | python -c "$P"
BLAH2HIP_RANGE_BUF=0 python bench.py --no-cpu-baseline --config cfg3 | python -c "$P"
