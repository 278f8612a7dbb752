P='import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["config"]["fft_len"], d["config"]["n_seg"], d["roofline"]["kernel_us_per_step"])'
python bench.py --no-cpu-baseline | python -c "$P"
BLAH2HIP_FFT_LEN=1024 python bench.py --no-cpu-baseline | python -c "$P"
BLAH2HIP_FFT_LEN=4096 python bench.py --no-cpu-baseline | python -c "$P"
BLAH2HIP_RANGE_E8=1 python bench.py --no-cpu-baseline | python -c "$P"
