#!/usr/bin/env python3
"""configs[2] full chain: what would the bytes be worth?  (the review's "one pass fewer".)

The clutter filter's correlations, its FIR and the range kernel each read x and y from HBM, and the filtered channel
is written and read again: 3.6 x B_amb per CPI.  Fusing the FIR into the range kernel's loads would remove the filtered
plane's write and re-read and one read of x.  This tool measures an UPPER bound of what removing bytes can buy without
writing that kernel: the same launches with every CPI of the batch at the SAME addresses (cpi stride 0: one 160 MB CPI,
which the 256 MB Infinity Cache holds, read 32 times; the filtered plane written 32 times over itself).  The transforms,
barriers and instruction streams are identical -- only the DRAM traffic is gone.  Per-kernel HIP-event times beside the
normal run on the same box.

The product library rejects a CPI stride of 0 (blah2hip_amb_process_dev: "cpi_stride < samples used per CPI"); the
check is compiled out only under -DB2_EXPERIMENT_ALIASED_CPIS.  Build that variant beside the product library and select it:

    bash tools/build_variant.sh aliased -DB2_EXPERIMENT_ALIASED_CPIS
    gpurun -- 'BLAH2HIP_LIBRARY=$PWD/tools/ab/libblah2hip_aliased.so python tools/gpu_cfg3_bytes.py --json gpurun_out/cfg3_bytes.json'
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    import torch

    import blah2_amd as b2
    if "aliased" not in os.path.basename(os.environ.get("BLAH2HIP_LIBRARY", "")):
        raise SystemExit("tools/gpu_cfg3_bytes.py needs the -DB2_EXPERIMENT_ALIASED_CPIS build of the library: "
                         "bash tools/build_variant.sh aliased -DB2_EXPERIMENT_ALIASED_CPIS, then "
                         "BLAH2HIP_LIBRARY=$PWD/tools/ab/libblah2hip_aliased.so (the product library refuses a CPI stride of 0)")
    dmin, dmax, fmin, fmax, fs, n = (-24, 2023, -512, 512, 10_000_000, 10_000_000)
    B = a.batch
    dev = torch.device("cuda", 0)
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
    wh = b2.WienerHopf(dmin, dmax, n, max_batch=B)
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    ring = 2
    xs = [torch.view_as_complex(300 * torch.randn((B, n, 2), generator=g, device=dev)) for _ in range(ring)]
    ys = [0.8 * x + torch.view_as_complex(30 * torch.randn((B, n, 2), generator=g, device=dev)) for x in xs]
    yf = torch.empty((B, n), dtype=torch.complex64, device=dev)
    ok = torch.zeros(B, dtype=torch.int32, device=dev)
    out = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
    met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(stride, steps):
        for h in (amb, wh):
            h.set_timing(False)
        for i in range(3):
            step(i, stride)
        torch.cuda.synchronize()
        for h in (amb, wh):
            h.set_timing(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            step(i, stride)
        e1.record()
        torch.cuda.synchronize()
        kt = {}
        for h in (amb, wh):
            for k, (ms, cnt) in h.get_timing().items():
                if cnt:
                    kt[k] = ms / cnt * 1e3 / B  # us per CPI
            h.set_timing(False)
        kt["whole_chain"] = e0.elapsed_time(e1) / steps * 1e3 / B
        return kt

    def step(i, stride):
        r = i % ring
        wh.process_dev_fmt(b2.FMT_C32, xs[r].data_ptr(), ys[r].data_ptr(), B, stride, yf.data_ptr(), stride, ok.data_ptr(), st)
        amb.process_dev(b2.FMT_C32, xs[r].data_ptr(), yf.data_ptr(), B, stride, out.data_ptr(), met.data_ptr(), st)

    res = {"config": "BASELINE configs[2] (10 MS/s, 1 s CPI, 1025 x 2048, 2047 taps), clutter filter + ambiguity, batch %d" % B,
           "normal_us_per_cpi": run(n, a.steps), "same_addresses_us_per_cpi": run(0, a.steps), "normal_again_us_per_cpi": run(n, a.steps),
           "note": "same_addresses: every CPI of a batch reads the same x / y and writes the same filtered plane (cpi stride 0): "
                   "identical launches and instruction streams, DRAM traffic replaced by Infinity Cache hits"}
    nk, sk = res["normal_us_per_cpi"], res["same_addresses_us_per_cpi"]
    res["ratio_same_over_normal"] = {k: sk[k] / nk[k] for k in nk if k in sk and nk[k] > 0}
    print(json.dumps(res, indent=1))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
