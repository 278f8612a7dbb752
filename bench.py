#!/usr/bin/env python3
"""Benchmark of the cross-ambiguity hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the device-resident chain (range kernel -> Doppler
kernel -> metrics) over one batch of --batch synthetic CPIs already resident in
HBM.  Workload at N=1: BASELINE.json configs[1] (2 MS/s, 1 s CPI, +-256 Hz ->
513 Doppler bins x 411 delay bins, complex fp32 IQ).  CPIs shard one stream per
GPU with no data-path collective (weak scaling); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

CONFIGS = {
    # name: (delayMin, delayMax, dopplerMin, dopplerMax, fs, n)
    "cfg2": (-10, 400, -256, 256, 2_000_000, 2_000_000),
    "test": (-10, 300, -300, 300, 2_000_000, 1_000_000),
    "cfg3": (-24, 2023, -512, 512, 10_000_000, 10_000_000),
    "cfg5": (-10, 400, -512, 512, 20_000_000, 40_000_000),  # 2 s CPI -> 2049 Doppler bins; use --fmt f16
}


def synth_batch(torch, n_cpi, n, seed, fs, device):
    """Seeded synthetic IQ, int16-valued like the .rspduo wire format, as complex64 planes."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = 300.0 * torch.randn((n_cpi, n, 2), generator=g, device=device)
    noise = 30.0 * torch.randn((n_cpi, n, 2), generator=g, device=device)
    xc = torch.view_as_complex(x)
    t = torch.arange(n, device=device, dtype=torch.float64) / fs
    d, f, a = 37, -63.0, 0.05
    ph = torch.exp(2j * torch.pi * f * t).to(torch.complex64)
    xd = torch.roll(xc, d, dims=1)
    xd[:, :d] = 0
    yc = 0.8 * xc + a * xd * ph + torch.view_as_complex(noise)
    q = lambda v: torch.view_as_complex(torch.clamp(torch.round(torch.view_as_real(v)), -32768, 32767).contiguous())
    return q(xc).contiguous(), q(yc).contiguous()


def cpu_baseline(args_amb, budget_s=20.0):
    """The reference's own Ambiguity::process + set_metrics (oracle/_ref, shim FFT)
    or, if that library is absent, the NumPy restatement -- timed on the host cores."""
    import numpy as np
    from oracle import blah2_oracle as O
    from oracle import ref_lib as R
    dmin, dmax, fmin, fmax, fs, n = args_amb
    x, y = O.synth_iq(n, fs=fs)
    times = []
    if R.available():
        a = R.RefAmbiguity(dmin, dmax, fmin, fmax, fs, n, True)
        t_end = time.time() + budget_s
        while time.time() < t_end and len(times) < 12:
            a.process(x, y)
            times.append(a.last_seconds)
        kind = "reference"
        what = ("reference Ambiguity.cpp + Map::set_metrics compiled from /root/reference/src with the "
                "fp64 shim FFT (FFTW is not installed in this image), 1 thread")
    else:
        d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
        t_end = time.time() + budget_s
        while time.time() < t_end and len(times) < 12:
            t0 = time.time()
            m = O.ambiguity_process(d, x, y)
            O.map_metrics(m)
            times.append(time.time() - t0)
        kind = "port"
        what = "NumPy/pocketfft fp64 restatement (oracle/blah2_oracle.py), 1 thread"
    med = float(np.median(times))
    res = {"value": 1.0 / med, "unit": "CPIs/s", "cores": 1, "kind": kind,
           "sample": f"{len(times)} CPIs of the same workload, median {med*1e3:.0f} ms/CPI; {what}",
           "host_cores_available": os.cpu_count()}
    if kind == "reference":
        # beside it: the NumPy/pocketfft fp64 restatement of the same algorithm
        d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
        tp = []
        for _ in range(3):
            t0 = time.time()
            O.map_metrics(O.ambiguity_process(d, x, y))
            tp.append(time.time() - t0)
        res["port_value"] = 1.0 / float(np.median(tp))
        res["port_sample"] = f"3 CPIs, median {np.median(tp)*1e3:.0f} ms/CPI, NumPy/pocketfft fp64 restatement (oracle/blah2_oracle.py), 1 thread"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0,
                    help="CPIs per step (per GPU); default 128 for the 2 MS/s configs (4 GB of IQ per step: a pulse is "
                         "the scheduling unit of the range kernel and 128 x 513 pulses leave a 1.5 %% tail on 1024 "
                         "resident workgroups, 32 x 513 leave 6 %%), 8 for cfg3 (64 with --chain full: the Toeplitz solve costs "
                         "a fixed ~8 ms per launch there)")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--fmt", default="c32", choices=["c32", "i16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain", default="amb", choices=["amb", "full"],
                    help="amb: range+Doppler+metrics (BASELINE headline); full: clutter filter + amb + CFAR (configs[2])")
    ap.add_argument("--cfar", default="2d", choices=["1d", "2d"])
    ap.add_argument("--n-doppler", type=int, default=0,
                    help="explicit number of Doppler bins (extension; 0 = the reference constructor's rule, which gives "
                         "513 at the headline configuration; 512 gives the literal BASELINE wording)")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent CPI streams per GPU (engine handles on their own HIP streams); successive "
                         "steps alternate between them so one batch's Doppler stage overlaps the next batch's range stage")
    a = ap.parse_args()

    import torch
    import blah2_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):  # under torchrun, also at N = 1
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)

    cfg = CONFIGS[a.config]
    dmin, dmax, fmin, fmax, fs, n = cfg
    B = a.batch if a.batch > 0 else ({"cfg3": 64 if a.chain == "full" else 8, "cfg5": 4}.get(a.config, 128))
    NS = max(1, a.streams) if a.chain == "amb" else 1
    ambs = [blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, device=local, max_batch=B, n_doppler_bins=a.n_doppler)
            for _ in range(NS)]
    amb = ambs[0]
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    cells = nD * nC
    s_in = 8 if a.fmt == "c32" else 4
    # ring of distinct batches > 512 MB so every CPI read is real HBM traffic
    # (the 256 MB Infinity Cache would otherwise hold a 32 MB CPI)
    bytes_per_batch = 2 * n * s_in * B
    ring = max(2, -(-int(600e6) // bytes_per_batch))
    xs, ys, iqs = [], [], []
    for r in range(ring):
        parts = [synth_batch(torch, min(16, B - c0), n, 1000 + 17 * rank + 131 * r + c0, fs, dev) for c0 in range(0, B, 16)]
        x = torch.cat([p_[0] for p_ in parts])
        y = torch.cat([p_[1] for p_ in parts])
        del parts
        if a.fmt == "c32":
            xs.append(x)
            ys.append(y)
        elif a.fmt == "f16":  # half-precision storage of the int16-valued IQ (exact up to 2048, rounded above)
            xs.append(torch.view_as_real(x).to(torch.float16).contiguous())
            ys.append(torch.view_as_real(y).to(torch.float16).contiguous())
        else:
            iq = torch.stack([x.real, x.imag, y.real, y.imag], dim=-1).to(torch.int16).contiguous()
            iqs.append(iq)
    wh = None
    if a.chain == "full":
        if a.fmt != "c32":
            raise SystemExit("--chain full needs --fmt c32")
        wh = blah2_amd.WienerHopf(dmin, dmax, n, device=local, max_batch=B)  # config.yml uses the same lag window
        yfilt = torch.empty((B, n), dtype=torch.complex64, device=dev)
        okflag = torch.zeros(B, dtype=torch.int32, device=dev)
        hits = torch.zeros((B, 65536, 2), dtype=torch.float64, device=dev)  # 16-byte records
        hitcnt = torch.zeros(B, dtype=torch.int32, device=dev)
        L = blah2_amd.load()
    outs = [torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev) for _ in range(NS)]
    mets = [torch.zeros((B, 2), dtype=torch.float64, device=dev) for _ in range(NS)]
    out, met = outs[0], mets[0]
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    side = [torch.cuda.Stream(device=dev) for _ in range(NS)] if NS > 1 else [stream]
    sts = [s_.cuda_stream for s_ in side]

    def step(i):
        r = i % ring
        if wh is not None:
            wh.process_dev(xs[r].data_ptr(), ys[r].data_ptr(), B, n, yfilt.data_ptr(), okflag.data_ptr(), st)
            amb.process_dev(blah2_amd.FMT_C32, xs[r].data_ptr(), yfilt.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
            if a.cfar == "2d":
                blah2_amd._lib.check(L.blah2hip_cfar2d_dev(amb._h, out.data_ptr(), met.data_ptr(), B, 1e-5, 2, 6, 1, 3, 5, 15.0,
                                                           hits.data_ptr(), 65536, hitcnt.data_ptr(), st))
            else:
                blah2_amd._lib.check(L.blah2hip_cfar1d_dev(amb._h, out.data_ptr(), met.data_ptr(), B, 1e-5, 2, 6, 5, 15.0,
                                                           hits.data_ptr(), 65536, hitcnt.data_ptr(), st))
        elif a.fmt in ("c32", "f16"):
            q = i % NS
            ambs[q].process_dev(blah2_amd.FMT_C32 if a.fmt == "c32" else blah2_amd.FMT_F16, xs[r].data_ptr(), ys[r].data_ptr(),
                                B, n, outs[q].data_ptr(), mets[q].data_ptr(), sts[q])
        else:
            q = i % NS
            ambs[q].process_dev(blah2_amd.FMT_I16, iqs[r].data_ptr(), 0, B, n, outs[q].data_ptr(), mets[q].data_ptr(), sts[q])

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # second, identical region with every kernel bracketed by HIP events on the
    # launch stream: per-kernel durations for the roofline line
    for h_ in ambs:
        h_.set_timing(True)
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    kt = {}
    for h_ in ambs:
        for k_, (ms_, n_) in h_.get_timing().items():
            kt[k_] = (kt.get(k_, (0.0, 0))[0] + ms_, kt.get(k_, (0.0, 0))[1] + n_)
        h_.set_timing(False)
    range_ms, range_n = kt["range"]
    avg_range_s = (range_ms / max(range_n, 1)) * 1e-3
    # algorithmic bytes per launch of the range kernel: every input
    # sample once + the nD x nDelay complex fp32 range map once, x batch
    algo_bytes = (2 * n * s_in + cells * 8) * B
    achieved = algo_bytes / avg_range_s / 1e9 if avg_range_s > 0 else 0.0
    chain_s = sum(v[0] for v in kt.values()) * 1e-3 / max(range_n, 1)

    # HBM bytes per launch of the range kernel from the committed rocprofv3 PMC
    # passes of this same command (tools/summarize_prof.py -> profiles/*_traffic.json)
    traffic, traffic_src = None, None
    import glob
    for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            tj = json.load(open(pth))
            bc = tj.get("bench_config", {})
            if bc and (bc.get("config"), bc.get("batch"), bc.get("fmt")) != (a.config, B, a.fmt):
                continue
            traffic = tj["kernels"]["range_kernel"]["hbm_bytes"]
            traffic_src = os.path.relpath(pth, ROOT)
            break
        except Exception:
            continue

    # device-copy ceiling beside the 8 TB/s spec figure: a 1 GiB
    # device-to-device copy, read + written bytes per second
    src_ = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst_ = torch.empty_like(src_)
    for _ in range(3):
        dst_.copy_(src_)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst_.copy_(src_)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 10 * 2 * src_.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src_, dst_
    # reference-formulation flops per CPI (5 n log2 n per FFT, 3 nfft-point FFTs
    # per pulse + one nD-point FFT per delay column) -- what the CPU path would execute
    import math
    nfft_ref = amb.get_nfft()
    ref_flops = 3 * nD * 5 * nfft_ref * math.log2(nfft_ref) + nC * 5 * nD * math.log2(nD)

    # sanity: the timed outputs are real (metrics of the last batch are finite, target visible)
    mt = met.cpu().numpy()
    ok = bool((mt == mt).all() and (mt[:, 1] > 0).all())

    if rank == 0:
        total_cpis = world * B * a.steps
        res = {
            "metric": "CPIs/s", "value": total_cpis / elapsed, "unit": "CPIs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[1]: 2 MS/s, 1 s CPI, +-256 Hz -> {nD} Doppler x {nC} delay bins, "
                                    f"synthetic {a.fmt} IQ resident in HBM" if a.config == "cfg2" else a.config)
                       + ("" if a.chain == "amb" else f" + clutter filter + {a.cfar} CA-CFAR"),
                       "chain": a.chain,
                       "batch_cpis_per_step": B, "fmt": a.fmt, "n_samples": n, "fs": fs,
                       "n_doppler_bins": nD, "n_delay_bins": nC, "n_corr": amb.get_n_corr(),
                       "fft_len": amb.dims.fft_len, "n_seg": amb.dims.n_seg, "seg_len": amb.dims.seg_len,
                       "ring_batches": ring, "streams_per_gpu": NS, "sharding": f"{world} independent CPI streams, one per GPU"},
            "cells_per_s": total_cpis * cells / elapsed,
            "us_per_cpi": elapsed / (B * a.steps) * 1e6,
            "outputs_valid": ok,
            "roofline": {"bound": "hbm", "kernel": "range_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "copy_ceiling": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs,
                         "chain_achieved": (2 * n * s_in + cells * 8) * total_cpis / world / elapsed / 1e9,
                         "ref_equivalent_tflops": ref_flops * total_cpis / world / elapsed / 1e12,
                         "avg_launch_us": avg_range_s * 1e6, "launches_timed": range_n,
                         "kernel_us_per_step": {k: v[0] / max(v[1], 1) * 1e3 for k, v in kt.items() if v[1]},
                         "chain_us_per_step": chain_s * 1e6},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
