#!/usr/bin/env python3
"""Benchmark of the cross-ambiguity hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the device-resident chain (range kernel -> Doppler
kernel -> metrics; with --chain full: clutter filter in front, CFAR behind) over
one batch of --batch synthetic CPIs already resident in HBM.  Workload at N=1:
BASELINE.json configs[1] (2 MS/s, 1 s CPI, +-256 Hz -> 513 Doppler bins x 411
delay bins, complex fp32 IQ).  CPIs shard one stream per GPU with no data-path
collective (weak scaling); rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no torchrun environment re-executes this script under
torch.distributed.run with N ranks (one per GPU, RCCL); it refuses to run when
fewer than N devices are visible, and a torchrun launch whose WORLD_SIZE differs
from --gpus is an error, so `n_gpus` in the output is always the rank count that ran.

After the timed region the last batch's first and last CPI are compared with the
fp64 NumPy oracle (checker only, never timed); a violation exits non-zero.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak

CONFIGS = {
    # name: ((delayMin, delayMax, dopplerMin, dopplerMax, fs, n), description)
    "cfg2": ((-10, 400, -256, 256, 2_000_000, 2_000_000), "BASELINE configs[1]: 2 MS/s, 1 s CPI, +-256 Hz"),
    "test": ((-10, 300, -300, 300, 2_000_000, 1_000_000), "TestAmbiguity.cpp geometry: 2 MS/s, 0.5 s CPI, +-300 Hz"),
    "small": ((-10, 100, -100, 100, 1_000_000, 100_000), "tests/golden `medium` geometry: 1 MS/s, 0.1 s CPI, +-100 Hz (F = 1024 range kernel)"),
    "yml": ((-10, 400, -200, 200, 2_000_000, 1_500_000), "config/config.yml defaults: 2 MS/s, 0.75 s CPI, +-200 Hz"),
    "cfg3": ((-24, 2023, -512, 512, 10_000_000, 10_000_000), "BASELINE configs[2]: 10 MS/s, 1 s CPI, +-512 Hz"),
    "cfg5": ((-10, 400, -512, 512, 20_000_000, 40_000_000), "BASELINE configs[4]: 20 MS/s, 2 s CPI, +-512 Hz"),  # use --fmt f16
}

# parity gates of the in-bench check (the same numbers as tests/test_ambiguity_gpu.py and
# tests/test_full_chain_gpu.py, where they are derived)
GATE_PEAK_REL, GATE_DB_MAP, GATE_METRICS_DB, GATE_CHAIN_DIRECT = 1e-5, 0.005, 1e-3, 1e-4
# The JSON-map gate, as tests/gates.py states it: 0.005 dB on every cell down to DB_FLOOR dB below the map's mean
# level (the one consumer of the document, html/js/plot_map.js:170, clamps at the mean level itself); deeper cells --
# the few deep nulls of a Rayleigh floor -- are held to the absolute error 0.005 dB means AT that line.  ONE threshold.
DB_FLOOR = 20.0


# ----------------------------------------------------------------------------- launcher
def plan_launch(gpus, env, n_devices):
    """What `bench.py --gpus N` does.  Returns (action, detail):
    "inline"  run in this process (N == 1 without torchrun, or a torchrun rank whose WORLD_SIZE == N)
    "spawn"   re-execute under torch.distributed.run with N ranks (detail = nproc)
    "error"   refuse (detail = message)."""
    if gpus < 1:
        return "error", f"--gpus {gpus}: need at least one GPU"
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            return "error", f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks"
        if int(env.get("LOCAL_RANK", "0")) >= n_devices:
            return "error", f"{world} ranks requested, {n_devices} device(s) visible"
        return "inline", world
    if n_devices < gpus:
        return "error", f"{gpus} ranks requested, {n_devices} device(s) visible"
    if gpus == 1:
        return "inline", 1
    return "spawn", gpus


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(nproc, argv):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------- algorithmic bytes
def algorithmic_bytes(n, s_in, cells, B, cfar="2d"):
    """ALGORITHMIC bytes per launch of B CPIs: what a kernel must move
    if every input is read once and every output written once.  s_in = bytes per input sample per channel."""
    return {
        "range": (2 * n * s_in + cells * 8) * B,          # every input sample once + the range map once
        "doppler": 2 * cells * 8 * B,                     # range map in, final map out
        "clutter_corr": 2 * n * 8 * B,                    # x and y once
        "clutter_fir": 3 * n * 8 * B,                     # x, y in; filtered y out
        "cfar": cells * 8 * B,                            # the map once (1-D, and the fused 2-D detector)
        "sat_rows": 2 * cells * 8 * B,                    # 2-D detector, large windows: map in, fp64 row prefixes out
        "sat_cols": 2 * cells * 8 * B,                    # fp64 table in and out
        "rotate": 3 * n * 8 * B,
    }


# ----------------------------------------------------------------------------- data
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


# ----------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args_amb, budget_s=20.0):
    """The reference's own Ambiguity::process + set_metrics (oracle/_ref, shim FFT)
    or, if that library is absent, the NumPy restatement -- timed on the host cores."""
    import numpy as np
    from oracle import blah2_oracle as O
    from oracle import ref_lib as R
    dmin, dmax, fmin, fmax, fs, n = args_amb
    x, y = O.synth_iq(n, fs=fs)
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)

    def time_port(workers, reps):
        tp = []
        for _ in range(reps):
            t0 = time.time()
            O.map_metrics(O.ambiguity_process(d, x, y, workers=workers))
            tp.append(time.time() - t0)
        return float(np.median(tp))

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
        t_end = time.time() + budget_s
        while time.time() < t_end and len(times) < 12:
            times.append(time_port(None, 1))
        kind = "port"
        what = "NumPy/pocketfft fp64 restatement (oracle/blah2_oracle.py), 1 thread"
    med = float(np.median(times))
    # The line leads with the closest stand-in for what blah2 runs -- FFTW planned with 4 threads (blah2.cpp:115-120): the
    # NumPy/pocketfft fp64 restatement of the same algorithm with 4 workers over the batch of pulses.  The reference's own
    # sources compiled against the fp64 shim FFT (FFTW is not installable in this image; the shim is ~5x slower than
    # pocketfft) and the 1-thread restatement stand beside it.  None of them is credit: the roofline fraction is.
    p1, p4 = time_port(None, 3), time_port(4, 3)
    res = {"value": 1.0 / p4, "unit": "CPIs/s", "cores": 4, "kind": "port",
           "sample": f"3 CPIs of the same workload, median {p4*1e3:.0f} ms/CPI; NumPy/pocketfft fp64 restatement "
                     "(oracle/blah2_oracle.py), 4 pocketfft workers over the batch of pulses -- the stand-in for "
                     "fftw_plan_with_nthreads(4), blah2.cpp:120",
           "host_cores_available": os.cpu_count(),
           "threads_1": {"value": 1.0 / p1, "ms_per_cpi": p1 * 1e3, "unit": "CPIs/s"},
           ("reference_source" if kind == "reference" else "port_loop"): {
               "value": 1.0 / med, "unit": "CPIs/s", "cores": 1,
               "sample": f"{len(times)} CPIs of the same workload, median {med*1e3:.0f} ms/CPI; {what}"}}
    return res


# ----------------------------------------------------------------------------- host-buffer boundary
def e2e_host(cfg, local, np, blah2_amd):
    """One CPI handed over in HOST buffers (the PCIe-inclusive figures, never `value`): ms per CPI through
    blah2hip_amb_process_c32 (pageable complex64 in, map out), _c64 (complex128 narrowed into pinned staging), and
    through the C++ drop-in classes -- the whole sequence of blah2.cpp:264-287 (Spectrum, WienerHopf, Ambiguity,
    set_metrics, CFAR on IqData FIFOs) at BASELINE configs[1], timed by blah2_amd/host/test/test_ambiguity --sequence."""
    import re
    dmin, dmax, fmin, fmax, fs, n = cfg
    amb = blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, device=local)
    rng = np.random.default_rng(1)
    x64 = np.round(300 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
    y64 = np.round(0.8 * x64 + 30 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
    x32, y32 = x64.astype(np.complex64), y64.astype(np.complex64)
    out = {}
    for key, (xa, ya) in (("c32_ms", (x32, y32)), ("c64_ms", (x64, y64))):
        for _ in range(3):
            amb.process(xa, ya)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            amb.process(xa, ya)
        out[key] = (time.perf_counter() - t0) / reps * 1e3
    amb.close()
    exe = os.path.join(ROOT, "blah2_amd", "host", "test", "test_ambiguity")
    out["classes_ms"] = None
    if os.path.exists(exe):
        try:
            p = subprocess.run([exe, "--sequence"], capture_output=True, text=True, timeout=120,
                               env=dict(os.environ, BLAH2HIP_DEVICE=str(local)))
            m = re.search(r"CFAR\): ([0-9.]+) ms/CPI\s+\[spectrum ([0-9.]+), filter ([0-9.]+), ambiguity ([0-9.]+), set_metrics ([0-9.]+), cfar ([0-9.]+)\]", p.stdout)
            if m and p.returncode == 0:
                v = [float(g) for g in m.groups()]
                out["classes_ms"] = v[0]
                out["classes_stages_ms"] = dict(zip(("spectrum", "filter", "ambiguity", "set_metrics", "cfar"), v[1:]))
            m = re.search(r"^cut .*: ([0-9.]+) ms/CPI", p.stdout, re.M)
            if m and p.returncode == 0:
                out["classes_cut_ms"] = float(m.group(1))
        except Exception as e:  # the figure is informative; a missing binary must not void the bench line
            out["classes_error"] = str(e)
    out["note"] = ("host-buffer boundary, one CPI per call; classes_ms = SpectrumAnalyser + WienerHopf (410 taps) + Ambiguity + "
                   "set_metrics + CfarDetector1D through the C++ drop-in classes (blah2.cpp:264-287), 2 x 2 M complex<double> in IqData; "
                   "classes_cut_ms = the 2 x 2 M push_back calls of blah2.cpp:254-258 before it, inside which the samples are narrowed "
                   "and uploaded in 256 k stretches")
    return out


# ----------------------------------------------------------------------------- parity gate
def parity_check(np, O, G, cfg, fmt, chain, cfar, n_doppler, x_h, y_h, got_map, got_met, det_params=None, got_hits=None,
                 got_ok=None):
    """One CPI of the timed batch against the fp64 oracle, by the gates of oracle/gates.py (= SURVEY.md 8d, the ones the
    tests apply).  x_h / y_h: the exact values the device read.  For the full chain also: the filter's ``ok`` flag and the
    detector's hit list ``got_hits`` = (delay[], doppler[]) of this CPI, by the margin rule sized from the measured map error."""
    dmin, dmax, fmin, fmax, fs, n = cfg
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True, n_doppler_bins=n_doppler)
    res = {}
    got = got_map.astype(np.complex128)
    if chain == "full":
        ok, yf, w, r, b = O.wiener_hopf(x_h, y_h, dmin, dmax, return_filter=True)
        ref = O.ambiguity_process(d, x_h, yf)
        noise, mx = O.map_metrics(ref)
        level = float(np.max(np.abs(b))) * (d.n_corr * d.n_doppler_bins / n)  # uncancelled direct-path level
        res["chain_err_over_direct_path"] = float(np.abs(got - ref).max() / level)
        res["filter_ok"] = bool(ok) == bool(got_ok) if got_ok is not None else None
        res["pass"] = bool(ok and res["chain_err_over_direct_path"] <= GATE_CHAIN_DIRECT and res["filter_ok"] is not False)
        nm = G.notch_mask(ref.shape, d.doppler, d.delay, dmin, dmax)
        cell = G.map_cell_gate(got, ref, noise, notch=nm)                          # 1e-4 on every cell above the map's own mean level
        dbg = G.db_map_gate(got, got_met[0], ref, noise, notch=nm)
        res["cell_rel_above_mean"], res["peak_rel"] = cell["cell_rel_above_mean"], cell["peak_rel"]
        res["cell_rel_above_mean_outside_notch"] = cell["cell_rel_above_mean_outside_notch"]
        res["db_max"], res["db_max_all_cells"] = dbg["db_max_shown"], dbg["db_max_all"]
        res["notch_db_max"], res["notch_abs_err_over_mean_level"] = dbg.get("notch_db_max"), dbg.get("notch_abs_err_over_mean_level")
        res["pass"] = bool(res["pass"] and cell["ok"] and dbg["ok"])
        if got_hits is not None:
            if cfar == "2d":
                dl, dp, _, margin = O.cfar2d(ref, d.delay, d.doppler, noise, *det_params, return_margin=True)
            else:
                dl, dp, _ = O.cfar1d_fast(ref, d.delay, d.doppler, noise, *det_params)
                margin = G.cfar1d_margins(ref, det_params[0], det_params[1], det_params[2])
            dg = G.detection_gate(zip(dl, dp), zip(*got_hits), margin, d.doppler, d.delay[0], G.margin_eps(cell))
            res["detections_ok"] = dg["ok"]
            res["detections"] = {k: dg[k] for k in ("n_ref", "n_got", "n_differ", "margin_tol", "worst_margin_off_one")}
            res["pass"] = bool(res["pass"] and dg["ok"])
    else:
        ref = O.ambiguity_process(d, x_h, y_h)
        noise, mx = O.map_metrics(ref)
        cell = G.map_cell_gate(got, ref, noise, peak_tol=GATE_PEAK_REL)
        dbg = G.db_map_gate(got, got_met[0], ref, noise)
        res["peak_rel"], res["cell_rel_above_mean"] = cell["peak_rel"], cell["cell_rel_above_mean"]
        res["db_max"] = dbg["db_max_shown"]                    # the gated figure: cells within DB_FLOOR of the mean level
        res["db_max_all_cells"] = dbg["db_max_all"]            # reported: includes the deep nulls
        res["cells_over_gate_below_floor"] = dbg["cells_over_all"] - dbg["cells_over_shown"]
        res["abs_err_below_floor_over_floor_level"] = dbg["abs_err_over_floor_level"]
        res["pass"] = bool(cell["ok"] and dbg["ok"])
    res["metrics_db"] = float(max(abs(got_met[0] - noise), abs(got_met[1] - mx)))
    res["pass"] = bool(res["pass"] and res["metrics_db"] <= GATE_METRICS_DB)
    return res


# ----------------------------------------------------------------------------- one measurement
def measure(a, env):
    """One bench line's worth of measurement for the configuration in ``a``: data, handles, the timed region (barrier +
    synchronize on both sides), a second region with every kernel bracketed by HIP events, the parity gate.  Returns
    (result dict on rank 0 else None, parity dict or None).  Frees what it allocated, also when it raises: the default run
    calls it once for the headline and once per secondary BASELINE configuration (``config_legs``)."""
    handles = []
    try:
        return _measure(a, env, handles)
    finally:
        for h_ in handles:
            try:
                h_.close()
            except Exception as e:  # a close that fails must not hide what the measurement raised
                print(f"bench.py: closing an engine handle failed: {e}", file=sys.stderr)


def _measure(a, env, handles):
    torch, np, blah2_amd = env.torch, env.np, env.b2
    rank, world, local, dev, dist = env.rank, env.world, env.local, env.dev, env.dist
    cfg, cfg_desc = CONFIGS[a.config]
    dmin, dmax, fmin, fmax, fs, n = cfg
    B = a.batch if a.batch > 0 else ({"cfg3": 256 if a.chain == "full" else 32, "cfg5": 8, "small": 1024}.get(a.config, 256))
    NS = max(1, a.streams)  # independent batches in flight, each on its own stream with its own handles and result buffers
    ambs = []
    for _ in range(NS):
        ambs.append(blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, device=local, max_batch=B, n_doppler_bins=a.n_doppler))
        handles.append(ambs[-1])
    for h_ in ambs:
        h_.set_doppler_kernel(a.doppler_kernel)
        if a.hot_columns != "auto":
            h_.set_hot_columns(a.hot_columns)
        if a.fft_len:
            h_.set_fft_len(a.fft_len)
        if a.range_grid:
            h_.set_range_grid(a.range_grid)
        if a.range_kernel != "auto":
            h_.set_range_kernel({"wave": blah2_amd._lib.RANGE_WAVE, "wave1k": blah2_amd._lib.RANGE_WAVE1K, "ps": blah2_amd._lib.RANGE_PS, "e8": blah2_amd._lib.RANGE_E8,
                                 "e16": blah2_amd._lib.RANGE_E16}[a.range_kernel])
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
    fused_fir = False
    CAP = 65536
    if a.chain == "full":
        if a.fmt not in ("c32", "i16"):
            raise SystemExit("--chain full needs --fmt c32 or i16")
        # one filter handle and one set of intermediate / result buffers per stream: with --streams 2 the (latency-bound, few
        # workgroups) Toeplitz solve of one batch runs beside the transforms of the other
        whs = []
        for _ in range(NS):
            whs.append(blah2_amd.WienerHopf(dmin, dmax, n, device=local, max_batch=B))  # config.yml uses the same lag window
            handles.append(whs[-1])
        wh = whs[0]
        yfilts = [torch.empty((B, n), dtype=torch.complex64, device=dev) for _ in range(NS)]
        okflags = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NS)]
        hitss = [torch.zeros((B, CAP, 2), dtype=torch.float64, device=dev) for _ in range(NS)]  # 16-byte records
        hitcnts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NS)]
        yfilt, okflag, hits, hitcnt = yfilts[0], okflags[0], hitss[0], hitcnts[0]
        det_params = (1e-5, 2, 6, 1, 3, 5, 15.0) if a.cfar == "2d" else (1e-5, 2, 6, 5, 15.0)  # config.yml:36-40 (+ Doppler guard 1, train 3)
        det = blah2_amd.CfarDetector2D(*det_params) if a.cfar == "2d" else blah2_amd.CfarDetector1D(*det_params)
        # the filter's FIR inside the range kernel (range_fir_kernel) where one 4096-point transform covers the geometry
        fmt_id = blah2_amd.FMT_I16 if a.fmt == "i16" else blah2_amd.FMT_C32
        why_not = next((w for w in (ambs[q].fir_fusable(whs[q], fmt_id) for q in range(NS)) if w), None)
        if a.fir == "fused" and why_not:
            raise SystemExit(f"--fir fused: {why_not}")
        fused_fir = a.fir != "two-stage" and why_not is None
        if fused_fir:
            for q in range(NS):
                ambs[q].set_fir(whs[q])
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
            q = i % NS
            w_, a_, s_, yf_, ok_, o_, m_, h_, c_ = whs[q], ambs[q], sts[q], yfilts[q], okflags[q], outs[q], mets[q], hitss[q], hitcnts[q]
            if fused_fir:  # the filter's taps only; the range kernel filters the unfiltered channels on the fly
                if a.fmt == "i16":
                    w_.estimate_dev_fmt(blah2_amd.FMT_I16, iqs[r].data_ptr(), None, B, n, ok_.data_ptr(), s_)
                    a_.process_dev(blah2_amd.FMT_I16, iqs[r].data_ptr(), None, B, n, o_.data_ptr(), m_.data_ptr(), s_)
                else:
                    w_.estimate_dev_fmt(blah2_amd.FMT_C32, xs[r].data_ptr(), ys[r].data_ptr(), B, n, ok_.data_ptr(), s_)
                    a_.process_dev(blah2_amd.FMT_C32, xs[r].data_ptr(), ys[r].data_ptr(), B, n, o_.data_ptr(), m_.data_ptr(), s_)
            elif a.fmt == "i16":  # the replay format: the filter and the range kernel read the .rspduo words
                w_.process_dev_fmt(blah2_amd.FMT_I16, iqs[r].data_ptr(), None, B, n, yf_.data_ptr(), n, ok_.data_ptr(), s_)
                a_.process_dev(blah2_amd.FMT_I16X_C32Y, iqs[r].data_ptr(), yf_.data_ptr(), B, n, o_.data_ptr(), m_.data_ptr(), s_)
            else:
                w_.process_dev(xs[r].data_ptr(), ys[r].data_ptr(), B, n, yf_.data_ptr(), ok_.data_ptr(), s_)
                a_.process_dev(blah2_amd.FMT_C32, xs[r].data_ptr(), yf_.data_ptr(), B, n, o_.data_ptr(), m_.data_ptr(), s_)
            det.process_dev(a_, B, h_.data_ptr(), CAP, c_.data_ptr(), o_.data_ptr(), m_.data_ptr(), s_)
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

    # clock ramp (untimed, reported as config.prewarm_s): same steps, same buffers
    t_pre = time.perf_counter()
    n_pre = 0
    while a.prewarm_s > 0 and time.perf_counter() - t_pre < a.prewarm_s:
        for i in range(4):
            step(n_pre + i)
        n_pre += 4
        torch.cuda.synchronize()
    if a.target_s > 0:  # steps for a timed region of about target_s (every rank computes the same number: max over ranks)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        per = (time.perf_counter() - tp) / 3
        if dist is not None:
            tt = torch.tensor([per], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            per = float(tt.item())
        a.steps = int(min(5000, max(8, round(a.target_s / max(per, 1e-7)))))
    for i in range(a.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    sync()
    elapsed = time.perf_counter() - t0
    ranks_seen = 1
    if dist is not None:
        t = torch.tensor([elapsed, 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t[0:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:2], op=dist.ReduceOp.SUM)
        elapsed, ranks_seen = float(t[0].item()), int(round(float(t[1].item())))
        assert ranks_seen == world

    # the same region again, long enough (>= --long-s) that clock ramps and box-to-box jitter average out: `headline_long`
    last = a.warmup + a.steps - 1
    long_res = None
    if a.long_s > 0:
        n_long = int(min(20000, max(a.steps, math.ceil(a.long_s / (elapsed / a.steps)))))
        sync()
        t0 = time.perf_counter()
        for i in range(n_long):
            step(a.warmup + a.steps + i)
        sync()
        el_long = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el_long], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_long = float(t.item())
        last = a.warmup + a.steps + n_long - 1
        long_res = {"steps": n_long, "seconds": el_long, "ms_per_step": el_long / n_long * 1e3,
                    "value": world * B * n_long / el_long, "unit": "CPIs/s", "us_per_cpi": el_long / (B * n_long) * 1e6,
                    "chain_frac": (2 * n * s_in + cells * 8) * B * n_long / el_long / 1e9 / HBM_PEAK_GBS,
                    "note": "the timed region of `value` repeated for >= --long-s seconds (same steps, same buffers, barrier + "
                            "synchronize on both sides, max over ranks); `value` itself stays the driver's --steps region"}

    # the last timed step's outputs, kept for the parity gate
    q_last, r_last = (last % NS, last % ring)
    keep_map = outs[q_last][[0, B - 1]].cpu().numpy()
    keep_met = mets[q_last][[0, B - 1]].cpu().numpy()
    keep_ok = keep_hits = None
    if wh is not None:  # what the filter and the detector of the timed chain left for the same two CPIs
        keep_ok = okflags[q_last][[0, B - 1]].cpu().numpy()
        keep_cnt = hitcnts[q_last][[0, B - 1]].cpu().numpy()
        keep_hits = []
        for c_, k_ in zip((0, B - 1), keep_cnt):
            if int(k_) > CAP:
                raise RuntimeError(f"bench.py: {int(k_)} detections in CPI {c_}, capacity {CAP}")
            rec = hitss[q_last][c_, :max(int(k_), 1)].cpu().numpy().view(blah2_amd.HIT_DTYPE).reshape(-1)
            dt_ = blah2_amd.hits_to_detection(ambs[q_last], rec, int(k_), CAP)
            keep_hits.append((dt_.get_delay(), dt_.get_doppler()))

    # second, identical region with every kernel bracketed by HIP events on the
    # launch stream: per-kernel durations for the roofline line
    # (the device idled while the maps above were copied out: ramp the clock up again first, see --prewarm-s)
    t_pre = time.perf_counter()
    while a.prewarm_s > 0 and time.perf_counter() - t_pre < 0.5 * a.prewarm_s:
        for i in range(4):
            step(a.warmup + i)
        torch.cuda.synchronize()
    for h_ in ambs:
        h_.set_timing(True)
    for w_ in (whs if wh is not None else []):
        w_.set_timing(True)
    for i in range(a.steps):
        step(a.warmup + i)
        if NS > 1:  # per-kernel durations are one kernel's own: with batches in flight on several streams the events of one
            torch.cuda.synchronize()  # would bracket the other's kernels too; this pass runs the batches one after the other
    torch.cuda.synchronize()
    kt = {}
    for h_ in ambs + (whs if wh is not None else []):
        for k_, (ms_, n_) in h_.get_timing().items():
            kt[k_] = (kt.get(k_, (0.0, 0))[0] + ms_, kt.get(k_, (0.0, 0))[1] + n_)
        h_.set_timing(False)
    range_ms, range_n = kt["range"]
    avg_range_s = (range_ms / max(range_n, 1)) * 1e-3
    # ALGORITHMIC bytes per launch: what a kernel must move if every
    # input is read once and every output written once
    algo = algorithmic_bytes(n, s_in, cells, B, a.cfar)
    algo_bytes = algo["range"]
    achieved = algo_bytes / avg_range_s / 1e9 if avg_range_s > 0 else 0.0
    steps_timed = max(range_n, 1)
    chain_s = sum(v[0] for v in kt.values()) * 1e-3 / steps_timed
    kernels = []
    for k_, (ms_, n_) in kt.items():
        if not n_:
            continue
        us = ms_ / n_ * 1e3
        e = {"kernel": k_, "us_per_launch": us, "us_per_cpi": us / B}
        if k_ in algo:
            gbs = algo[k_] / (us * 1e-6) / 1e9
            e.update(algorithmic_bytes=algo[k_], achieved_gbs=gbs, frac_hbm=gbs / HBM_PEAK_GBS)
        else:
            e["note"] = "latency-bound: KBs of traffic (reductions / Toeplitz solve)"
        kernels.append(e)
    # the range kernel's arithmetic: (2 nSeg + 1) F-point transforms + nSeg spectrum products per pulse
    F_, nSeg_ = amb.dims.fft_len, amb.dims.n_seg
    range_flops = nD * B * ((2 * nSeg_ + 1) * 5 * F_ * math.log2(F_) + nSeg_ * 8 * F_)
    range_tflops = range_flops / avg_range_s / 1e12 if avg_range_s > 0 else 0.0

    # HBM bytes per launch of the range kernel from the committed rocprofv3 PMC
    # passes of this same command (tools/summarize_prof.py -> profiles/*_traffic.json)
    traffic, traffic_src = None, None
    import glob
    ran_range = {1: "range_kernel", 2: "range8_kernel", 3: "rangew_kernel", 5: "rangew1k_kernel", 6: "rangeps_kernel", 7: "range_fir_kernel"}.get(
        amb.info(blah2_amd._lib.INFO_LAST_RANGE_KERNEL), "range_kernel")
    prof_names = {"range": (ran_range,), "doppler": ("doppler_",),  # the range kernel this run launched, no other
                  "metrics": ("metrics_kernel",), "cfar": ("cfar2d_stream_kernel", "cfar2d_tile_kernel", "cfar2d_kernel", "cfar1d_kernel"),
                  "sat_rows": ("sat_rows_kernel",), "sat_cols": ("sat_cols_kernel",), "rotate": ("rotate_kernel",),
                  "clutter_corr": ("clutter_corr_half_kernel", "clutter_corr_kernel"), "clutter_fir": ("clutter_fir_kernel",),
                  "clutter_solve": ("clutter_solve_la_kernel", "clutter_solve_kernel"), "clutter_reduce": ("clutter_reduce_kernel",)}
    for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            tj = json.load(open(pth))
            bc = tj.get("bench_config", {})
            if (bc.get("config"), bc.get("batch"), bc.get("fmt"), bc.get("chain", "amb")) != (a.config, B, a.fmt, a.chain):
                continue
            traffic_src = os.path.relpath(pth, ROOT)
            # HBM bytes per launch of every kernel of this command that the PMC passes saw (summed over the instantiations
            # one bench key covers, e.g. a command that ran two Doppler kernels)
            for e in kernels:
                hit = [v_ for k_, v_ in tj["kernels"].items() if any(k_.startswith(pre) for pre in prof_names.get(e["kernel"], ()))]
                if hit:
                    e["traffic"] = sum(h_["hbm_bytes"] for h_ in hit)
                    if "algorithmic_bytes" in e:
                        e["traffic_over_algorithmic"] = e["traffic"] / e["algorithmic_bytes"]
            traffic = next((e.get("traffic") for e in kernels if e["kernel"] == "range"), None)
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
    # ... and what the memory system delivers to a kernel that only READS (16-byte loads over the same 1 GiB): the ceiling
    # for the range kernel, whose traffic is 99 % loads (blah2hip_stream_read_dev, csrc/capi.hip)
    L_ = blah2_amd._lib.load()
    for _ in range(3):
        blah2_amd._lib.check(L_.blah2hip_stream_read_dev(src_.data_ptr(), src_.numel() * 4, None, st))
    e0.record()
    for _ in range(10):
        blah2_amd._lib.check(L_.blah2hip_stream_read_dev(src_.data_ptr(), src_.numel() * 4, None, st))
    e1.record()
    torch.cuda.synchronize()
    read_gbs = 10 * src_.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src_, dst_
    # reference-formulation flops per CPI (5 n log2 n per FFT, 3 nfft-point FFTs
    # per pulse + one nD-point FFT per delay column) -- what the CPU path would execute
    nfft_ref = amb.get_nfft()
    ref_flops = 3 * nD * 5 * nfft_ref * math.log2(nfft_ref) + nC * 5 * nD * math.log2(nD)

    # parity gate (rank 0): first and last CPI of the last timed batch against the fp64 oracle
    parity = None
    if rank == 0 and not a.no_parity:
        from oracle import blah2_oracle as O  # checker only; nothing above this line touched it
        from oracle import gates as G         # the gates the tests apply, stated once
        checks = []
        for slot, c in enumerate((0, B - 1) if B > 1 else (0,)):  # the last CPI's tiles are later iterations of the persistent kernels
            if a.parity_cpis == "last" and c != B - 1:
                continue
            if a.fmt == "c32":
                x_h = xs[r_last][c].cpu().numpy().astype(np.complex128)
                y_h = ys[r_last][c].cpu().numpy().astype(np.complex128)
            elif a.fmt == "f16":
                xv, yv = xs[r_last][c].cpu().numpy().astype(np.float64), ys[r_last][c].cpu().numpy().astype(np.float64)
                x_h, y_h = xv[:, 0] + 1j * xv[:, 1], yv[:, 0] + 1j * yv[:, 1]
            else:
                v = iqs[r_last][c].cpu().numpy().astype(np.float64)
                x_h, y_h = v[:, 0] + 1j * v[:, 1], v[:, 2] + 1j * v[:, 3]
            sl = slot if B > 1 else 0
            checks.append(dict(cpi=c, **parity_check(np, O, G, cfg, a.fmt, a.chain, a.cfar, a.n_doppler, x_h, y_h,
                                                     keep_map[sl], keep_met[sl],
                                                     det_params if wh is not None else None,
                                                     keep_hits[sl] if keep_hits is not None else None,
                                                     int(keep_ok[sl]) if keep_ok is not None else None)))
        parity = {"pass": all(c_["pass"] for c_ in checks), "cpis": checks,
                  "oracle": "oracle/blah2_oracle.py (fp64 NumPy restatement of Ambiguity.cpp:92-172, Map.cpp:187-206"
                            + (", WienerHopf.cpp:58-163, CfarDetector1D.cpp:23-100)" if a.chain == "full" else ")")
                            + "; gates: oracle/gates.py",
                  "gates": {"peak_rel": GATE_PEAK_REL if a.chain != "full" else G.CELL_TOL,
                            ("cell_rel_above_mean" if a.chain != "full" else "cell_rel_above_mean_outside_notch"): G.CELL_TOL,
                            "db_max": GATE_DB_MAP, "db_floor_below_mean_level": DB_FLOOR,
                            "metrics_db": GATE_METRICS_DB, "chain_err_over_direct_path": GATE_CHAIN_DIRECT,
                            "detections": f"identical up to cells whose threshold margin is within {G.MARGIN_K:g} x the measured map error of 1",
                            "notch_abs_err_over_mean_level": G.NOTCH_ABS}}
        for key in ("peak_rel", "cell_rel_above_mean", "cell_rel_above_mean_outside_notch", "db_max", "metrics_db",
                    "chain_err_over_direct_path", "notch_db_max", "notch_abs_err_over_mean_level"):
            vals = [c_[key] for c_ in checks if c_.get(key) is not None]
            if vals:
                parity[key] = max(vals)
        if a.chain == "full":
            parity["detections_ok"] = all(c_.get("detections_ok", False) for c_ in checks)
            parity["filter_ok"] = all(c_.get("filter_ok") is True for c_ in checks)
            parity["detections"] = [c_.get("detections") for c_ in checks]

    res = None
    if rank == 0:
        total_cpis = world * B * a.steps
        res = {
            "metric": "CPIs/s", "value": total_cpis / elapsed, "unit": "CPIs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_desc} -> {nD} Doppler x {nC} delay bins, synthetic {a.fmt} IQ resident in HBM"
                       + ("" if a.chain == "amb" else f" + clutter filter ({dmax - dmin} taps) + {a.cfar} CA-CFAR"),
                       "chain": a.chain,
                       "batch_cpis_per_step": B, "fmt": a.fmt, "n_samples": n, "fs": fs,
                       "n_doppler_bins": nD, "n_delay_bins": nC, "n_corr": amb.get_n_corr(),
                       "fft_len": amb.dims.fft_len, "n_seg": amb.dims.n_seg, "seg_len": amb.dims.seg_len,
                       "doppler_kernel": amb.last_doppler_kernel(),
                       "prewarm_s": a.prewarm_s, "prewarm_steps": n_pre,
                       "range_kernel": {1: "e16", 2: "e8", 3: "wave", 5: "wave1k", 6: "ps", 7: "fir+e16"}.get(amb.info(blah2_amd._lib.INFO_LAST_RANGE_KERNEL)),
                       "fir": ("fused into the range kernel (range_fir_kernel)" if fused_fir else "two-stage (clutter_fir_kernel)") if wh is not None else None,
                       "ring_batches": ring, "streams_per_gpu": NS, "sharding": f"{world} independent CPI streams, one per GPU",
                       "ranks_seen_by_rccl": ranks_seen if dist is not None else None},
            "cells_per_s": total_cpis * cells / elapsed,
            "us_per_cpi": elapsed / (B * a.steps) * 1e6,
            "per_gpu_cpis_per_s": total_cpis / elapsed / world,
            "parity": parity,
            "headline_long": long_res,
            "roofline": {"bound": "hbm", "kernel": {1: "range_kernel", 2: "range8_kernel", 3: "rangew_kernel", 5: "rangew1k_kernel", 6: "rangeps_kernel", 7: "range_fir_kernel"}.get(
                             amb.info(blah2_amd._lib.INFO_LAST_RANGE_KERNEL), "range_kernel"), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "traffic_measured_on": (None if traffic is None else
                                                 f"the builder's box: rocprofv3 PMC passes of this same command, {traffic_src} "
                                                 "(not a measurement of this run)"),
                         "copy_ceiling": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs,
                         "read_ceiling": read_gbs, "frac_of_read_ceiling": achieved / read_gbs,
                         "ceilings_note": "measured in this process: copy = torch device-to-device copy of 1 GiB (read + written "
                                          "bytes per second); read = a kernel of 16-byte loads over the same 1 GiB",
                         "valu": {"achieved_tflops": range_tflops, "peak_tflops": VALU_PEAK_TFLOPS,
                                  "frac": range_tflops / VALU_PEAK_TFLOPS,
                                  "flops_counted": "(2 nSeg + 1) transforms x 5 F log2 F + nSeg x 8 F per pulse"},
                         "chain_achieved": (2 * n * s_in + cells * 8) * total_cpis / world / elapsed / 1e9,
                         # B_amb over the time of the whole kernel chain (`frac` above is the dominant kernel's)
                         "chain_frac": (2 * n * s_in + cells * 8) * total_cpis / world / elapsed / 1e9 / HBM_PEAK_GBS,
                         "ref_equivalent_tflops": ref_flops * total_cpis / world / elapsed / 1e12,
                         "avg_launch_us": avg_range_s * 1e6, "launches_timed": range_n,
                         "kernels": kernels,
                         "kernel_us_per_step": {k: v[0] / max(v[1], 1) * 1e3 for k, v in kt.items() if v[1]},
                         "chain_us_per_step": chain_s * 1e6},
        }
    return res, parity


# ----------------------------------------------------------------------------- the other BASELINE configurations
LEGS = [
    # (key, argv beyond the headline's, what BASELINE.json calls it)
    ("configs[2]", ["--config", "cfg3", "--chain", "full", "--cfar", "2d", "--batch", "32", "--streams", "2"],
     "1xMI355X, 10 MS/s, 1 s CPI, 1024 Doppler x 2048 range, + clutter filter + 2D CA-CFAR"),
    ("configs[2] ambiguity only", ["--config", "cfg3", "--batch", "32"],
     "the same geometry, range + Doppler + metrics only (what B_amb prices)"),
    ("configs[4]", ["--config", "cfg5", "--fmt", "f16", "--batch", "8"],
     "1xMI355X, 20 MS/s, 2 s CPI, 2048 Doppler bins, fp16 IQ storage with fp32 accumulate"),
    ("configs[1], one CPI per launch", ["--config", "cfg2", "--batch", "1"],
     "real-time single stream (blah2.cpp:263-289): a lone CPI per launch, range + Doppler + metrics"),
    ("configs[1], one CPI per launch, full chain", ["--config", "cfg2", "--batch", "1", "--chain", "full", "--cfar", "1d"],
     "the same with the clutter filter (410 taps) in front and the 1-D detector behind"),
]


def _could_not_run(e, env):
    """The errors that mean a leg COULD NOT RUN on this box (memory on a shared device, /dev/shm too small) -- anything
    else (a HIP launch failure, an illegal address, a Python error, a violated gate) is a regression and propagates."""
    if isinstance(e, env.torch.cuda.OutOfMemoryError):
        return True
    if isinstance(e, env.b2.Blah2HipError) and e.code == env.b2._lib.ERR_HIP and "out of memory" in str(e).lower():
        return True
    return isinstance(e, LegSkipped)


class LegSkipped(RuntimeError):
    """A leg that has nothing to measure on this box (says why)."""


def config_legs(a, env):
    """Short measurements of every other single-GPU configuration BASELINE.json names, in this process after the
    headline: the same measure() (same timed region, same HIP-event pass, same oracle gates on the last CPI of the last
    timed batch), about 0.3 s of timed work each.  A leg that cannot run for lack of memory is recorded as such; any other
    error -- and a violated gate -- ends the run with a non-zero status (after the line has been printed, see main)."""
    out = []
    for key, extra, what in LEGS:
        t0 = time.perf_counter()
        try:
            env.torch.cuda.empty_cache()
            la = _leg_args(a, extra)
            r, par = measure(la, env)
            rl = r["roofline"]
            ks = [{"kernel": k["kernel"], "us_per_cpi": k["us_per_cpi"], "frac_hbm": k.get("frac_hbm")} for k in rl["kernels"]]
            dom = max(ks, key=lambda k: k["us_per_cpi"]) if ks else None
            out.append({"baseline_config": key, "baseline_wording": what, "workload": r["config"]["workload"],
                        "chain": la.chain, "fmt": la.fmt, "batch_cpis_per_step": r["config"]["batch_cpis_per_step"],
                        "streams_per_gpu": r["config"]["streams_per_gpu"], "steps": r["steps"],
                        "cpis_per_s": r["value"], "us_per_cpi": r["us_per_cpi"], "cells_per_s": r["cells_per_s"],
                        "n_doppler_bins": r["config"]["n_doppler_bins"], "n_delay_bins": r["config"]["n_delay_bins"],
                        "range_kernel": r["config"]["range_kernel"], "doppler_kernel": r["config"]["doppler_kernel"],
                        "fir": r["config"].get("fir"),
                        "chain_frac": rl["chain_frac"],  # B_amb over the time of the whole chain, against 8 TB/s
                        "dominant_kernel": dom, "kernels": ks,
                        "parity": None if par is None else {k: par[k] for k in (
                            "pass", "peak_rel", "cell_rel_above_mean", "cell_rel_above_mean_outside_notch", "db_max", "metrics_db", "chain_err_over_direct_path",
                            "detections_ok", "filter_ok", "detections", "notch_db_max", "notch_abs_err_over_mean_level") if k in par},
                        "leg_wall_s": time.perf_counter() - t0})
        except Exception as e:
            if not _could_not_run(e, env):
                env.failed_legs.append((key, e))  # main() raises it after the line is out
                out.append({"baseline_config": key, "failed": f"{type(e).__name__}: {e}"[:600], "leg_wall_s": time.perf_counter() - t0})
                if isinstance(e, env.b2.Blah2HipError) or "HIP" in str(e):
                    break  # a sticky device error would fail every later leg the same way
            else:
                out.append({"baseline_config": key, "error": f"{type(e).__name__}: {e}"[:600], "leg_wall_s": time.perf_counter() - t0})
        env.torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------- replay legs (host buffers: never `value`)
REPLAY_LEGS = [
    # (key, config, CPIs in the /dev/shm capture, batch, clutter filter, what BASELINE.json calls it)
    ("configs[3] at N = 1", "cfg3", 96, 16, False,
     "8xMI355X CPI-sharded replay over xGMI, 10 MS/s, 1 s CPI, 1024 Doppler bins -- its one-rank point: "
     "int16 capture -> pinned ring -> PCIe -> range + Doppler + metrics + 1-D CFAR (RspDuo.cpp:150-179, blah2.cpp:254-258)"),
    ("configs[1] geometry, replay", "cfg2", 512, 16, False, "the same path at 2 MS/s, 1 s CPI, 513 x 411"),
]


def write_capture(torch, path, n, n_cpis, fs, dev, distinct=2):
    """A seeded synthetic .rspduo capture (int16 I1 Q1 I2 Q2, RspDuo.cpp:512-526): `distinct` CPIs synthesised on the
    device, repeated to n_cpis (the replay's cost does not depend on the values)."""
    x, y = synth_batch(torch, distinct, n, 4242, fs, dev)
    iq = torch.stack([x.real, x.imag, y.real, y.imag], dim=-1).to(torch.int16).contiguous().cpu().numpy()
    blobs = [iq[k].tobytes() for k in range(distinct)]
    del x, y, iq
    with open(path, "wb") as f:
        for k in range(n_cpis):
            f.write(blobs[k % distinct])


def pinned_h2d_rate(torch, dev, nbytes=256 << 20):
    """The pinned host-to-device rate of this box (bytes/s), the bound of a replay: the best of three timed stretches of
    eight 256 MB copies after an untimed one (the first copies out of a fresh pinned buffer run at half the rate)."""
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    h.fill_(1)
    d_ = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    best = 0.0
    for rep in range(4):
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(8):
            d_.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        if rep:
            best = max(best, 8 * nbytes / (time.perf_counter() - tc))
    del h, d_
    return best


def replay_legs(a, env, min_seconds=2.0):
    """BASELINE configs[3] ("CPI-sharded replay") at N = 1, measured in this process: a capture in /dev/shm read
    cyclically for >= 2 s through blah2_amd.replay.GpuChain (reader threads -> pinned ring -> copy stream -> the device
    chain -> results back), one rank.  Reports CPIs/s, the fraction of the pinned host-to-device rate measured beside it,
    and the share of the wall clock the GPU spent computing.  Host buffers cross PCIe here: this is never `value`."""
    torch = env.torch
    from blah2_amd import replay as R
    out = []
    for key, config, n_file, batch, clutter, what in REPLAY_LEGS:
        t0 = time.perf_counter()
        path = f"/dev/shm/blah2_bench_{os.getpid()}_{config}.rspduo"
        chain = None
        try:
            (dmin, dmax, fmin, fmax, fs, n), desc = CONFIGS[config]
            bytes_per_cpi = n * R.BYTES_PER_SAMPLE
            st_ = os.statvfs("/dev/shm")
            if st_.f_bavail * st_.f_frsize < 1.25 * n_file * bytes_per_cpi:
                raise LegSkipped(f"/dev/shm has {st_.f_bavail * st_.f_frsize / 1e9:.1f} GB free, the capture needs {n_file * bytes_per_cpi / 1e9:.1f}")
            torch.cuda.empty_cache()
            R.pin_to_device_node(torch, env.local)  # the capture's page-cache pages on the GPU's NUMA node too
            write_capture(torch, path, n, n_file, fs, env.dev)
            t_written = time.perf_counter() - t0
            rate = pinned_h2d_rate(torch, env.dev)
            cfg = {"fs": fs, "n_samples": n,
                   "ambiguity": {"delayMin": dmin, "delayMax": dmax, "dopplerMin": fmin, "dopplerMax": fmax},
                   "clutter": {"enable": clutter, "delayMin": dmin, "delayMax": dmax},
                   "detection": {"enable": True, "pfa": 1e-5, "nGuard": 2, "nTrain": 6, "minDelay": 5, "minDoppler": 15.0}}
            chain = R.GpuChain(cfg, env.local, batch, depth=3, reader_threads=4, read_mode="memmove")
            warm = R.LoopedCapture(path, n, 1)
            R.replay(warm, chain, batch, emit=lambda r: None)  # untimed: clock ramp, first touch of the pinned ring
            warm.close()
            est = n_file * bytes_per_cpi / (0.85 * rate)
            times = max(1, math.ceil(min_seconds / est))
            chain.busy_ms, chain.batches_done = 0.0, 0
            first, cnt, last_r = [None], [0], [None]

            def emit(r):
                if first[0] is None:
                    first[0] = time.perf_counter() - tr
                cnt[0] += 1
                last_r[0] = r

            el, cpu_s, expect = 0.0, 0.0, 0
            while el < min_seconds:  # one pipelined stream of `times` passes; again if the estimate was short
                cap = R.LoopedCapture(path, n, times)
                cpu0 = time.process_time()
                tr = time.perf_counter()
                R.replay(cap, chain, batch, emit=emit)
                el += time.perf_counter() - tr
                cpu_s += time.process_time() - cpu0
                expect += n_file * times
                cap.close()
            nD, nC = chain.amb.get_n_doppler_bins(), chain.amb.get_n_delay_bins()
            if cnt[0] != expect or "noisePower" not in (last_r[0] or {}):
                raise RuntimeError(f"replay emitted {cnt[0]} of {expect} CPIs")
            out.append({"baseline_config": key, "baseline_wording": what, "workload": f"{desc} -> {nD} x {nC}, int16 .rspduo capture "
                        f"of {n_file} CPIs ({n_file * bytes_per_cpi / 1e9:.1f} GB) in /dev/shm read cyclically ({expect // n_file} passes), one rank",
                        "n_gpus": 1, "chain": ("clutter+" if clutter else "") + "ambiguity+metrics+cfar1d", "batch_cpis": batch,
                        "reader_threads": 4, "read_mode": chain.read_mode, "cpis_timed": cnt[0], "seconds": el,
                        "cpis_per_s": cnt[0] / el, "cells_per_s": cnt[0] * nD * nC / el,
                        "effective_GBps": cnt[0] * bytes_per_cpi / el / 1e9, "pinned_h2d_GBps": rate / 1e9,
                        "frac_of_pinned_link": cnt[0] * bytes_per_cpi / el / rate,
                        "gpu_busy_share": chain.busy_ms * 1e-3 / el, "gpu_us_per_cpi": chain.busy_ms * 1e3 / max(cnt[0], 1),
                        "first_result_after_s": first[0], "host_cpu_s_per_cpi": cpu_s / max(cnt[0], 1),
                        "cpus_allowed": len(os.sched_getaffinity(0)), "capture_written_s": t_written,
                        "last_cpi": {k: last_r[0][k] for k in ("noisePower", "maxPower")} | {"detections": len(last_r[0].get("delay", []))},
                        "note": "PCIe-inclusive (host buffers): bound by the host link, not by the kernels; never `value`",
                        "leg_wall_s": time.perf_counter() - t0})
        except Exception as e:
            if not _could_not_run(e, env):
                env.failed_legs.append((key, e))
                out.append({"baseline_config": key, "failed": f"{type(e).__name__}: {e}"[:600], "leg_wall_s": time.perf_counter() - t0})
            else:
                out.append({"baseline_config": key, "error": f"{type(e).__name__}: {e}"[:600], "leg_wall_s": time.perf_counter() - t0})
        finally:
            if chain is not None:
                try:
                    chain.close()
                except Exception as e:
                    print(f"bench.py: closing the replay chain failed: {e}", file=sys.stderr)
            if os.path.exists(path):
                os.remove(path)
            torch.cuda.empty_cache()
    return out


def _leg_args(a, extra):
    import copy
    la = copy.copy(a)
    la.config, la.chain, la.fmt, la.cfar, la.batch, la.streams = "cfg2", "amb", "c32", "2d", 0, 1
    la.n_doppler, la.doppler_kernel, la.range_kernel, la.fft_len, la.range_grid = 0, "auto", "auto", 0, 0
    it = iter(extra)
    for k in it:
        v = next(it)
        name = k[2:].replace("-", "_")
        setattr(la, name, int(v) if name in ("batch", "streams") else v)
    la.target_s, la.warmup, la.prewarm_s, la.parity_cpis, la.long_s = 0.3, 3, 0.3, "last", 0.0
    la.no_parity = a.no_parity
    return la


# ----------------------------------------------------------------------------- main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0,
                    help="CPIs per step (per GPU); default 256 for the 2 MS/s configs (8 GB of fp32 IQ per step: a pulse is the scheduling unit of the range kernel, and 256 x 513 pulses are 42.75 rounds of its "
                         "3072 resident waves -- measured on one box: 98.8 k / 107.7 k / 110.3 k CPIs/s at 32 / 128 / 256); "
                         "32 for cfg3 (256 with --chain full, 41 GB of IQ: the Toeplitz solve takes 1.6 ms per launch whatever the batch -- "
                         "4.69 k CPIs/s at 128, 4.92 k at 256), 8 for cfg5 -- "
                         "measured: cfg3 93.9 / 88.8 / 84.7 us/CPI at 8 / 16 / 32, cfg5 142.8 / 138.4 / 138.8 at 4 / 8 / 16")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--fmt", default="c32", choices=["c32", "i16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison after the timed region")
    ap.add_argument("--chain", default="amb", choices=["amb", "full"],
                    help="amb: range+Doppler+metrics (BASELINE headline); full: clutter filter + amb + CFAR (configs[2])")
    ap.add_argument("--cfar", default="2d", choices=["1d", "2d"])
    ap.add_argument("--n-doppler", type=int, default=0,
                    help="explicit number of Doppler bins (extension; 0 = the reference constructor's rule, which gives "
                         "513 at the headline configuration; 512 gives the literal BASELINE wording)")
    ap.add_argument("--doppler-kernel", default="auto", help="force a Doppler kernel (auto, tile8, tile8k, tile16, tile16wg, sub4, tilew, tilew2, tilew4, tilem, column, direct)")
    ap.add_argument("--prewarm-s", type=float, default=0.6,
                    help="seconds of untimed steps BEFORE the W warmup steps: the shader clock needs ~0.3 s of load to ramp up "
                         "from idle (measured: steps 5..25 of a cold run are 4-5 %% slower than steady state)")
    ap.add_argument("--range-kernel", default="auto", choices=["auto", "wave", "wave1k", "ps", "e16", "e8"],
                    help="range kernel: by transform length (F = 2048: the one-wave kernel, F = 4096: the two-wave kernel), or forced")
    ap.add_argument("--fft-len", type=int, default=0, choices=[0, 1024, 2048, 4096],
                    help="force the range transform length (0 = the planner's choice); diagnostics")
    ap.add_argument("--range-grid", type=int, default=0,
                    help="cap the range kernel's grid (workgroups; 0 = its residency): e.g. the CU count for ONE workgroup per CU; diagnostics")
    ap.add_argument("--hot-columns", default="auto", choices=["auto", "off", "always"],
                    help="BLAH2HIP_OPT_HOT_COLUMNS: the fp64 Doppler transform of the columns under the tallest peaks (an A/B switch; auto is the engine's default)")
    ap.add_argument("--fir", default="auto", choices=["auto", "fused", "two-stage"],
                    help="--chain full: the clutter filter's FIR fused into the range kernel where it is covered (auto), required (fused) or never")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent CPI streams per GPU (engine handles on their own HIP streams); successive "
                         "steps alternate between them so one batch's Doppler stage overlaps the next batch's range stage")
    ap.add_argument("--no-configs", action="store_true",
                    help="headline only: skip the short legs for the other single-GPU BASELINE configurations (`configs` in the JSON line)")
    ap.add_argument("--long-s", type=float, default=0.5,
                    help="after the K timed steps, the same region again for at least this long (`headline_long` in the line); 0 = skip")
    ap.add_argument("--no-replay", action="store_true",
                    help="skip the PCIe-inclusive replay legs (`replay` in the JSON line: BASELINE configs[3] at N = 1)")
    ap.add_argument("--target-s", type=float, default=0.0,
                    help="pick --steps so that the timed region lasts about this long (the legs of `configs` use it)")
    ap.add_argument("--parity-cpis", default="both", choices=["both", "last"],
                    help="CPIs of the last timed batch compared with the oracle: first and last, or the last only")
    a = ap.parse_args(argv)

    import torch
    action, detail = plan_launch(a.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0)
    if action == "error":
        raise SystemExit(f"bench.py: {detail} (the HIP path has no CPU fallback)")
    if action == "spawn":
        raise SystemExit(spawn_ranks(detail, argv))

    import numpy as np
    import blah2_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:  # under torchrun, also at N = 1
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == a.gpus

    import types
    env = types.SimpleNamespace(torch=torch, np=np, b2=blah2_amd, rank=rank, world=world, local=local, dev=dev, dist=dist,
                                failed_legs=[])
    res, parity = measure(a, env)
    if rank == 0:
        cfg = CONFIGS[a.config][0]
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg)
            if a.config == "cfg2":
                res["e2e_host"] = e2e_host(cfg, local, np, blah2_amd)
        if world == 1 and not a.no_configs and (a.config, a.chain, a.fmt) == ("cfg2", "amb", "c32"):
            res["configs"] = config_legs(a, env)  # every other single-GPU BASELINE configuration, same process, same box
            if not a.no_replay and not env.failed_legs:
                res["replay"] = replay_legs(a, env)  # BASELINE configs[3] at N = 1 (PCIe-inclusive; never `value`)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()  # rank 0 has been checking parity: every rank leaves the group together
        dist.destroy_process_group()
    if parity is not None and not parity["pass"]:
        raise SystemExit("bench.py: parity gate violated: " + json.dumps(parity))
    # after the line has been printed: the headline stands.  A leg whose ORACLE GATE failed, or that raised anything but
    # "could not run on this box" (out of memory on a shared device, no room in /dev/shm), sets the exit status like the
    # headline's would; a leg that could not run is reported on stderr only -- it says nothing about the headline
    legs = (res or {}).get("configs", []) + (res or {}).get("replay", [])
    for c in legs:
        if "error" in c:
            print(f"bench.py: leg {c['baseline_config']} did not run: {c['error']}", file=sys.stderr)
    bad = [c for c in legs if c.get("parity") and not c["parity"]["pass"]]
    if bad:
        raise SystemExit("bench.py: parity gate violated in a secondary configuration: " + json.dumps(bad))
    if env.failed_legs:
        key, e = env.failed_legs[0]
        print(f"bench.py: leg {key} FAILED: {type(e).__name__}: {e}", file=sys.stderr)
        raise e


if __name__ == "__main__":
    main()
