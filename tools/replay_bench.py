#!/usr/bin/env python3
"""Replay throughput on the GPU box: the pipelined .rspduo replay (blah2_amd/replay.py) against the host link's bound.

    python tools/replay_bench.py [--config cfg2] [--cpis 96] [--batch 8] [--out gpurun_out/replay.json]

Writes a synthetic capture (int16 I1 Q1 I2 Q2, seeded) to /dev/shm, measures the pinned host-to-device copy rate, then
replays the capture through the device chain (ambiguity + 1-D CFAR; with the clutter filter in front) and reports
CPIs/s beside the PCIe bound (bytes per CPI / measured pinned H2D rate) and the HBM-resident figure of bench.py.
Finally it replays the same file with TWO ranks sharing the one GPU (gloo group, `python -m blah2_amd.replay` under
torch.distributed.run) and checks that rank 0 emitted every CPI once, in file order.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from blah2_amd import replay as R


def make_capture(path, n, n_cpis, fs, distinct=12):
    """n_cpis CPIs, the first `distinct` of them generated and then repeated (the replay cost does not depend on the values)."""
    rng = np.random.default_rng(2024)
    blobs = []
    for k in range(min(distinct, n_cpis)):
        x = np.round(300 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
        t = np.arange(n) / fs
        xd = np.concatenate([np.zeros(37, dtype=complex), x[:-37]]) * np.exp(2j * np.pi * (-63.0) * t)
        y = np.round(0.8 * x + 0.05 * xd + 30 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
        blobs.append(np.stack([x.real, x.imag, y.real, y.imag], axis=-1).astype("<i2").tobytes())
    with open(path, "wb") as f:
        for k in range(n_cpis):
            f.write(blobs[k % len(blobs)])


def h2d_rate(dev, nbytes=256 << 20, reps=10):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for _ in range(2):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    return reps * nbytes / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--cpis", type=int, default=6144,
                    help="CPIs in the capture (cut down to what /dev/shm holds): at cfg 2 a pass over 6144 CPIs (98 GB) takes "
                         "about two seconds at the link's rate -- round 4 timed passes of 0.12 s")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="a timed figure is passes over the capture until this long")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "replay.json"))
    a = ap.parse_args()
    (dmin, dmax, fmin, fmax, fs, n), desc = bench.CONFIGS[a.config]
    dev = torch.device("cuda", 0)
    numa = R.pin_to_device_node(torch, 0)  # before the capture is written: its page-cache pages on the GPU's node too
    path = f"/dev/shm/blah2_replay_{a.config}.rspduo"
    st = os.statvfs("/dev/shm")
    fit = int(st.f_bavail * st.f_frsize * 0.7 / (n * R.BYTES_PER_SAMPLE)) // a.batch * a.batch
    if fit < a.cpis:
        print(f"[replay_bench] /dev/shm holds {fit} CPIs of the {a.cpis} asked for", file=sys.stderr)
        a.cpis = max(6 * a.batch, fit)
    t0 = time.perf_counter()
    make_capture(path, n, a.cpis, fs)
    t_gen = time.perf_counter() - t0
    bytes_per_cpi = n * R.BYTES_PER_SAMPLE
    rate = h2d_rate(dev)
    bound = rate / bytes_per_cpi
    res = {"config": a.config, "workload": desc, "cpis": a.cpis, "batch": a.batch, "bytes_per_cpi": bytes_per_cpi,
           "pinned_h2d_GBps": rate / 1e9, "pcie_bound_cpis_per_s": bound, "capture": path, "capture_written_s": t_gen,
           "host_cores": os.cpu_count(), "pinned_to_gpu_numa_node": numa, "cpus_allowed": len(os.sched_getaffinity(0)), "runs": []}
    base = {"fs": fs, "n_samples": n,
            "ambiguity": {"delayMin": dmin, "delayMax": dmax, "dopplerMin": fmin, "dopplerMax": fmax},
            "detection": {"enable": True, "pfa": 1e-5, "nGuard": 2, "nTrain": 6, "minDelay": 5, "minDoppler": 15.0}}
    # read modes (GpuChain): "memmove" copies out of the mapping into a pinned ring, "pread" the same through the kernel,
    # "mapped" registers the page cache's own pages with the device (no CPU copy)
    for name, clutter, depth, threads, mode, *warm in (
            ("ambiguity+cfar, 4 reader threads", False, 3, 4, "memmove"),
            ("ambiguity+cfar, 8 reader threads", False, 3, 8, "memmove"),
            ("clutter+ambiguity+cfar, 4 reader threads", True, 3, 4, "memmove"),
            ("ambiguity+cfar, pread, 4 reader threads", False, 3, 4, "pread"),
            ("ambiguity+cfar, mapped, 1 reader thread", False, 3, 1, "mapped"),
            ("ambiguity+cfar, mapped, 1 reader thread, the same mapping again (warm page tables)", False, 3, 1, "mapped", True),
            # the first configuration once more: the first run over a freshly written capture also pays its cold start
            ("ambiguity+cfar, 4 reader threads (again, after the others)", False, 3, 4, "memmove")):
        cfg = dict(base, clutter={"enable": clutter, "delayMin": dmin, "delayMax": dmax})
        chain = R.GpuChain(cfg, 0, a.batch, depth=depth, reader_threads=threads, read_mode=mode)
        cap = R.RspduoFile(path, n)
        # an untimed stretch (clock ramp, first touch of the pinned ring); for the warm-mapping run a whole pass, which is
        # what fills the mapping's page tables
        R.replay(cap, chain, a.batch, limit=None if warm else min(a.cpis, 12 * a.batch), emit=lambda r: None)
        chain.release_all()
        if not warm:
            cap.close()
        passes = []
        cpu0 = time.process_time()
        t_all = time.perf_counter()
        while not passes or time.perf_counter() - t_all < a.min_seconds:
            cnt = [0]
            first = [None]
            # a NEW mapping of the capture per pass: a replay touches every page of its file once, so the mapped path pays the
            # (minor) page faults of a fresh mapping inside the timed region, like a first pass over a capture in the page cache
            if not warm or cap._mm.size == 0:
                cap = R.RspduoFile(path, n)
            t0 = time.perf_counter()

            def emit(r):
                if first[0] is None:
                    first[0] = time.perf_counter() - t0
                cnt[0] += 1

            R.replay(cap, chain, a.batch, emit=emit)
            passes.append((time.perf_counter() - t0, first[0], cnt[0]))
            chain.release_all()
            if not warm:
                cap.close()
        cpu_s = time.process_time() - cpu0  # user + system time of all threads of this process, all passes
        mode_ran = chain.read_mode
        chain.close()
        el, n_done = sum(p_[0] for p_ in passes), sum(p_[2] for p_ in passes)  # every timed pass: >= min-seconds of replay
        first_s = passes[0][1]
        run = {"chain": name, "read_mode": mode_ran, "host_cpu_s_per_cpi": cpu_s / max(n_done, 1), "cpis_per_s": n_done / el, "frac_of_pcie_bound": n_done / el / bound, "seconds": el,
               "first_result_after_s": first_s, "depth": depth, "reader_threads": threads,
               "effective_GBps": n_done * bytes_per_cpi / el / 1e9, "cpis_timed": n_done, "passes_s": [p_[0] for p_ in passes]}
        print(json.dumps(run), flush=True)
        res["runs"].append(run)
    # two ranks on the one GPU (gloo), in-order emission through the per-round gather
    import yaml
    cfgp = "/dev/shm/blah2_replay_cfg.yml"
    yaml.safe_dump({"capture": {"fs": fs}, "process": {"data": {"cpi": n / fs}, "ambiguity": base["ambiguity"],
                                                       "clutter": {"enable": False, "delayMin": dmin, "delayMax": dmax},
                                                       "detection": base["detection"]},
                    "network": {"ip": "0.0.0.0", "ports": {"map": 3001, "detection": 3002}}}, open(cfgp, "w"))
    lim = min(a.cpis, 6 * a.batch)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", "-m", "blah2_amd.replay", path, "-c", cfgp, "--batch", str(a.batch), "--limit", str(lim)]
    t0 = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
    el = time.perf_counter() - t0
    order = [json.loads(l)["cpi"] for l in p.stdout.splitlines() if l.startswith("{")]
    res["two_ranks_one_gpu"] = {"rc": p.returncode, "emitted": len(order), "in_file_order": order == list(range(lim)),
                                "expected": lim, "seconds_incl_startup": el, "stderr_tail": p.stderr[-300:] if p.returncode else ""}
    print(json.dumps(res["two_ranks_one_gpu"]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    os.remove(path)


if __name__ == "__main__":
    main()
