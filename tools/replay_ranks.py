#!/usr/bin/env python3
"""The HOST side of an N-rank replay on one box, upload stubbed: can eight ranks share one host?

    python tools/replay_ranks.py --ranks 1,2,4,8 [--seconds 2.5] [--out gpurun_out/replay_ranks.json]

An 8-GPU replay (batches of consecutive CPIs sharded round-robin over the ranks, no data-path collective)
needs every rank to move its shard page cache -> pinned ring -> PCIe at the link's rate (57 GB/s pinned on this host, so
8 x 57 = 456 GB/s of reads AND writes through the two sockets' memory).  The one-GPU box cannot run eight uploads, but it
can run the eight READ paths: N processes, each with its own mapping of the capture, its own pinned ring of three
16-CPI slots and four reader threads pinned to "its GPU's" NUMA node (the node map of an 8-GPU box: ranks split evenly
over the nodes), walking its own shard of the batches for at least --seconds; the upload is a no-op.  Reported per N and
read mode (memmove / pread out of the page cache; `mapped` = the hipHostRegister + unregister of the windows, which is all
the zero-copy path costs the host): aggregate GB/s, GB/s per rank against the 57 a link takes, host CPU-seconds per CPI.
The cgroup's CPU quota of the box is reported beside it -- when it is below ranks x threads, IT is the bound measured.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def node_cpus():
    nodes = []
    base = "/sys/devices/system/node"
    try:
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                cpus = set()
                for part in open(f"{base}/{d}/cpulist").read().strip().split(","):
                    if part:
                        a, _, b = part.partition("-")
                        cpus.update(range(int(a), int(b or a) + 1))
                cpus &= os.sched_getaffinity(0)
                if cpus:
                    nodes.append(sorted(cpus))
    except OSError:
        pass
    return nodes or [sorted(os.sched_getaffinity(0))]


def cgroup_quota():
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            s = open(p).read().split()
            if p.endswith("cpu.max"):
                return None if s[0] == "max" else float(s[0]) / float(s[1])
            q = float(s[0])
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return None if q < 0 else q / per
        except (OSError, ValueError, IndexError):
            continue
    return None


def worker(a):
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from blah2_amd import replay as R
    rank, world = a.rank, a.world
    nodes = node_cpus()
    mine = nodes[rank * len(nodes) // world]  # ranks split evenly over the nodes, like the GPUs of an 8-GPU box
    os.sched_setaffinity(0, mine)
    n = a.n
    cap = R.RspduoFile(a.capture, n)
    B = a.batch
    shards = R.shard_batches(cap.n_cpis, B, rank, world)
    ring = [torch.empty((B, n, 4), dtype=torch.int16) for _ in range(3)]
    if torch.cuda.is_available():  # (a dry run of the tool itself works without a device, on pageable buffers)
        ring = [r.pin_memory() for r in ring]
    pool = ThreadPoolExecutor(max_workers=a.threads)
    hip = R._hip_runtime(torch) if a.mode == "mapped" else None
    if hip is not None:
        torch.zeros(1, device="cuda")  # a context for hipHostRegister
    # one untimed pass over the ring (first touch of the pinned pages, page tables of the mapping's first windows)
    for s, (k0, cnt) in zip(ring, shards):
        cap.read_into(k0, cnt, s.numpy(), pool, a.threads, how="memmove" if a.mode == "mapped" else a.mode)
    while time.time() < a.start_at:
        time.sleep(0.001)
    t0 = time.perf_counter()
    c0 = time.process_time()
    done = 0
    i = 0
    while True:
        k0, cnt = shards[i % len(shards)]
        if a.mode == "mapped" and i % len(shards) == 0:
            # a replay touches every page of its capture once: a NEW mapping per pass over the shard, so that the
            # registrations pay the page-table fill of a first touch (a mapping that stays registers several times faster)
            cap.close()
            cap = R.RspduoFile(a.capture, n)
        if a.mode == "mapped":
            addr, nbytes = cap.window(k0, cnt)
            head, pieces, tail = R.page_split(addr, nbytes, a.threads)

            def reg(pc):
                rc = hip.hipHostRegister(addr + pc[0], pc[1], 0)
                if rc == 0:
                    hip.hipHostUnregister(addr + pc[0])
                return rc
            rcs = list(pool.map(reg, pieces))
            if any(rcs):
                print(json.dumps({"rank": rank, "error": "hipHostRegister failed"}), flush=True)
                return
        else:
            cap.read_into(k0, cnt, ring[i % 3].numpy(), pool, a.threads, how=a.mode)
        done += cnt
        i += 1
        if time.perf_counter() - t0 >= a.seconds and i >= 3:
            break
    el = time.perf_counter() - t0
    cpu = time.process_time() - c0
    print(json.dumps({"rank": rank, "node_cpus": len(mine), "cpis": done, "seconds": el, "cpu_s": cpu,
                      "GBps": done * n * R.BYTES_PER_SAMPLE / el / 1e9}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--modes", default="memmove,pread,mapped")
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--cpis", type=int, default=768)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "replay_ranks.json"))
    # worker
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--mode", default="memmove")
    ap.add_argument("--capture", default="")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--start-at", type=float, default=0.0)
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    import bench
    import tools.replay_bench as RB
    (dmin, dmax, fmin, fmax, fs, n), desc = bench.CONFIGS[a.config]
    path = f"/dev/shm/blah2_ranks_{a.config}.rspduo"
    st = os.statvfs("/dev/shm")
    need = a.cpis * n * 8
    if st.f_bavail * st.f_frsize < need * 1.1:
        a.cpis = max(96, int(st.f_bavail * st.f_frsize * 0.8 / (n * 8)) // a.batch * a.batch)
    RB.make_capture(path, n, a.cpis, fs)
    nodes = node_cpus()
    res = {"config": a.config, "workload": desc, "capture_cpis": a.cpis, "capture_GB": a.cpis * n * 8 / 1e9, "batch": a.batch,
           "reader_threads_per_rank": a.threads, "numa_nodes": len(nodes), "cpus_per_node_allowed": [len(c) for c in nodes],
           "host_cores": os.cpu_count(), "cgroup_cpu_quota": cgroup_quota(), "link_GBps_per_gpu": 57.0, "runs": []}
    for mode in a.modes.split(","):
        for world in [int(v) for v in a.ranks.split(",")]:
            start_at = time.time() + 4.0 + 1.5 * world  # every rank has imported torch and filled its ring by then
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", "--rank", str(r), "--world", str(world),
                                       "--mode", mode, "--capture", path, "--n", str(n), "--start-at", str(start_at),
                                       "--seconds", str(a.seconds), "--threads", str(a.threads), "--batch", str(a.batch)],
                                      stdout=subprocess.PIPE, text=True, cwd=ROOT) for r in range(world)]
            outs = []
            for p in procs:
                o, _ = p.communicate(timeout=300)
                for line in o.splitlines():
                    if line.startswith("{"):
                        outs.append(json.loads(line))
            ok = [o for o in outs if "GBps" in o]
            run = {"mode": mode, "ranks": world, "ranks_reported": len(ok)}
            if ok:
                agg = sum(o["GBps"] for o in ok)
                cpis = sum(o["cpis"] for o in ok)
                run.update(aggregate_GBps=agg, per_rank_GBps=[round(o["GBps"], 2) for o in sorted(ok, key=lambda o: o["rank"])],
                           frac_of_ranks_x_link=agg / (world * res["link_GBps_per_gpu"]),
                           host_cpu_s_per_cpi=sum(o["cpu_s"] for o in ok) / max(cpis, 1),
                           host_cpus_busy=sum(o["cpu_s"] / o["seconds"] for o in ok),
                           seconds=max(o["seconds"] for o in ok), cpis=cpis)
            else:
                run["error"] = outs
            print(json.dumps(run), flush=True)
            res["runs"].append(run)
    os.unlink(path)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
