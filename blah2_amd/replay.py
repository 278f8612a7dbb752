"""Offline replay of .rspduo captures, CPI-sharded across GPUs.

The reference replays a capture by pushing int16 I1 Q1 I2 Q2 samples into the
two IqData FIFOs (src/capture/rspduo/RspDuo.cpp:150-179) and the processing
thread cuts non-overlapping CPIs of nSamples out of them (blah2.cpp:254-258):
CPI k is exactly file bytes [k*nSamples*8, (k+1)*nSamples*8).  Ambiguity,
WienerHopf, CfarDetector1D and Map::set_metrics carry no state from one CPI to
the next, so whole CPIs shard round-robin over the ranks with NO data-path
collective; only the per-CPI results (two doubles + the detection list) are
gathered to rank 0, which emits them in file order.

`processor` is any callable ``(int16 array [B, nSamples, 4]) -> list of B
result dicts``; :func:`gpu_processor` builds the real one on the HIP engine.
The sharding / gathering logic itself has no GPU dependency and is covered by
the world_size=2 gloo tests.
"""
from __future__ import annotations

import json
import os
import sys
from typing import Callable, List, Optional

import numpy as np

BYTES_PER_SAMPLE = 8  # int16 I1 Q1 I2 Q2


class RspduoFile:
    """Memory-mapped .rspduo capture, indexed by CPI."""

    def __init__(self, path: str, n_samples: int):
        self.path = path
        self.n_samples = int(n_samples)
        size = os.path.getsize(path)
        self.n_cpis = size // (self.n_samples * BYTES_PER_SAMPLE)
        self._mm = np.memmap(path, dtype="<i2", mode="r") if size else np.zeros(0, dtype="<i2")

    def cpi(self, k: int) -> np.ndarray:
        if not 0 <= k < self.n_cpis:
            raise IndexError(k)
        a = self._mm[k * self.n_samples * 4:(k + 1) * self.n_samples * 4]
        return np.asarray(a).reshape(self.n_samples, 4)

    def batch(self, ks) -> np.ndarray:
        return np.stack([self.cpi(k) for k in ks]) if len(ks) else np.zeros((0, self.n_samples, 4), dtype=np.int16)


def shard_cpis(n_cpis: int, rank: int, world: int) -> List[int]:
    """Round-robin: rank r owns CPIs r, r+world, r+2*world, ..."""
    return list(range(rank, n_cpis, world))


def replay(capture: RspduoFile, processor: Callable, batch: int = 1, dist=None,
           limit: Optional[int] = None) -> Optional[List[dict]]:
    """Processes every CPI of ``capture`` exactly once across the ranks of
    ``dist`` (a torch.distributed module with an initialised default group, or
    None for a single process).  Returns the per-CPI results in file order on
    rank 0 and None elsewhere."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    n = capture.n_cpis if limit is None else min(limit, capture.n_cpis)
    mine = shard_cpis(n, rank, world)
    local = []
    for i in range(0, len(mine), batch):
        ks = mine[i:i + batch]
        out = processor(capture.batch(ks))
        if len(out) != len(ks):
            raise RuntimeError("processor returned a different number of results than CPIs")
        for k, r in zip(ks, out):
            local.append(dict(r, cpi=k))
    if dist is None:
        return local
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    # a counter every rank agrees on (the throughput figure's numerator)
    import torch
    cnt = torch.tensor([len(local)], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        cnt = cnt.cuda()
    dist.all_reduce(cnt)
    if int(cnt.item()) != n:
        raise RuntimeError(f"replay processed {int(cnt.item())} CPIs, expected {n}")
    if rank != 0:
        return None
    merged = sorted((r for part in gathered for r in part), key=lambda r: r["cpi"])
    assert [r["cpi"] for r in merged] == list(range(n))
    return merged


def gpu_processor(cfg: dict, device: int = 0, batch: int = 1):
    """Clutter filter (optional) -> ambiguity -> metrics -> CFAR on the HIP engine,
    with the keys of the reference's config.yml ``process`` section."""
    import torch

    import blah2_amd
    amb_c, det_c, clu_c = cfg["ambiguity"], cfg.get("detection", {}), cfg.get("clutter", {})
    fs, n = int(cfg["fs"]), int(cfg["n_samples"])
    amb = blah2_amd.Ambiguity(amb_c["delayMin"], amb_c["delayMax"], amb_c["dopplerMin"], amb_c["dopplerMax"],
                              fs, n, True, device=device, max_batch=batch)
    wh = None
    if clu_c.get("enable", False):
        wh = blah2_amd.WienerHopf(clu_c["delayMin"], clu_c["delayMax"], n, device=device, max_batch=batch)
    cfar = None
    if det_c.get("enable", False):
        cfar = blah2_amd.CfarDetector1D(det_c["pfa"], det_c["nGuard"], det_c["nTrain"], det_c["minDelay"],
                                        det_c["minDoppler"])
    dev = torch.device("cuda", device)

    def run(iq: np.ndarray):
        B = iq.shape[0]
        st = torch.cuda.current_stream(dev).cuda_stream
        if wh is None:
            d = torch.from_numpy(np.ascontiguousarray(iq)).to(dev)
            amb.process_dev(blah2_amd.FMT_I16, d.data_ptr(), 0, B, n, None, None, st)
        else:
            f = torch.from_numpy(np.ascontiguousarray(iq)).to(dev).to(torch.float32)
            x = torch.view_as_complex(f[..., 0:2].contiguous())
            y = torch.view_as_complex(f[..., 2:4].contiguous())
            wh.process_dev(x.data_ptr(), y.data_ptr(), B, n, y.data_ptr(), None, st)
            amb.process_dev(blah2_amd.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, None, None, st)
        out = []
        for b in range(B):
            m = amb.read_last(b)
            r = {"noisePower": m.noisePower, "maxPower": m.maxPower}
            if cfar is not None:
                det = cfar.process(m)
                r.update(delay=det.get_delay().tolist(), doppler=det.get_doppler().tolist(),
                         snr=det.get_snr().tolist())
            out.append(r)
        return out

    return run


def main(argv=None):
    import argparse

    import yaml
    ap = argparse.ArgumentParser(description="CPI-sharded replay of a .rspduo capture on MI355X")
    ap.add_argument("capture")
    ap.add_argument("-c", "--config", required=True, help="blah2 config.yml")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--limit", type=int, default=None)
    a = ap.parse_args(argv)
    y = yaml.safe_load(open(a.config))
    fs = int(y["capture"]["fs"])
    n = int(fs * float(y["process"]["data"]["cpi"]))  # blah2.cpp:142-144
    cfg = dict(y["process"], fs=fs, n_samples=n)
    dist = None
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist_
        torch.cuda.set_device(local)
        dist_.init_process_group("nccl")
        dist = dist_
    res = replay(RspduoFile(a.capture, n), gpu_processor(cfg, local, a.batch), a.batch, dist, a.limit)
    if res is not None:
        for r in res:
            sys.stdout.write(json.dumps(r) + "\n")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
