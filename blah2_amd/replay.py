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


def gpu_processor(cfg: dict, device: int = 0, batch: int = 1, want_map: bool = False):
    """blah2.cpp:268-287 on the HIP engine for batches of CPIs, device resident from the int16 upload
    to the hit lists: clutter filter (optional) -> ambiguity -> metrics -> CFAR (blah2hip_cfar1d_dev),
    then Centroid and Interpolate (host arithmetic on a handful of detections) when ``nCentroid`` is
    configured.  ``cfg`` carries the keys of the reference's config.yml ``process`` section plus
    ``fs`` and ``n_samples``.  One D2H copy of {metrics, hit records, ok flags} per batch; the maps
    come back only with ``want_map`` (the --json mode needs them, blah2.cpp:304).

    A CPI whose clutter filter fails (normal equations not positive definite) is SKIPPED like the
    reference does (``if (!filter->process(x, y)) continue;`` blah2.cpp:270-273): its result is
    ``{"skipped": True}``."""
    import torch

    import blah2_amd
    amb_c, det_c, clu_c = cfg["ambiguity"], cfg.get("detection", {}), cfg.get("clutter", {})
    fs, n = int(cfg["fs"]), int(cfg["n_samples"])
    amb = blah2_amd.Ambiguity(amb_c["delayMin"], amb_c["delayMax"], amb_c["dopplerMin"], amb_c["dopplerMax"],
                              fs, n, True, device=device, max_batch=batch)
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    wh = None
    if clu_c.get("enable", False):
        wh = blah2_amd.WienerHopf(clu_c["delayMin"], clu_c["delayMax"], n, device=device, max_batch=batch)
    cfar = centroid = interp = None
    if det_c.get("enable", False):
        cfar = blah2_amd.CfarDetector1D(det_c["pfa"], det_c["nGuard"], det_c["nTrain"], det_c["minDelay"],
                                        det_c["minDoppler"])
        if "nCentroid" in det_c:  # blah2.cpp:176-181
            t_cpi = n / fs
            centroid = blah2_amd.Centroid(det_c["nCentroid"], det_c["nCentroid"], 1 / t_cpi)
            interp = blah2_amd.Interpolate(True, True)
    dev = torch.device("cuda", device)
    cap = int(det_c.get("capacity", min(nD * nC, 1 << 16)))
    out = torch.zeros((batch, nD, nC), dtype=torch.complex64, device=dev)
    met = torch.zeros((batch, 2), dtype=torch.float64, device=dev)
    okf = torch.ones(batch, dtype=torch.int32, device=dev)
    hits = torch.zeros((batch, cap, 2), dtype=torch.float64, device=dev)  # blah2hip_hit_t records, 16 bytes
    cnt = torch.zeros(batch, dtype=torch.int32, device=dev)

    def run(iq: np.ndarray):
        B = iq.shape[0]
        st = torch.cuda.current_stream(dev).cuda_stream
        d = torch.from_numpy(np.ascontiguousarray(iq)).to(dev)
        if wh is None:
            amb.process_dev(blah2_amd.FMT_I16, d.data_ptr(), 0, B, n, out.data_ptr(), met.data_ptr(), st)
        else:
            f = d.to(torch.float32)
            x = torch.view_as_complex(f[..., 0:2].contiguous())
            y = torch.view_as_complex(f[..., 2:4].contiguous())
            wh.process_dev(x.data_ptr(), y.data_ptr(), B, n, y.data_ptr(), okf.data_ptr(), st)
            amb.process_dev(blah2_amd.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
        if cfar is not None:
            cfar.process_dev(amb, B, hits.data_ptr(), cap, cnt.data_ptr(), out.data_ptr(), met.data_ptr(), st)
        # one synchronising round of copies per batch
        met_h = met[:B].cpu().numpy()
        ok_h = okf[:B].cpu().numpy() if wh is not None else np.ones(B, dtype=np.int32)
        need_map = want_map or interp is not None
        res = []
        if cfar is not None:
            cnt_h = cnt[:B].cpu().numpy()
            kmax = int(cnt_h.max()) if B else 0
            if kmax > cap:
                raise blah2_amd.Blah2HipError(blah2_amd._lib.ERR_CAPACITY, f"{kmax} detections in one CPI, capacity {cap}")
            hits_h = hits[:B, :max(kmax, 1)].cpu().numpy().view(blah2_amd.HIT_DTYPE).reshape(B, -1)
        map_h = out[:B].cpu().numpy() if need_map else None
        for b in range(B):
            if not ok_h[b]:
                res.append({"skipped": True})
                continue
            r = {"noisePower": float(met_h[b, 0]), "maxPower": float(met_h[b, 1])}
            if cfar is not None:
                det = blah2_amd.hits_to_detection(amb, hits_h[b], int(cnt_h[b]), cap)
                if centroid is not None:
                    det = centroid.process(det)
                    m = blah2_amd.Map(amb, map_h[b], amb.delay, amb.doppler, r["noisePower"], r["maxPower"], b)
                    det = interp.process(det, m)
                r.update(delay=det.get_delay().tolist(), doppler=det.get_doppler().tolist(),
                         snr=det.get_snr().tolist())
            if want_map:
                r["map"] = map_h[b]
            res.append(r)
        return res

    run.amb = amb
    return run


def frames_for(result: dict, amb, fs: int, timestamp: int):
    """The two JSON documents blah2.cpp sends per CPI (:304-317): the map (``Map::to_json`` +
    ``delay_bin_to_km``) and the detections (``Detection::to_json`` + ``delay_bin_to_km``), produced by
    the C++ host classes (libblah2host.so, include/blah2host.h).  A document is one TCP "frame": the
    Node API appends chunks until the buffer ends with ``}`` (api/server.js:123-136)."""
    from . import _hostlib as H
    docs = {"map": H.map_json(result["map"], amb.delay, amb.doppler, result["noisePower"], result["maxPower"], timestamp, fs)}
    if "delay" in result:
        docs["detection"] = H.detection_json(result["delay"], result["doppler"], result["snr"], timestamp, fs)
    return docs


MTU = 1024  # src/process/utility/Socket.cpp:5


def send_frame(sock, doc: str):
    """Socket::sendData (Socket.cpp:21-32): the document in MTU-sized writes, no terminator."""
    raw = doc.encode("ascii")
    for i in range(0, len(raw), MTU):
        sock.sendall(raw[i:i + MTU])


def main(argv=None):
    import argparse
    import socket

    import yaml
    ap = argparse.ArgumentParser(description="CPI-sharded replay of a .rspduo capture on MI355X")
    ap.add_argument("capture")
    ap.add_argument("-c", "--config", required=True, help="blah2 config.yml")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--json", action="store_true",
                    help="emit, per CPI in file order, the map and detection documents blah2.cpp sends (one per line on "
                         "stdout, or as TCP frames with --connect)")
    ap.add_argument("--connect", action="store_true",
                    help="with --json: send the documents to network.ip:ports.map / ports.detection of the config, "
                         "framed like Socket::sendData, instead of printing them")
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
    proc = gpu_processor(cfg, local, a.batch, want_map=a.json)
    res = replay(RspduoFile(a.capture, n), proc, a.batch, dist, a.limit)
    if res is not None:
        socks = {}
        if a.json and a.connect:
            ip = y["network"]["ip"]
            ip = "127.0.0.1" if ip == "0.0.0.0" else ip
            for name in ("map", "detection"):
                socks[name] = socket.create_connection((ip, int(y["network"]["ports"][name])))
        t_cpi_ms = int(round(1000.0 * n / fs))
        for r in res:
            if r.get("skipped"):
                continue
            if not a.json:
                sys.stdout.write(json.dumps(r) + "\n")
                continue
            # replay has no wall clock: CPI k is stamped k * tCpi in ms (blah2.cpp uses the capture time in ms)
            for name, doc in frames_for(r, proc.amb, fs, r["cpi"] * t_cpi_ms).items():
                if socks:
                    send_frame(socks[name], doc)
                else:
                    sys.stdout.write(doc + "\n")
        for s in socks.values():
            s.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
