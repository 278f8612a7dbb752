"""Offline replay of .rspduo captures, CPI-sharded across GPUs.

The reference replays a capture by pushing int16 I1 Q1 I2 Q2 samples into the
two IqData FIFOs (src/capture/rspduo/RspDuo.cpp:150-179) and the processing
thread cuts non-overlapping CPIs of nSamples out of them (blah2.cpp:254-258):
CPI k is exactly file bytes [k*nSamples*8, (k+1)*nSamples*8); each CPI's
products are emitted as soon as it is done, in order (blah2.cpp:299-321).
Ambiguity, WienerHopf, CfarDetector1D and Map::set_metrics carry no state from
one CPI to the next, so the capture shards over the ranks with NO data-path
collective.

The unit of work is a BATCH of `batch` consecutive CPIs (one contiguous read,
one launch of the device chain).  Batch b belongs to rank b mod world; a ROUND
is `world` consecutive batches.  Only the per-CPI results (two doubles + the
detection list, and the map when JSON is wanted) travel: after each round the
ranks' results are gathered to rank 0, which emits them in file order -- so
output streams while the capture is still being processed and every rank
holds one round of results at most.

On a rank the batches flow through a pipeline (:class:`GpuChain`): reader
threads copy the next batch out of the memory-mapped file into a ring of pinned
host buffers (or, ``read_mode="mapped"``, only register its pages with the
device: no CPU copy), a copy stream uploads batch k+1 and brings back the
results of batch k-1 while the kernels of batch k run on the compute stream;
HIP events order the three.
At 16 MB of int16 per 1 s CPI the GPU needs 9 us for what PCIe needs 290 us to
deliver: a replay is bound by the host link, and the pipeline's job is to keep
that link busy (tools/replay_bench.py measures both).

`processor` is a :class:`GpuChain`, or -- for the CPU tests of the sharding,
which has no GPU dependency -- any callable ``(int16 array [B, nSamples, 4])
-> list of B result dicts``.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time
import pickle
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np

BYTES_PER_SAMPLE = 8  # int16 I1 Q1 I2 Q2
PAGE = 4096
_MADV_POPULATE_READ = 22  # Linux 5.14+
_LIBC = C.CDLL(None, use_errno=True)
_LIBC.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]


class RspduoFile:
    """A .rspduo capture, indexed by CPI (memory-mapped for random access; :meth:`read_into` for streaming)."""

    def __init__(self, path: str, n_samples: int):
        self.path = path
        self.n_samples = int(n_samples)
        size = os.path.getsize(path)
        self.n_cpis = size // (self.n_samples * BYTES_PER_SAMPLE)
        self._mm = np.memmap(path, dtype="<i2", mode="r") if size else np.zeros(0, dtype="<i2")
        self._fd = None

    def cpi(self, k: int) -> np.ndarray:
        if not 0 <= k < self.n_cpis:
            raise IndexError(k)
        a = self._mm[k * self.n_samples * 4:(k + 1) * self.n_samples * 4]
        return np.asarray(a).reshape(self.n_samples, 4)

    def batch(self, ks) -> np.ndarray:
        return np.stack([self.cpi(k) for k in ks]) if len(ks) else np.zeros((0, self.n_samples, 4), dtype=np.int16)

    def read_into(self, k0: int, count: int, dst: np.ndarray, pool: Optional[ThreadPoolExecutor] = None, parts: int = 8,
                  how: str = "memmove"):
        """CPIs k0 .. k0+count-1 (contiguous in the file) into the first bytes of ``dst`` (any writable buffer, e.g. a
        pinned one), split over ``pool``'s threads.  ``how="memmove"``: out of the shared mapping -- each thread has the
        kernel fill the page tables of its piece (madvise MADV_POPULATE_READ: 47 GB/s per thread, against a fault per 64 KB)
        and copies it with memmove (non-temporal stores for large sizes): 12.5 GB/s per thread on the MI355X host;
        ``how="pread"``: the kernel's copy_to_user, 9.5 GB/s per thread.  PCIe takes 57."""
        nbytes = count * self.n_samples * BYTES_PER_SAMPLE
        off0 = k0 * self.n_samples * BYTES_PER_SAMPLE
        if how == "pread":
            if self._fd is None:
                self._fd = os.open(self.path, os.O_RDONLY)
            mv = memoryview(dst).cast("B")[:nbytes]

            def rd(a, b):
                pos = a
                while pos < b:
                    got = os.preadv(self._fd, [mv[pos:b]], off0 + pos)
                    if got <= 0:
                        raise IOError(f"short read from {self.path} at {off0 + pos}")
                    pos += got
        else:
            src = self._mm.ctypes.data + off0
            dstp = dst.ctypes.data
            if dst.nbytes < nbytes:
                raise ValueError("destination smaller than the CPIs asked for")

            def rd(a, b):
                lo = (src + a) // PAGE * PAGE  # madvise wants a page-aligned start; a failure only means the copy faults the pages in
                _LIBC.madvise(lo, src + b - lo, _MADV_POPULATE_READ)
                C.memmove(dstp + a, src + a, b - a)

        if pool is None or parts <= 1 or nbytes < (1 << 22):
            rd(0, nbytes)
            return
        step = -(-nbytes // parts)
        step += -step % 4096
        futs = [pool.submit(rd, a, min(a + step, nbytes)) for a in range(0, nbytes, step)]
        for f in futs:
            f.result()

    def window(self, k0: int, count: int) -> Tuple[int, int]:
        """(address, bytes) of CPIs k0 .. k0+count-1 inside the read-only shared mapping of the file: the page cache's own
        pages, which the zero-copy read path registers with the device and uploads from (``GpuChain(read_mode="mapped")``)."""
        nbytes = count * self.n_samples * BYTES_PER_SAMPLE
        off0 = k0 * self.n_samples * BYTES_PER_SAMPLE
        return self._mm.ctypes.data + off0, nbytes

    def close(self):
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None
        self._mm = np.zeros(0, dtype="<i2")  # the mapping goes with its last reference (anything registered of it must be released first)


class LoopedCapture(RspduoFile):
    """A capture read ``times`` times over, end to end, as ONE stream of ``times * n_cpis`` CPIs (throughput measurements
    that must last seconds from a file that fits /dev/shm: CPI k is CPI k mod n of the file).  A batch must not straddle
    the wrap: the file's CPI count has to be a multiple of the batch."""

    def __init__(self, path: str, n_samples: int, times: int):
        super().__init__(path, n_samples)
        self.n_file = self.n_cpis
        self.n_cpis = self.n_file * max(1, int(times))

    def _fold(self, k0: int, count: int) -> int:
        k = k0 % self.n_file
        if k + count > self.n_file:
            raise ValueError(f"batch [{k0}, {k0 + count}) straddles the end of a {self.n_file}-CPI capture read cyclically")
        return k

    def cpi(self, k: int) -> np.ndarray:
        if not 0 <= k < self.n_cpis:
            raise IndexError(k)
        k %= self.n_file
        return np.asarray(self._mm[k * self.n_samples * 4:(k + 1) * self.n_samples * 4]).reshape(self.n_samples, 4)

    def read_into(self, k0, count, dst, pool=None, parts=8, how="memmove"):
        return super().read_into(self._fold(k0, count), count, dst, pool, parts, how)

    def window(self, k0, count):
        return super().window(self._fold(k0, count), count)


def page_split(addr: int, nbytes: int, parts: int, page: int = PAGE):
    """The byte range [addr, addr + nbytes) as (head, [whole-page pieces], tail): ``head`` and ``tail`` are the ragged
    ends (offset, length) relative to ``addr`` that do not fill a page, the pieces are at most ``parts`` page-aligned
    (offset, length) runs between them.  The pieces are what gets registered with the device (two neighbouring
    batches never share a page that way); the ends, under a page each, travel through a small pinned buffer."""
    end = addr + nbytes
    lo = min(-(-addr // page) * page, end)
    hi = max(end // page * page, lo)
    head = (0, lo - addr)
    tail = (hi - addr, end - hi)
    pieces = []
    pages = (hi - lo) // page
    if pages > 0:
        parts = max(1, min(int(parts), pages))
        per = -(-pages // parts)
        for k in range(0, pages, per):
            cnt = min(per, pages - k)
            pieces.append((lo - addr + k * page, cnt * page))
    return head, pieces, tail


def shard_cpis(n_cpis: int, rank: int, world: int) -> List[int]:
    """Round-robin over single CPIs (batch = 1): rank r owns CPIs r, r+world, r+2*world, ..."""
    return list(range(rank, n_cpis, world))


def shard_batches(n_cpis: int, batch: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """(first CPI, count) of the batches rank ``rank`` owns: batch b = CPIs [b*batch, (b+1)*batch) goes to rank b % world."""
    n_batches = -(-n_cpis // batch) if n_cpis else 0
    return [(b * batch, min(batch, n_cpis - b * batch)) for b in range(rank, n_batches, world)]


def _iter_local(capture: RspduoFile, processor, mine) -> Iterator[List[dict]]:
    """Results of this rank's batches, one list per batch, in order."""
    if hasattr(processor, "run_batches"):  # the pipelined device chain
        yield from processor.run_batches(capture, mine)
        return
    for k0, cnt in mine:
        out = processor(capture.batch(range(k0, k0 + cnt)))
        if len(out) != cnt:
            raise RuntimeError("processor returned a different number of results than CPIs")
        yield [dict(r, cpi=k0 + i) for i, r in enumerate(out)]


def _host_group(dist):
    """A gloo group over the same ranks for host-resident bytes (the default group may be RCCL, whose tensors live in
    HBM): results leave the device on their owning rank, so the gather moves host buffers."""
    if dist.get_backend() == "gloo":
        return None  # the default group will do
    return dist.new_group(backend="gloo")


def _gather_bytes(dist, group, payload: bytes, rank: int, world: int) -> Optional[List[bytes]]:
    """One round's results as TENSORS: every rank's byte count first (one int64 each), then the payloads padded to the
    longest.  What travels is what the owning rank serialised; rank 0 receives buffers, it does not unpickle objects it
    then has to format."""
    import torch
    n = torch.tensor([len(payload)], dtype=torch.int64)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    longest = max(int(v.item()) for v in sizes)
    buf = torch.zeros(max(longest, 1), dtype=torch.uint8)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    parts = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0, group=group)
    if rank != 0:
        return None
    return [bytes(p[:int(sz.item())].numpy().tobytes()) for p, sz in zip(parts, sizes)]


def replay(capture: RspduoFile, processor, batch: int = 1, dist=None, limit: Optional[int] = None,
           emit: Optional[Callable[[dict], None]] = None, serialise: Optional[Callable[[dict], dict]] = None,
           stats: Optional[dict] = None) -> Optional[List[dict]]:
    """Processes every CPI of ``capture`` exactly once across the ranks of ``dist`` (a torch.distributed module with an
    initialised default group, or None for a single process).

    With ``emit``: rank 0 calls ``emit(result)`` for every CPI in file order, round by round, while later rounds are still
    being processed (bounded memory), and the function returns the number of CPIs on rank 0.  Without: the results are
    collected and returned as a list on rank 0 (tests, small captures).  Other ranks return None.

    ``serialise`` runs on the rank that OWNS a CPI, before the gather: it turns the processor's result (which may hold a
    16 MB map) into what rank 0 has to emit -- the JSON documents of blah2.cpp:299-321, say -- and must keep the "cpi"
    key.  Rank 0's work per foreign CPI is then a receive and a write.  ``stats`` (a dict) receives this rank's seconds
    spent serialising, in the gather and (rank 0) emitting."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    n = capture.n_cpis if limit is None else min(limit, capture.n_cpis)
    mine = shard_batches(n, batch, rank, world)
    n_batches = n_batches_hint(n, batch)
    n_rounds = -(-n_batches // world) if n_batches else 0
    collected: Optional[List[dict]] = [] if (emit is None and rank == 0) else None
    expect = 0
    done = 0
    t_ser = t_gather = t_emit = 0.0
    group = _host_group(dist) if dist is not None else None

    def deliver(results: List[dict]):
        nonlocal expect, done, t_emit
        t0 = time.perf_counter()
        for r in results:
            if r["cpi"] != expect:
                raise RuntimeError(f"replay order: CPI {r['cpi']} arrived where {expect} was due")
            expect += 1
            done += 1
            if emit is not None:
                emit(r)
            else:
                collected.append(r)
        t_emit += time.perf_counter() - t0

    local = _iter_local(capture, processor, mine)
    for g in range(n_rounds):
        own = next(local) if g * world + rank < n_batches else []
        if serialise is not None:
            t0 = time.perf_counter()
            own = [serialise(r) for r in own]
            t_ser += time.perf_counter() - t0
        if dist is None:
            deliver(own)
            continue
        t0 = time.perf_counter()
        parts = _gather_bytes(dist, group, pickle.dumps(own, protocol=pickle.HIGHEST_PROTOCOL) if own else b"", rank, world)
        t_gather += time.perf_counter() - t0
        if rank == 0:
            for part in parts:  # one round: `world` batches, rank order = file order
                deliver(pickle.loads(part) if part else [])
    if dist is not None:
        # a counter every rank agrees on (the throughput figure's numerator)
        import torch
        cnt = torch.tensor([sum(c for _, c in mine)], dtype=torch.int64)
        dist.all_reduce(cnt, group=group)
        if int(cnt.item()) != n:
            raise RuntimeError(f"replay processed {int(cnt.item())} CPIs, expected {n}")
    if stats is not None:
        stats.update(serialise_s=t_ser, gather_s=t_gather, emit_s=t_emit, cpis_owned=sum(c for _, c in mine))
    if rank != 0:
        return None
    if done != n:
        raise RuntimeError(f"replay emitted {done} CPIs, expected {n}")
    return done if emit is not None else collected


def n_batches_hint(n: int, batch: int) -> int:
    return -(-n // batch) if n else 0


def device_numa_cpus(torch, device: int = 0):
    """The CPUs of the NUMA node the GPU hangs off (sysfs, through the device's PCI address), or None when it cannot be told
    (one node, no sysfs, a container that hides it)."""
    try:
        bus = torch.cuda.get_device_properties(device).pci_bus_id  # torch >= 2.2
        dom = getattr(torch.cuda.get_device_properties(device), "pci_domain_id", 0)
        dev = getattr(torch.cuda.get_device_properties(device), "pci_device_id", 0)
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev:02x}.0"
        node = int(open(path + "/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        return cpus or None
    except Exception:  # no device, no sysfs, an older torch: nothing to pin to
        return None


def pin_to_device_node(torch, device: int = 0) -> bool:
    """Run this thread -- and the threads it starts, and the pinned memory it allocates from now on -- on the GPU's NUMA
    node: the reader threads' copies then stay inside one socket's memory (a two-socket host moves them over the
    inter-socket links otherwise).  True if the affinity was changed."""
    cpus = device_numa_cpus(torch, device)
    if not cpus or cpus == os.sched_getaffinity(0):
        return False
    os.sched_setaffinity(0, cpus)
    return True


def _hip_runtime(torch):
    """libamdhip64 of the running torch, for hipHostRegister / hipHostUnregister / hipMemcpyAsync on raw addresses (torch
    has no tensor over read-only memory); None if it cannot be loaded."""
    try:
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
        hip.hipHostUnregister.argtypes = [C.c_void_p]
        hip.hipSetDevice.argtypes = [C.c_int]
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hip.hipGetErrorString.restype = C.c_char_p
        return hip
    except OSError:
        return None


class GpuChain:
    """blah2.cpp:268-287 on the HIP engine for batches of CPIs, device resident from the int16 upload to the hit lists:
    clutter filter (optional; reads the .rspduo words directly) -> ambiguity -> metrics -> CFAR (blah2hip_cfar1d_dev),
    then Centroid and Interpolate (host arithmetic on a handful of detections) when ``nCentroid`` is configured.  ``cfg``
    carries the keys of the reference's config.yml ``process`` section plus ``fs`` and ``n_samples``.

    A ring of ``depth`` slots, each with a pinned host batch, its device copy, device result buffers and their pinned
    host copies.  For batch k: reader threads fill the slot's host buffer; the COPY stream uploads it; the COMPUTE stream
    (after the upload's event) runs the chain into the slot's result buffers; the copy stream (after the compute event)
    brings {metrics, ok flags, hit counts, the first ``hit_copy`` hit records per CPI, and the map when wanted} back.
    Batch k+1's read and upload and batch k-1's download overlap batch k's kernels.

    A CPI whose clutter filter fails (normal equations not positive definite) is SKIPPED like the reference does
    (``if (!filter->process(x, y)) continue;`` blah2.cpp:270-273): its result is ``{"skipped": True}``."""

    def __init__(self, cfg: dict, device: int = 0, batch: int = 1, want_map: bool = False, depth: int = 3,
                 reader_threads: int = 4, hit_copy: int = 4096, read_mode: str = "memmove", numa: bool = True):
        import torch

        import blah2_amd
        self.torch, self.b2 = torch, blah2_amd
        # The host side of the ring on the GPU's NUMA node (EPYC 9575F x 2, tools/gpu_hostreg.py: a reader thread copies
        # 17-19 GB/s inside the node, 12.5 when the scheduler puts it on the other socket; four threads 51-54 GB/s against
        # 41): the chain's OWN threads pin themselves when they start, and the pinned buffers are allocated (first touch)
        # with the calling thread on the node for the length of that allocation only -- the caller's affinity is what it
        # was when the constructor returns (a host program that builds chains for several devices, or has threads of its
        # own, is not narrowed to the last GPU's node).
        self._node_cpus = device_numa_cpus(torch, device) if numa else None
        self.numa_pinned = bool(self._node_cpus)
        with self._on_node():
            self._build(cfg, device, batch, want_map, depth, reader_threads, hit_copy, read_mode)

    def _pin_worker(self):
        if self._node_cpus:
            try:
                os.sched_setaffinity(0, self._node_cpus)  # the calling thread only
            except OSError:
                pass

    def _on_node(self):
        """The calling thread on the GPU's node for the length of a ``with`` block (pinned allocations), then back."""
        chain = self

        class _Ctx:
            def __enter__(self):
                self.prev = None
                if chain._node_cpus:
                    try:
                        self.prev = os.sched_getaffinity(0)
                        os.sched_setaffinity(0, chain._node_cpus)
                    except OSError:
                        self.prev = None

            def __exit__(self, *exc):
                if self.prev is not None:
                    os.sched_setaffinity(0, self.prev)
                return False
        return _Ctx()

    def _build(self, cfg, device, batch, want_map, depth, reader_threads, hit_copy, read_mode):
        torch, blah2_amd = self.torch, self.b2
        amb_c, det_c, clu_c = cfg["ambiguity"], cfg.get("detection", {}), cfg.get("clutter", {})
        self.fs, self.n = int(cfg["fs"]), int(cfg["n_samples"])
        n, B = self.n, int(batch)
        self.batch, self.depth, self.want_map = B, max(2, int(depth)), bool(want_map)
        self.amb = blah2_amd.Ambiguity(amb_c["delayMin"], amb_c["delayMax"], amb_c["dopplerMin"], amb_c["dopplerMax"],
                                       self.fs, n, True, device=device, max_batch=B)
        nD, nC = self.amb.get_n_doppler_bins(), self.amb.get_n_delay_bins()
        self.wh = None
        self.fused_fir = False
        if clu_c.get("enable", False):
            self.wh = blah2_amd.WienerHopf(clu_c["delayMin"], clu_c["delayMax"], n, device=device, max_batch=B)
            # the filter's FIR inside the range kernel where one 4096-point transform covers the geometry (range_fir_kernel):
            # the filtered channel never crosses HBM.  clutter: {fused: false} keeps the two-stage chain.
            self.fused_fir = bool(clu_c.get("fused", True)) and self.amb.fir_fusable(self.wh, blah2_amd.FMT_I16) is None
            if self.fused_fir:
                self.amb.set_fir(self.wh)
        self.cfar = self.centroid = self.interp = None
        if det_c.get("enable", False):
            self.cfar = blah2_amd.CfarDetector1D(det_c["pfa"], det_c["nGuard"], det_c["nTrain"], det_c["minDelay"],
                                                 det_c["minDoppler"])
            if "nCentroid" in det_c:  # blah2.cpp:176-181
                self.centroid = blah2_amd.Centroid(det_c["nCentroid"], det_c["nCentroid"], 1 / (n / self.fs))
                self.interp = blah2_amd.Interpolate(True, True)
        self.need_map = self.want_map or self.interp is not None
        dev = self.dev = torch.device("cuda", device)
        self.cap = int(det_c.get("capacity", min(nD * nC, 1 << 16)))
        self.hit_copy = min(self.cap, int(hit_copy))
        self.compute = torch.cuda.Stream(device=dev)
        self.copy = torch.cuda.Stream(device=dev)
        self.pool = ThreadPoolExecutor(max_workers=max(1, reader_threads), initializer=self._pin_worker)
        self._bg = ThreadPoolExecutor(max_workers=1, initializer=self._pin_worker)  # starts and awaits a batch's read while the main thread works on another
        self.reader_threads = max(1, reader_threads)
        # How a batch gets from the page cache to the copy engine (tools/gpu_hostreg.py, tools/replay_bench.py; MI355X host):
        #   "memmove"  reader threads copy out of the file's shared mapping into a pinned ring (page tables filled by madvise
        #              first): 17-19 GB/s per thread on the GPU's NUMA node (12.5 across sockets), FOUR threads keep the link
        #              at 0.905 of its pinned rate (57 GB/s); three passes over DRAM
        #   "pread"    the same through the kernel's copy_to_user: 12.6 GB/s per thread on the node (9.5): 0.76 with four threads
        #   "mapped"   no CPU copy and ONE pass over DRAM: the threads only REGISTER whole-page pieces of the mapping with the
        #              device (hipHostRegister) and the copy engine reads the page cache's own pages.  A third to a half of the
        #              host CPU time -- but the driver registers 4 KB pages at 17-25 GB/s whatever the thread count (12-15 GB/s
        #              with the page-table fill of a first pass), so a replay that touches every page once gets 0.60 of the
        #              link from one thread and less from more.  For captures replayed repeatedly out of a mapping that stays
        #              (warm page tables) it reaches 0.94 of it with one thread.
        if read_mode not in ("memmove", "pread", "mapped"):
            raise ValueError("read_mode: 'memmove', 'pread' or 'mapped'")
        self.read_mode = read_mode
        self._hip = _hip_runtime(torch) if read_mode == "mapped" else None
        if read_mode == "mapped" and self._hip is None:
            self.read_mode = "memmove"
        # the filtered surveillance channel (one buffer: the compute stream is in order)
        self.yf = torch.empty((B, n), dtype=torch.complex64, device=dev) if self.wh is not None and not self.fused_fir else None
        self.busy_ms, self.batches_done = 0.0, 0  # kernels' time on the compute stream / batches collected, since construction
        self.slots = []
        for _ in range(self.depth):
            s = {
                "h_iq": None,  # pinned staging batch of the pread path, allocated when that path first runs
                "h_ends": torch.empty(2 * PAGE, dtype=torch.uint8).pin_memory(),  # the ragged ends of a mapped batch
                "registered": [],
                "d_iq": torch.empty((B, n, 4), dtype=torch.int16, device=dev),
                "d_met": torch.zeros((B, 2), dtype=torch.float64, device=dev),
                "d_ok": torch.ones(B, dtype=torch.int32, device=dev),
                "d_hits": torch.zeros((B, self.cap, 2), dtype=torch.float64, device=dev),  # blah2hip_hit_t records, 16 bytes
                "d_cnt": torch.zeros(B, dtype=torch.int32, device=dev),
                "d_map": torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev),
                "h_met": torch.zeros((B, 2), dtype=torch.float64).pin_memory(),
                "h_ok": torch.ones(B, dtype=torch.int32).pin_memory(),
                "h_hits": torch.zeros((B, self.hit_copy, 2), dtype=torch.float64).pin_memory(),
                "h_cnt": torch.zeros(B, dtype=torch.int32).pin_memory(),
                "h_map": torch.zeros((B, nD, nC), dtype=torch.complex64).pin_memory() if self.need_map else None,
                "uploaded": torch.cuda.Event(), "downloaded": torch.cuda.Event(),
                # the two ends of the batch's kernels on the (in-order) compute stream, timed: their sum over a replay is the
                # time the GPU spent computing (`busy_ms`), the rest of the wall clock it waited for the host link
                "started": torch.cuda.Event(enable_timing=True), "computed": torch.cuda.Event(enable_timing=True),
            }
            self.slots.append(s)

    # -- the three stages of one batch -------------------------------------------------
    def _read(self, capture: RspduoFile, slot: dict, k0: int, cnt: int):
        slot["mapped"] = None
        if self.read_mode == "mapped":
            addr, nbytes = capture.window(k0, cnt)
            head, pieces, tail = page_split(addr, nbytes, self.reader_threads)
            hip, dev = self._hip, self.dev.index

            def register(pc):
                hip.hipSetDevice(dev)  # the pool's threads start on device 0
                return hip.hipHostRegister(addr + pc[0], pc[1], 0)

            rcs = list(self.pool.map(register, pieces))
            done = [pc for pc, rc in zip(pieces, rcs) if rc == 0]
            if len(done) == len(pieces):
                ends = slot["h_ends"].numpy()
                for (o, ln), at in ((head, 0), (tail, PAGE)):  # under a page each: a copy
                    if ln:
                        C.memmove(ends.ctypes.data + at, addr + o, ln)
                slot["registered"] = [addr + o for o, _ in pieces]
                slot["mapped"] = (addr, head, pieces, tail)
                return
            for o, _ in done:
                hip.hipHostUnregister(addr + o)
            self.read_mode = "memmove"  # this runtime / this file system does not register file mappings
            print(f"[blah2_amd.replay] hipHostRegister of the mapped capture failed ({hip.hipGetErrorString(max(rcs)).decode()}): "
                  "copying into a pinned buffer instead", file=sys.stderr)
        if slot["h_iq"] is None:
            with self._on_node():
                slot["h_iq"] = self.torch.empty((self.batch, self.n, 4), dtype=self.torch.int16).pin_memory()
        capture.read_into(k0, cnt, slot["h_iq"].numpy(), self.pool, self.reader_threads, how=self.read_mode)

    def _release(self, slot: dict):
        """The pieces of the mapping registered for this slot's batch (its upload has completed)."""
        for p in slot["registered"]:
            self._hip.hipHostUnregister(p)
        slot["registered"] = []

    def _submit(self, slot: dict, cnt: int):
        torch, b2, n, amb = self.torch, self.b2, self.n, self.amb
        with torch.cuda.stream(self.copy):
            if slot["mapped"] is not None:
                addr, head, pieces, tail = slot["mapped"]
                dst, st, hip = slot["d_iq"].data_ptr(), self.copy.cuda_stream, self._hip
                ends = slot["h_ends"].data_ptr()
                for src, (o, ln) in [(addr + o, (o, ln)) for o, ln in pieces] + [(ends, head), (ends + PAGE, tail)]:
                    if ln:
                        rc = hip.hipMemcpyAsync(dst + o, src, ln, 1, st)  # hipMemcpyHostToDevice
                        if rc:
                            raise RuntimeError(f"hipMemcpyAsync from the mapped capture: {hip.hipGetErrorString(rc).decode()}")
            else:
                slot["d_iq"][:cnt].copy_(slot["h_iq"][:cnt], non_blocking=True)
            slot["uploaded"].record(self.copy)
        with torch.cuda.stream(self.compute):
            self.compute.wait_event(slot["uploaded"])
            slot["started"].record(self.compute)
            st = self.compute.cuda_stream
            iq = slot["d_iq"].data_ptr()
            if self.wh is None:
                amb.process_dev(b2.FMT_I16, iq, 0, cnt, n, slot["d_map"].data_ptr(), slot["d_met"].data_ptr(), st)
            elif self.fused_fir:
                self.wh.estimate_dev_fmt(b2.FMT_I16, iq, None, cnt, n, slot["d_ok"].data_ptr(), st)
                amb.process_dev(b2.FMT_I16, iq, None, cnt, n, slot["d_map"].data_ptr(), slot["d_met"].data_ptr(), st)
            else:
                self.wh.process_dev_fmt(b2.FMT_I16, iq, None, cnt, n, self.yf.data_ptr(), n, slot["d_ok"].data_ptr(), st)
                amb.process_dev(b2.FMT_I16X_C32Y, iq, self.yf.data_ptr(), cnt, n, slot["d_map"].data_ptr(),
                                slot["d_met"].data_ptr(), st)
            if self.cfar is not None:
                self.cfar.process_dev(amb, cnt, slot["d_hits"].data_ptr(), self.cap, slot["d_cnt"].data_ptr(),
                                      slot["d_map"].data_ptr(), slot["d_met"].data_ptr(), st)
            slot["computed"].record(self.compute)
        with torch.cuda.stream(self.copy):
            self.copy.wait_event(slot["computed"])
            slot["h_met"][:cnt].copy_(slot["d_met"][:cnt], non_blocking=True)
            if self.wh is not None:
                slot["h_ok"][:cnt].copy_(slot["d_ok"][:cnt], non_blocking=True)
            if self.cfar is not None:
                slot["h_cnt"][:cnt].copy_(slot["d_cnt"][:cnt], non_blocking=True)
                slot["h_hits"][:cnt].copy_(slot["d_hits"][:cnt, :self.hit_copy], non_blocking=True)
            if self.need_map:
                slot["h_map"][:cnt].copy_(slot["d_map"][:cnt], non_blocking=True)
            slot["downloaded"].record(self.copy)

    def _collect(self, slot: dict, k0: int, cnt: int) -> List[dict]:
        b2, amb = self.b2, self.amb
        slot["downloaded"].synchronize()
        self.busy_ms += slot["started"].elapsed_time(slot["computed"])
        self.batches_done += 1
        met_h = slot["h_met"].numpy()
        ok_h = slot["h_ok"].numpy() if self.wh is not None else np.ones(cnt, dtype=np.int32)
        res = []
        for b in range(cnt):
            if not ok_h[b]:
                res.append({"skipped": True, "cpi": k0 + b})
                continue
            r = {"noisePower": float(met_h[b, 0]), "maxPower": float(met_h[b, 1]), "cpi": k0 + b}
            if self.cfar is not None:
                k = int(slot["h_cnt"][b])
                if k > self.cap:
                    raise b2.Blah2HipError(b2._lib.ERR_CAPACITY, f"{k} detections in one CPI, capacity {self.cap}")
                if k > self.hit_copy:  # rare: more hits than the pipelined copy carries -- fetch this CPI's records now
                    recs = slot["d_hits"][b, :k].cpu().numpy()
                else:
                    recs = slot["h_hits"][b, :max(k, 1)].numpy()
                det = b2.hits_to_detection(amb, recs.view(b2.HIT_DTYPE).reshape(-1), k, self.cap)
                if self.centroid is not None:
                    det = self.centroid.process(det)
                    m = b2.Map(amb, slot["h_map"][b].numpy(), amb.delay, amb.doppler, r["noisePower"], r["maxPower"], b)
                    det = self.interp.process(det, m)
                r.update(delay=det.get_delay().tolist(), doppler=det.get_doppler().tolist(), snr=det.get_snr().tolist())
            if self.want_map:
                r["map"] = slot["h_map"][b].numpy().copy()
            res.append(r)
        return res

    def run_batches(self, capture: RspduoFile, batches) -> Iterator[List[dict]]:
        """The pipeline over this rank's batches: yields each batch's results in order.  Up to ``depth`` batches are in
        flight; batch i's slot is reused by batch i + depth.  A slot has two halves with different lifetimes: its INPUT
        (the pinned batch, or the registered pages) is free as soon as the upload has completed, its RESULT buffers when the
        batch has been collected -- so the read of the next batch is started (on a background thread that drives the reader
        threads) the moment the previous read has ended, before this thread enqueues, synchronises, converts and yields:
        the reader threads never wait for Python."""
        batches = list(batches)
        D, n = self.depth, len(batches)
        reads = {}

        def start(j):
            slot = self.slots[j % D]
            slot["uploaded"].synchronize()  # the slot's previous upload (batch j - depth) is done with the input half
            self._release(slot)
            reads[j] = self._bg.submit(self._read, capture, slot, *batches[j])

        try:
            if n:
                start(0)
            for i in range(min(D - 1, n)):  # fill
                reads.pop(i).result()
                if i + 1 < n:
                    start(i + 1)
                self._submit(self.slots[i % D], batches[i][1])
            for i, (k0, cnt) in enumerate(batches):
                j = i + D - 1
                if j < n:
                    reads.pop(j).result()
                    if j + 1 < n:
                        start(j + 1)  # into the input half of the slot whose results are collected below
                    self._submit(self.slots[j % D], batches[j][1])
                yield self._collect(self.slots[i % D], k0, cnt)
        finally:
            for f in reads.values():  # a consumer that stopped early: let the read finish before its pages are released
                f.result()
            self.release_all()

    def __call__(self, iq: np.ndarray) -> List[dict]:
        """One batch, synchronously, from a host array [B, nSamples, 4] (tests; a caller that has the samples in memory)."""
        cnt = iq.shape[0]
        slot = self.slots[0]
        if slot["h_iq"] is None:
            with self._on_node():
                slot["h_iq"] = self.torch.empty((self.batch, self.n, 4), dtype=self.torch.int16).pin_memory()
        slot["h_iq"][:cnt].copy_(self.torch.from_numpy(np.ascontiguousarray(iq)))
        slot["mapped"] = None
        self._submit(slot, cnt)
        out = self._collect(slot, 0, cnt)
        for r in out:
            r.pop("cpi", None)
        return out

    def release_all(self):
        """Unregister whatever is still registered (before the capture's mapping goes away)."""
        self.torch.cuda.synchronize(self.dev)
        for slot in self.slots:
            self._release(slot)

    def close(self):
        self._bg.shutdown(wait=True)
        self.pool.shutdown(wait=True)
        self.release_all()


def gpu_processor(cfg: dict, device: int = 0, batch: int = 1, want_map: bool = False, **kw) -> GpuChain:
    """The device chain for :func:`replay` (kept under its old name)."""
    return GpuChain(cfg, device, batch, want_map, **kw)


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
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist_.init_process_group("nccl" if torch.cuda.device_count() >= int(os.environ["WORLD_SIZE"]) else "gloo")
        dist = dist_
    import torch
    proc = gpu_processor(cfg, local % max(1, torch.cuda.device_count()), a.batch, want_map=a.json)
    rank0 = dist is None or dist.get_rank() == 0
    socks = {}
    if rank0 and a.json and a.connect:
        ip = y["network"]["ip"]
        ip = "127.0.0.1" if ip == "0.0.0.0" else ip
        for name in ("map", "detection"):
            socks[name] = socket.create_connection((ip, int(y["network"]["ports"][name])))
    t_cpi_ms = int(round(1000.0 * n / fs))

    def serialise(r):  # on the rank that owns the CPI: Map::to_json + delay_bin_to_km there, documents on the wire
        if r.get("skipped") or not a.json:
            return r
        # replay has no wall clock: CPI k is stamped k * tCpi in ms (blah2.cpp uses the capture time in ms)
        return {"cpi": r["cpi"], "frames": frames_for(r, proc.amb, fs, r["cpi"] * t_cpi_ms)}

    def emit(r):  # rank 0, file order, as the rounds complete: a write per document
        if r.get("skipped"):
            return
        if not a.json:
            sys.stdout.write(json.dumps(r) + "\n")
            return
        for name, doc in r["frames"].items():
            if socks:
                send_frame(socks[name], doc)
            else:
                sys.stdout.write(doc + "\n")
        sys.stdout.flush()

    replay(RspduoFile(a.capture, n), proc, a.batch, dist, a.limit, emit=emit, serialise=serialise)
    for s in socks.values():
        s.close()
    proc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
