"""CPU suite: the CPI sharding / gather logic of the replay path under a real
world_size=2 torch.distributed group (gloo), with a stub processor (the HIP
engine has no CPU implementation, and the sharding has no GPU dependency)."""
import os
import socket

import numpy as np
import pytest

from blah2_amd import replay as R

N_SAMPLES = 257


def make_capture(path, n_cpis, tail=5):
    rng = np.random.default_rng(99)
    a = rng.integers(-2000, 2000, size=(n_cpis * N_SAMPLES + tail, 4), dtype=np.int16)
    a.tofile(path)
    return a


def stub(iq):
    # order-revealing per-CPI summary
    return [{"noisePower": float(np.abs(c[:, 0].astype(np.float64)).mean()), "maxPower": float(c[:, 3].max())}
            for c in iq]


def test_rspduo_file_layout(tmp_path):
    p = str(tmp_path / "a.rspduo")
    a = make_capture(p, 5)
    f = R.RspduoFile(p, N_SAMPLES)
    assert f.n_cpis == 5  # the ragged tail is not a CPI
    assert np.array_equal(f.cpi(3), a[3 * N_SAMPLES:4 * N_SAMPLES])
    with pytest.raises(IndexError):
        f.cpi(5)


@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (5, 2), (8, 2), (7, 3), (3, 8)])
def test_shards_partition_all_cpis(n, world):
    parts = [R.shard_cpis(n, r, world) for r in range(world)]
    assert sorted(k for p in parts for k in p) == list(range(n))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_replay(tmp_path):
    p = str(tmp_path / "a.rspduo")
    make_capture(p, 5)
    f = R.RspduoFile(p, N_SAMPLES)
    res = R.replay(f, stub, batch=2)
    assert [r["cpi"] for r in res] == [0, 1, 2, 3, 4]
    assert res[3]["noisePower"] == stub(f.batch([3]))[0]["noisePower"]


def _worker(rank, world, port, path, n_cpis, batch, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = R.replay(R.RspduoFile(path, N_SAMPLES), stub, batch=batch, dist=dist)
        if rank == 0:
            np.save(out_path, np.array([[r["cpi"], r["noisePower"], r["maxPower"]] for r in res]))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cpis,batch", [(7, 2), (1, 1), (4, 3)])
def test_two_rank_replay_gloo(tmp_path, n_cpis, batch):
    import torch.multiprocessing as mp
    p = str(tmp_path / "b.rspduo")
    make_capture(p, n_cpis)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, port, p, n_cpis, batch, out), nprocs=2, join=True)
    got = np.load(out)
    f = R.RspduoFile(p, N_SAMPLES)
    want = stub(f.batch(list(range(n_cpis))))
    assert got[:, 0].tolist() == list(range(n_cpis))            # file order, each CPI once
    assert np.allclose(got[:, 1], [w["noisePower"] for w in want])
    assert np.allclose(got[:, 2], [w["maxPower"] for w in want])


@pytest.mark.parametrize("n,batch,world", [(0, 2, 2), (1, 2, 2), (9, 2, 2), (8, 4, 2), (7, 3, 3), (5, 1, 8)])
def test_batches_partition_all_cpis_contiguously(n, batch, world):
    parts = [R.shard_batches(n, batch, r, world) for r in range(world)]
    flat = sorted(b for p in parts for b in p)
    assert [k for k0, c in flat for k in range(k0, k0 + c)] == list(range(n))
    assert all(c == batch for _, c in flat[:-1])  # only the last batch may be short
    for r, p in enumerate(parts):  # batch b on rank b % world
        assert all((k0 // batch) % world == r for k0, _ in p)


def test_single_process_replay_streams_in_order(tmp_path):
    """emit() is called per CPI in file order WHILE later batches are still unprocessed (bounded memory, like the
    reference's per-CPI sends, blah2.cpp:299-321)."""
    p = str(tmp_path / "a.rspduo")
    make_capture(p, 7)
    f = R.RspduoFile(p, N_SAMPLES)
    calls = []

    def counting(iq):
        calls.append(len(iq))
        return stub(iq)

    seen = []
    n = R.replay(f, counting, batch=2, emit=lambda r: seen.append((r["cpi"], len(calls))))
    assert n == 7 and [c for c, _ in seen] == list(range(7))
    assert seen[0][1] == 1 and seen[2][1] == 2 and seen[6][1] == 4  # CPI 0 is out after the first batch, not after the last


def test_read_into_matches_the_memory_map(tmp_path):
    from concurrent.futures import ThreadPoolExecutor
    p = str(tmp_path / "a.rspduo")
    a = make_capture(p, 6)
    f = R.RspduoFile(p, N_SAMPLES)
    for how in ("memmove", "pread"):
        dst = np.zeros((4, N_SAMPLES, 4), dtype=np.int16)
        with ThreadPoolExecutor(3) as pool:
            f.read_into(1, 3, dst, pool, parts=3, how=how)
        assert np.array_equal(dst[:3].reshape(-1, 4), a[N_SAMPLES:4 * N_SAMPLES]) and not dst[3].any()
        dst[:] = 0
        f.read_into(2, 1, dst, how=how)  # one piece, no pool
        assert np.array_equal(dst[0].reshape(-1, 4), a[2 * N_SAMPLES:3 * N_SAMPLES]) and not dst[1:].any()
    f.close()


def _stream_worker(rank, world, port, path, n_cpis, batch, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def counting(iq):
            calls.append(len(iq))
            return stub(iq)

        seen = []
        res = R.replay(R.RspduoFile(path, N_SAMPLES), counting, batch=batch, dist=dist,
                       emit=lambda r: seen.append((r["cpi"], r["noisePower"], len(calls))))
        if rank == 0:
            assert res == n_cpis
            np.save(out_path, np.array(seen))
        else:
            assert res is None and not seen
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("addr,nbytes,parts", [
    (0, 16000000, 4), (16000000, 16000000, 4), (4096 * 7 + 3, 5, 4), (4096 * 7 + 4090, 10, 2), (4096 * 3, 4096 * 5, 8),
    (4096 * 3 + 1, 4096 * 5, 3), (1000, 0, 4), (8 * 2000000 * 5, 8 * 2000000 * 16, 4), (12345, 4096 * 2 - 1, 1),
])
def test_page_split_covers_the_window_once_with_whole_pages(addr, nbytes, parts):
    """The zero-copy read path registers whole pages of the mapped capture and sends the ragged ends through a small pinned
    buffer: head + pieces + tail tile [addr, addr + nbytes) exactly, the pieces start and end on page boundaries, the ends
    stay under a page, and two neighbouring windows never register the same page."""
    head, pieces, tail = R.page_split(addr, nbytes, parts)
    runs = [head] + pieces + [tail]
    pos = 0
    for o, ln in runs:
        assert o == pos or ln == 0
        assert ln >= 0
        pos = o + ln if ln else pos
    assert pos == nbytes
    assert head[1] < R.PAGE and tail[1] < R.PAGE and len(pieces) <= max(parts, 1)
    for o, ln in pieces:
        assert (addr + o) % R.PAGE == 0 and ln % R.PAGE == 0 and ln > 0
    # the next window's pieces start at or after this one's last registered page
    h2, p2, _ = R.page_split(addr + nbytes, nbytes, parts)
    if pieces and p2:
        assert addr + pieces[-1][0] + pieces[-1][1] <= addr + nbytes + p2[0][0]


def test_window_is_the_mapped_file(tmp_path):
    n = 1000
    rng = np.random.default_rng(3)
    data = rng.integers(-2048, 2047, (7, n, 4), dtype=np.int16)
    path = str(tmp_path / "w.rspduo")
    data.tofile(path)
    cap = R.RspduoFile(path, n)
    import ctypes
    for k0, cnt in [(0, 7), (2, 3), (6, 1)]:
        addr, nbytes = cap.window(k0, cnt)
        assert nbytes == cnt * n * 8
        got = np.frombuffer((ctypes.c_char * nbytes).from_address(addr), dtype=np.int16).reshape(cnt, n, 4)
        assert np.array_equal(got, data[k0:k0 + cnt])
    cap.close()


def test_two_rank_streaming_gather_gloo(tmp_path):
    """world_size 2, per-round gather: rank 0 emits CPIs 0..3 (round 0 = batch 0 of rank 0 + batch 1 of rank 1) after ONE
    of its own batches, in file order, and the whole capture exactly once."""
    import torch.multiprocessing as mp
    n_cpis, batch = 11, 2
    p = str(tmp_path / "c.rspduo")
    make_capture(p, n_cpis)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "seen.npy")
    mp.spawn(_stream_worker, args=(2, port, p, n_cpis, batch, out), nprocs=2, join=True)
    seen = np.load(out)
    f = R.RspduoFile(p, N_SAMPLES)
    want = stub(f.batch(list(range(n_cpis))))
    assert seen[:, 0].tolist() == list(range(n_cpis))
    assert np.allclose(seen[:, 1], [w["noisePower"] for w in want])
    assert seen[0, 2] == 1 and seen[3, 2] == 1 and seen[4, 2] == 2  # round g is out after rank 0's g+1-th batch


def _worker_serialise(rank, world, port, path, n_cpis, batch, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def processor(iq):  # a result with a large member that must never travel
            return [dict(r, map=np.full(300_000, 7.0)) for r in stub(iq)]

        def serialise(r):  # runs where the CPI was processed: the document leaves, the map stays
            return {"cpi": r["cpi"], "by": rank, "doc": "x" * (1000 + 37 * r["cpi"]) + f"{r['noisePower']:.6f}"}

        seen = []
        stats = {}
        n = R.replay(R.RspduoFile(path, N_SAMPLES), processor, batch=batch, dist=dist, emit=seen.append,
                     serialise=serialise, stats=stats)
        assert stats["cpis_owned"] == sum(c for _, c in R.shard_batches(n_cpis, batch, rank, world))
        if rank == 0:
            assert n == n_cpis
            np.save(out_path, np.array([[r["cpi"], r["by"], len(r["doc"]), float(r["doc"][1000 + 37 * r["cpi"]:])] for r in seen]))
            assert all("map" not in r for r in seen)
        else:
            assert n is None and not seen
    finally:
        dist.destroy_process_group()


def test_results_are_serialised_on_the_rank_that_owns_them(tmp_path):
    """--json replay: Map::to_json runs where the CPI was processed and rank 0 forwards documents (blah2.cpp:299-321);
    the gather moves byte tensors, ragged per rank, in file order."""
    import torch.multiprocessing as mp
    n_cpis, batch = 11, 2
    p = str(tmp_path / "c.rspduo")
    make_capture(p, n_cpis)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "ser.npy")
    mp.spawn(_worker_serialise, args=(2, port, p, n_cpis, batch, out), nprocs=2, join=True)
    got = np.load(out)
    f = R.RspduoFile(p, N_SAMPLES)
    assert got[:, 0].tolist() == list(range(n_cpis))
    for k in range(n_cpis):
        assert got[k, 1] == (k // batch) % 2          # batch b is rank b % world's
        assert got[k, 2] == 1000 + 37 * k + len(f"{stub(f.batch([k]))[0]['noisePower']:.6f}")
        assert abs(got[k, 3] - stub(f.batch([k]))[0]["noisePower"]) < 1e-5


def test_numa_pinning_is_a_no_op_without_a_device():
    """pin_to_device_node / device_numa_cpus (replay.py): without a GPU (this suite) or without sysfs topology there is nothing
    to pin to, and the process's affinity must be left alone."""
    import torch
    before = os.sched_getaffinity(0)
    assert R.device_numa_cpus(torch, 0) is None or isinstance(R.device_numa_cpus(torch, 0), set)
    if not torch.cuda.is_available():
        assert R.device_numa_cpus(torch, 0) is None
        assert R.pin_to_device_node(torch, 0) is False
        assert os.sched_getaffinity(0) == before


def _worker_world8(rank, world, port, path, n_cpis, batch, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def counting(iq):
            calls.append(len(iq))
            return stub(iq)

        def serialise(r):
            return {"cpi": r["cpi"], "by": rank, "noisePower": r["noisePower"]}

        seen = []
        stats = {}
        n = R.replay(R.RspduoFile(path, N_SAMPLES), counting, batch=batch, dist=dist, serialise=serialise, stats=stats,
                     emit=lambda r: seen.append((r["cpi"], r["by"], r["noisePower"], len(calls))))
        mine = R.shard_batches(n_cpis, batch, rank, world)
        assert calls == [c for _, c in mine]                       # this rank processed exactly its own batches, in order
        assert stats["cpis_owned"] == sum(c for _, c in mine)
        np.save(os.path.join(out_dir, f"owned_{rank}.npy"), np.array([stats["cpis_owned"], len(calls)]))
        if rank == 0:
            assert n == n_cpis
            np.save(os.path.join(out_dir, "seen.npy"), np.array(seen))
        else:
            assert n is None and not seen
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cpis,batch", [(37, 2), (5, 1)])
def test_eight_rank_replay_with_idle_ranks_gloo(tmp_path, n_cpis, batch):
    """The shape the 8-GPU node runs (blah2.cpp:254-258 cut into batches over 8 ranks), as 8 gloo PROCESSES: 37 CPIs in
    batches of 2 are 19 batches = two full rounds + a ragged third (ranks 0..2 busy, the last batch one CPI short,
    ranks 3..7 idle in it); 5 single-CPI batches leave ranks 5..7 with NOTHING for the whole capture.  Every rank takes part
    in every round's gather, rank 0 emits each CPI once, in file order, round by round while later rounds are unprocessed."""
    import torch.multiprocessing as mp
    world = 8
    p = str(tmp_path / "w8.rspduo")
    make_capture(p, n_cpis)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_world8, args=(world, port, p, n_cpis, batch, str(tmp_path)), nprocs=world, join=True)
    seen = np.load(str(tmp_path / "seen.npy"))
    f = R.RspduoFile(p, N_SAMPLES)
    want = stub(f.batch(list(range(n_cpis))))
    assert seen[:, 0].tolist() == list(range(n_cpis))                      # each CPI once, in file order
    assert np.allclose(seen[:, 2], [w["noisePower"] for w in want])
    assert seen[:, 1].tolist() == [(k // batch) % world for k in range(n_cpis)]   # serialised by the rank that owns the batch
    # streaming: a CPI of round g is emitted after rank 0's (g+1)-th own batch (or its last, in rounds where it has none)
    n_batches = -(-n_cpis // batch)
    own0 = len(R.shard_batches(n_cpis, batch, 0, world))
    for k in range(n_cpis):
        g = (k // batch) // world
        assert seen[k, 3] == min(g + 1, own0), (k, seen[k, 3])
    owned = [np.load(str(tmp_path / f"owned_{r}.npy")) for r in range(world)]
    assert sum(int(o[0]) for o in owned) == n_cpis
    idle = [r for r in range(world) if int(owned[r][1]) == 0]
    assert idle == list(range(min(n_batches, world), world))                # ranks beyond the batch count never work
    if n_batches % world:
        last_round_busy = n_batches % world
        assert all(int(owned[r][1]) == n_batches // world + (1 if r < last_round_busy else 0) for r in range(world))
