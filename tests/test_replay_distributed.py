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
