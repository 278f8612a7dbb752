"""CPU: the rank/launch logic of `bench.py --gpus N` (no GPU, no torch.distributed needed)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_single_gpu_runs_inline():
    assert bench.plan_launch(1, {}, 1) == ("inline", 1)
    assert bench.plan_launch(1, {}, 8) == ("inline", 1)


def test_multi_gpu_without_torchrun_spawns_one_rank_per_gpu():
    assert bench.plan_launch(8, {}, 8) == ("spawn", 8)
    assert bench.plan_launch(2, {}, 8) == ("spawn", 2)


def test_more_ranks_than_devices_is_refused():
    action, msg = bench.plan_launch(2, {}, 1)
    assert action == "error" and "2 ranks requested, 1 device(s) visible" in msg
    assert bench.plan_launch(1, {}, 0)[0] == "error"
    assert bench.plan_launch(0, {}, 4)[0] == "error"


def test_under_torchrun_world_size_must_equal_gpus():
    env = {"WORLD_SIZE": "4", "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1"}
    assert bench.plan_launch(4, env, 8) == ("inline", 4)
    action, msg = bench.plan_launch(8, env, 8)
    assert action == "error" and "WORLD_SIZE=4" in msg
    # a rank whose LOCAL_RANK has no device
    assert bench.plan_launch(4, dict(env, LOCAL_RANK="3"), 2)[0] == "error"
    # torchrun with one rank is still a distributed run of size 1
    assert bench.plan_launch(1, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, 1) == ("inline", 1)


def test_bench_refuses_loudly_on_this_gpu_less_host():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "2 ranks requested, 0 device(s) visible" in (r.stderr + r.stdout)


def test_free_port_is_bindable():
    import socket
    p = bench.free_port()
    with socket.socket() as s:
        s.bind(("127.0.0.1", p))
