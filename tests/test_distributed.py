"""Sharding over ranks: world_size-2 gloo on CPU (the GPU node runs the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cosim_amd import rng as crng
from cosim_amd.distributed import MetricsAccumulator, shard_range


def test_shard_ranges_partition_the_fleet():
    for total, world in ((4096, 8), (4097, 8), (10, 3), (7, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_random_streams_do_not_depend_on_sharding():
    full = crng.uniform(5, np.arange(64), 3, crng.PURPOSE_MASS, 2)
    parts = [crng.uniform(5, np.arange(*shard_range(64, r, 4)), 3, crng.PURPOSE_MASS, 2) for r in range(4)]
    np.testing.assert_array_equal(full, np.concatenate(parts))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(10, rank, world)
    acc = MetricsAccumulator(["a", "b"])
    vals = torch.arange(10, dtype=torch.float64)[:, None] * torch.tensor([[1.0, -2.0]])
    for _ in range(3):
        acc.update(vals[lo:hi])
    out = acc.reduce()
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_metrics_all_reduce_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = np.arange(10.0)
    assert out["a"]["count"] == 30 and out["a"]["mean"] == pytest.approx(x.mean()) and out["a"]["std"] == pytest.approx(x.std())
    assert out["b"]["mean"] == pytest.approx(-2 * x.mean()) and out["b"]["std"] == pytest.approx(2 * x.std())


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a torchrun environment: the parent spawns the two ranks (torch.distributed.run) before
    touching any GPU, rank 0 prints the one JSON line (skeleton only: --selftest-launch does no GPU work)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["max_rank"] == 1.0
    assert lines[0]["count"] == 8 and lines[0]["mean"] == pytest.approx(1.5)


def test_bench_launcher_command_and_rank_mismatch():
    import importlib.util
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    cmd = b.relaunch_command(["--gpus", "4", "--steps", "7"], 4, 12345)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert "127.0.0.1" in cmd and "12345" in cmd and cmd[cmd.index("12345") + 1].endswith("bench.py")
