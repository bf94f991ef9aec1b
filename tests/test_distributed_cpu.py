"""world_size 2 and 3 gloo jobs on the CPU covering the N > 1 path (no GPU needed)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_tiles():
    from wavenet_amd.parallel import shard_range
    for total in (0, 1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_data_parallel(world):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    for r in range(world):
        assert f"rank {r} OK" in res.stdout


@pytest.mark.parametrize("world,scaling", [(2, "strong"), (4, "strong"), (2, "weak")])
def test_bench_plan_under_torchrun(world, scaling):
    """bench.py as the driver launches it for N > 1, in its dry-run form (no GPU): every rank gets the headline
    workload; strong scaling splits the fixed global batch into contiguous shares, weak scaling keeps 4096 per rank."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--scaling", scaling, "--dry-run"]
    res = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    plans = sorted((json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [p["rank"] for p in plans] == list(range(world))
    shares = [p["plan"] for p in plans]
    assert all(len(s) == 1 and s[0]["name"] == "2d64" for s in shares)
    if scaling == "strong":
        assert sum(s[0]["batch"] for s in shares) == 4096 and all(s[0]["global_batch"] == 4096 for s in shares)
        assert [s[0]["first"] for s in shares] == [sum(t[0]["batch"] for t in shares[:r]) for r in range(world)]
    else:
        assert all(s[0]["batch"] == 4096 and s[0]["global_batch"] == 4096 * world for s in shares)


def test_bench_refuses_a_rank_count_mismatch():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "one process per GPU" in res.stderr
