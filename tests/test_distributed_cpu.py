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
