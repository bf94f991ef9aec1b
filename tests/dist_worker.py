"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-N gloo job on the CPU.

Checks the data-parallel driver logic (shard ranges, the all-reduce of N + 2 sums, identical update on
every replica) with the CPU oracle standing in for the per-rank kernel; the GPU path uses the same
wavenet_amd.parallel functions with wn_train_batch producing the per-rank sums.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import filters as F  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from wavenet_amd.parallel import allreduce_sums, dist_env, shard_range  # noqa: E402


def main():
    rank, _, world = dist_env()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, R, C, B, steps = 4, 8, 8, 22, 3   # 22 does not divide evenly: shards differ in size
    lam, alpha = 10.0, 1e-3
    rng = np.random.default_rng(5)
    X = rng.random((steps * B, R, C))
    a = F.perturbed(F.DB4, 1e-2, 3)

    ref = orc.Wavenet(lam=lam, alpha=alpha)   # single-process restatement of Wavenet::train
    ref.set_batch_size(B)
    ref.set_filter(a)
    for x in X:
        ref.train(x)
    ref_filters = ref.filter_log()[1:]
    ref_costs = ref.cost_log()[:-1]

    covered = []
    for s in range(steps):
        lo, hi = shard_range(B, rank, world)
        covered.append((lo, hi))
        g, ssum, bad = orc.band_train_batch(a, X[s * B + lo: s * B + hi])
        sums = torch.from_numpy(np.concatenate([g, [ssum, float(bad)]]))
        allreduce_sums(sums, world)
        sums = sums.numpy()
        gfull, sfull, _ = orc.band_train_batch(a, X[s * B:(s + 1) * B])
        assert np.allclose(sums[:N], gfull, rtol=1e-12, atol=1e-15), (rank, s)
        assert abs(sums[N] - sfull) <= 1e-12 * abs(sfull)
        assert sums[N + 1] == 0
        # identical update on every replica (Wavenet.cxx:717-765)
        gbar = (sums[:N] + B * lam * orc.reg_term_deriv(a)) / np.float64(np.float32(B))
        cost = (sums[N] + B * lam * orc.reg_term(a)) / np.float64(np.float32(B))
        a = a + (-alpha * gbar)
        assert np.allclose(a, ref_filters[s], rtol=1e-11, atol=0), (rank, s)
        assert abs(cost - ref_costs[s]) <= 1e-11 * abs(ref_costs[s])
    # replicas agree bit for bit
    t = torch.from_numpy(a.copy())
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    for other in gathered:
        assert torch.equal(other, gathered[0])
    # shard ranges tile [0, B)
    allr = [None] * world
    dist.all_gather_object(allr, covered[0])
    assert allr[0][0] == 0 and allr[-1][1] == B
    for i in range(1, world):
        assert allr[i][0] == allr[i - 1][1]
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} OK", flush=True)


if __name__ == "__main__":
    main()
