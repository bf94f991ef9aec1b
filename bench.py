#!/usr/bin/env python
"""Headline benchmark: training examples/s of the Wavenet hot path (forward + Gini gradient +
backpropagation + batch reduction + filter update) on MI355X, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path over one batch resident in HBM.  The default workload is the
configuration BASELINE.json's metric is quoted on: 2D 64x64 calorimeter-style images, batch 4096
per GPU (weak scaling), 4-tap filter.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec), GB/s

WORKLOADS = {
    # name: (R, C, per-GPU batch, default taps, generator)
    "2d64": (64, 64, 4096, 4, "calorimeter"),
    "1d1024": (1024, 1, 4096, 16, "bumps"),
    "2d512": (512, 512, 2048, 4, "calorimeter"),    # streamed path; 16384 images over 8 GPUs
    "1d1m": (1 << 20, 1, 8, 32, "bumps"),           # streamed path, 8 signals per GPU
}


def algorithmic_bytes(R, C):
    """Bytes per example when every API-visible operand of the training step crosses memory once
    per producing and once per consuming stage: 15 fp64 passes for 2D, 8 for 1D."""
    return 120 * R * C if (R > 1 and C > 1) else 64 * R * C


def make_inputs(gen, B, R, C, seed):
    from wavenet_amd import generators as G
    if gen == "calorimeter":
        return G.calorimeter(B, R, C, seed=seed)
    if gen == "bumps":
        return G.bumps(B, R * C, seed=seed).reshape(B, R, C)
    return G.uniform(B, R, C, seed=seed)


def cpu_baseline(X, a0, lam, alpha, budget_s=15.0):
    """Times the dense CPU restatement of the reference path (oracle/, one thread) on a bounded
    sample of the same examples: Wavenet::train per example, one batch update at the end."""
    from oracle import oracle as orc  # baseline leg only
    if max(X.shape[1], X.shape[2]) > 4096:
        # the dense operators of a line this long cannot be held (2^19 x 2^20 doubles = 4 TiB at 2^20
        # samples): time the banded restatement of the same arithmetic instead, and say so
        t0 = time.perf_counter()
        orc.band_train_batch(a0, X[:1])
        per = time.perf_counter() - t0
        count = int(max(1, min(len(X), budget_s / max(per, 1e-9))))
        t0 = time.perf_counter()
        orc.band_train_batch(a0, X[:count])
        dt = time.perf_counter() - t0
        return {"value": count / dt, "unit": "examples/s", "cores": 1, "kind": "port",
                "sample": f"{count} of the benchmark's examples through the BANDED restatement (forward, Gini gradient, "
                          f"backpropagation; the reference's dense operators would need 4 TiB at this length), {dt:.1f} s"}
    probe = 16
    w = orc.Wavenet(lam=lam, alpha=alpha)
    w.set_batch_size(1 << 30)
    w.set_filter(a0)
    t0 = time.perf_counter()
    for e in range(probe):
        w.train(X[e])
    per = (time.perf_counter() - t0) / probe
    count = int(max(probe, min(len(X), budget_s / max(per, 1e-9))))
    w = orc.Wavenet(lam=lam, alpha=alpha)
    w.set_batch_size(count)
    w.set_filter(a0)
    t0 = time.perf_counter()
    for e in range(count):
        w.train(X[e])
    dt = time.perf_counter() - t0
    return {"value": count / dt, "unit": "examples/s", "cores": 1, "kind": "port",
            "sample": f"{count} of the benchmark's examples through the dense restatement "
                      f"(matrix operators rebuilt per update, plain loops instead of BLAS), {dt:.1f} s"}


def load_traffic(workload, taps):
    """HBM bytes per training-kernel launch from the committed rocprofv3 PMC passes, if present."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(f"{workload}_n{taps}")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="2d64", choices=sorted(WORKLOADS))
    ap.add_argument("--taps", type=int, default=0, help="filter length N (default: per workload)")
    ap.add_argument("--batch", type=int, default=0, help="examples per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from wavenet_amd import Context
    from wavenet_amd.filters import daubechies, perturbed
    from wavenet_amd.generators import to_device_layout
    from wavenet_amd.parallel import DataParallelTrainer, dist_env

    rank, local_rank, world = dist_env()
    if args.gpus != world:
        if world == 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                     f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py ...")
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    R, C, B, N, gen = WORKLOADS[args.workload]
    if args.taps:
        N = args.taps
    if args.batch:
        B = args.batch
    lam, alpha = 10.0, 1e-3
    a0 = perturbed(daubechies(N), 1e-2, seed=7)

    ctx = Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    X = None
    if gen == "calorimeter":
        # generated on the device (bit-identical to generators.calorimeter; tests/test_gpu_generators.py)
        dX = torch.empty(B * R * C, dtype=torch.float64, device="cuda")
        for e0 in range(0, B, 32768):
            ctx.generate_calorimeter(dX[e0 * R * C:], min(32768, B - e0), R, C, seed=1000 + rank, first=e0)
    else:
        X = make_inputs(gen, B, R, C, seed=1000 + rank)
        dX = torch.from_numpy(to_device_layout(X)).cuda()
    ctx.set_filter(a0)
    trainer = DataParallelTrainer(ctx, world)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(dX, B, R, C, lam, alpha)
    ctx.profile_begin(args.steps)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(dX, B, R, C, lam, alpha)
    torch.cuda.synchronize()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    kernel_ms = ctx.profile_collect(args.steps)
    sums = trainer.sums()
    nonfinite = float(sums[-1])
    final_filter = ctx.filter()

    if rank == 0:
        value = world * B * args.steps / elapsed
        kavg = float(np.mean(kernel_ms)) * 1e-3
        achieved = B * algorithmic_bytes(R, C) / kavg / 1e9
        out = {
            "metric": f"training examples/sec (fwd+backprop+update), {'2D' if C > 1 else '1D'} {R}x{C}",
            "value": value, "unit": "examples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{'2D' if C > 1 else '1D'} Wavenet, {R}x{C} {gen} inputs, {N}-tap filter, "
                                   f"batch {B} per GPU, Coach batch loop (lambda {lam}, alpha {alpha})",
                       "shape": [R, C], "taps": N, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(args.workload, N),
                         "kernel": ("streamed path, all kernels of wn_train_batch" if R * C > 4096 else
                                    "k_train64" if (R == 64 and C == 64) else
                                    "k_train1d1024" if (R == 1024 and C == 1) else "k_train"), "kernel_ms": kavg * 1e3,
                         "algorithmic_bytes_per_launch": B * algorithmic_bytes(R, C)},
            "nonfinite": nonfinite,
            "final_filter": [float(v) for v in final_filter],
        }
        if world == 1 and not args.no_cpu_baseline:
            if X is None:
                # the same sequence of examples, on the host: no more of them than the sample can use
                X = make_inputs(gen, min(2 * B, 16384, max(64, (1 << 24) // (R * C))), R, C, seed=1000 + rank)
            out["cpu_baseline"] = cpu_baseline(X, a0, lam, alpha)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
