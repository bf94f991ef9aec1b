#!/usr/bin/env python
"""Benchmark of the Wavenet training hot path (forward + Gini gradient + backpropagation + batch
reduction + filter update) on MI355X, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME|all|headline] [--scaling weak|strong]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path over one batch resident in HBM.  Rank 0 prints one JSON line per
workload; the LAST line is the headline, the configuration the headline metric is quoted on: 2D 64x64
calorimeter-style images, batch 4096 per GPU, 4-tap filter.  On one GPU the default run first prints a
line each for the other configurations (64x64 with 16 taps, 1D 1024 / 16 taps, the per-GPU slices of
512x512 / batch 16384 over 8 GPUs and of 2^20-sample signals / 32 taps); with more ranks only the headline
is run unless --workload says otherwise.
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
    # name: (R, C, per-GPU batch, taps, generator, seconds of CPU baseline)
    "2d64": (64, 64, 4096, 4, "calorimeter", 15.0),
    "2d64n16": (64, 64, 4096, 16, "calorimeter", 6.0),
    "1d1024": (1024, 1, 4096, 16, "bumps", 6.0),
    "2d512": (512, 512, 2048, 4, "calorimeter", 8.0),    # streamed path; 16384 images over 8 GPUs
    "2d512n16": (512, 512, 2048, 16, "calorimeter", 5.0),
    "1d1m": (1 << 20, 1, 8, 32, "bumps", 6.0),           # streamed path, 8 signals per GPU
    "1d1mb64": (1 << 20, 1, 64, 32, "bumps", 6.0),       # the same with 64 signals per GPU: enough work per launch
}
HEADLINE = "2d64"
ALL_ORDER = ["2d64n16", "1d1024", "2d512", "2d512n16", "1d1m", "1d1mb64", HEADLINE]
# Distinct resident batches the timed steps rotate over, so that more than the 256 MiB Infinity Cache lies between two
# uses of an input (64x64: 3 x 134 MB); a batch of 512x512 images (4.3 GB) is beyond it by itself.
ROTATE = {"2d64": 3, "2d64n16": 3, "1d1024": 10, "2d512": 1, "2d512n16": 1, "1d1m": 6, "1d1mb64": 1}

# What actually limits the dominant kernel (the graded roofline fraction prices the step in the bytes an
# unfused pipeline would move; these kernels keep most of those passes on chip).
LIMITER = {
    "2d64": "VALU/LDS issue: the 4096-key bitonic sort of the Gini ranks is about half of the kernel's cycles",
    "2d64n16": "VALU issue (filter cascade and sort) with one wave per SIMD",
    "1d1024": "VALU/LDS issue at two waves per SIMD: the sort's exchanges through LDS and the filter cascade",
    "2d512": "HBM passes of the four line-transform kernels; the per-bucket LDS sort of the ranking (instruction issue, barriers)",
    "2d512n16": "vector issue of the two backward line kernels (one and two waves per SIMD), then the ranking",
    "1d1m": "kernel count and tail latency: 8 signals per step give each launch little work",
    "1d1mb64": "HBM passes of the level-by-level transform (32 taps) and of the ranking",
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


def cpu_baseline(X, a0, lam, alpha, budget_s):
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

    def run(count):
        w = orc.Wavenet(lam=lam, alpha=alpha)
        w.set_batch_size(max(count, 2))
        w.set_filter(a0)
        t0 = time.perf_counter()
        for e in range(count):
            w.train(X[e])
        if count < 2:
            w.flush()
        return time.perf_counter() - t0

    per = run(1)                                   # also builds the weight matrices once, as the reference does
    probe = int(max(1, min(8, len(X), 0.2 * budget_s / max(per, 1e-9))))
    per = run(probe) / probe
    count = int(max(probe, min(len(X), budget_s / max(per, 1e-9))))
    dt = run(count)
    return {"value": count / dt, "unit": "examples/s", "cores": 1, "kind": "port",
            "sample": f"{count} of the benchmark's examples through the dense restatement "
                      f"(matrix operators rebuilt per update, plain loops instead of BLAS), {dt:.1f} s"}


def exchange_probe(ctx, dX, B, R, C, lam, alpha, steps, fused_ms):
    """Times wn_train_batch + one-rank wn_allreduce_sums + wn_apply_update per step against the fused wn_train_step
    (fused_ms), on one GPU.  The difference is the cost of splitting the step around the collective -- what stays
    on the serial path of every batch when the sums have to travel -- and bounds the weak-scaling efficiency."""
    import torch
    # (RCCL prints a version banner on standard output when it starts: sent to standard error, the JSON lines stay alone)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        ctx.comm_init(ctx.comm_unique_id(), 1, 0)
    except Exception as exc:  # no RCCL on this box: say so instead of failing the bench
        return {"error": str(exc)}
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one():
        ctx.train_batch(dX, B, R, C)
        ctx.allreduce_sums()
        ctx.apply_update(B, lam * B, alpha, 0.0, True)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        one()
    ev1.record()
    torch.cuda.synchronize()
    wall_ms = 1e3 * (time.perf_counter() - t0) / steps
    split_ms = ev0.elapsed_time(ev1) / steps
    over_us = 1e3 * (max(split_ms, wall_ms) - fused_ms)
    return {"split_step_ms_device": split_ms, "split_step_ms_wall": wall_ms, "fused_step_ms": fused_ms,
            "exchange_overhead_us": over_us,
            "implied_weak_scaling_bound": fused_ms / max(split_ms, wall_ms, fused_ms),
            "note": "one-rank RCCL all-reduce of N+2 doubles between wn_train_batch and wn_apply_update on one GPU; "
                    "with more ranks the collective's latency over xGMI comes on top"}


def coach_cpp_probe(R, C, N, B, a0):
    """Coach::run of the C++ host layer (tests/cpp/host_driver bench), as a child process: the reference's own
    training entry point (src/Coach.cxx:71-296) on a device-resident set of 8 batches, 6 epochs, and on batches drawn
    from the on-device generator; and the reference's usual regime of batch 8 with the adaptive learning rate."""
    import subprocess
    import tempfile
    driver = os.path.join(ROOT, "tests", "cpp", "host_driver")
    if not os.path.exists(driver):
        return {"error": "tests/cpp/host_driver not built"}
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        np.asarray(a0, dtype=np.float64).tofile(f)
        path = f.name
    out = {}
    try:
        for key, cfg in (("resident_batch%d" % B, [B, 8 * B, 6, 0, "resident"]),
                         ("generated_batch%d" % B, [B, 8 * B, 6, 0, "generated"]),
                         ("adaptive_batch8", [8, 4096, 4, 1, "resident"])):
            cmd = [driver, "bench", R, C, N] + cfg + ["calorimeter", path]
            res = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=300)
            if res.returncode != 0:
                out[key] = {"error": res.stderr[-300:]}
                continue
            line = json.loads(res.stdout.strip().splitlines()[-1])
            out[key] = {"examples_per_s": line["coach_run_examples_per_s"], "updates": line["updates"],
                        "events_per_epoch": line["events_per_epoch"], "epochs": line["epochs"]}
    finally:
        os.unlink(path)
    return out


def load_traffic(workload):
    """HBM bytes per step from the committed rocprofv3 PMC passes (profiles/traffic.json), if present."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


def run_workload(name, args, rank, local_rank, world, dist):
    import torch
    from wavenet_amd import Context
    from wavenet_amd.filters import daubechies, perturbed
    from wavenet_amd.generators import to_device_layout
    from wavenet_amd.parallel import DataParallelTrainer, shard_range

    R, C, B, N, gen, cpu_budget = WORKLOADS[name]
    if args.taps:
        N = args.taps
    if args.batch:
        B = args.batch
    if args.scaling == "strong":
        # the global batch is what one GPU takes in the weak form; every rank trains its contiguous share
        global_batch = B
        first, last = shard_range(global_batch, rank, world)
        B = last - first
    else:
        global_batch = B * world
        first = rank * B
    lam, alpha = 10.0, 1e-3
    a0 = perturbed(daubechies(N), 1e-2, seed=7)

    ctx = Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    X = None
    seed = 1000
    nrot = max(1, args.rotate if args.rotate else ROTATE.get(name, 1))
    batches = []
    for k in range(nrot):
        # batch k of the rotation continues the sequence of examples behind the ranks' shares of batch k - 1
        firstk = first + k * global_batch
        if gen == "calorimeter":
            # generated on the device (bit-identical to generators.calorimeter; tests/test_gpu_generators.py);
            # example numbers continue across ranks, so the ranks hold disjoint parts of one sequence
            dXk = torch.empty(max(B, 1) * R * C, dtype=torch.float64, device="cuda")
            for e0 in range(0, B, 32768):
                ctx.generate_calorimeter(dXk[e0 * R * C:], min(32768, B - e0), R, C, seed=seed, first=firstk + e0)
        else:
            Xk = make_inputs(gen, firstk + B, R, C, seed=seed)[firstk:]
            if k == 0:
                X = Xk
            dXk = torch.from_numpy(to_device_layout(Xk)).cuda()
        batches.append(dXk)
    dX = batches[0]
    ctx.set_filter(a0)
    trainer = DataParallelTrainer(ctx, world)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, rotate):
        """(wall seconds, device ms per step) of `steps` steps bracketed by barrier + synchronize."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            trainer.step(batches[i % nrot] if rotate else dX, B, R, C, lam, alpha, B_total=global_batch)
        ev1.record()
        torch.cuda.synchronize()
        sync_all()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) / steps

    for i in range(args.warmup):
        trainer.step(batches[i % nrot], B, R, C, lam, alpha, B_total=global_batch)
    # device time of the timed region: a pair of events on the stream the kernels are launched on (the
    # context runs on torch's current stream)
    elapsed, device_ms = timed(args.steps, True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_filter = ctx.filter()
    # the same number of steps on ONE resident batch, which the Infinity Cache can hold on to: does the time move?
    single_ms = None
    if nrot > 1 and world == 1 and not args.no_single_batch:
        single_ms = timed(args.steps, False)[1]
    # What every rank of a multi-GPU run pays on the serial path of a batch, measured on this one GPU: the step as
    # three submissions with a one-rank RCCL all-reduce of the sums in the middle against the fused single submission.
    exchange = None
    if world == 1 and name == HEADLINE and not args.no_exchange_probe:
        exchange = exchange_probe(ctx, dX, B, R, C, lam, alpha, args.steps, device_ms)

    sums = trainer.sums()
    nonfinite = float(sums[-1])

    out = None
    if rank == 0:
        value = global_batch * args.steps / elapsed
        kavg = device_ms * 1e-3
        achieved = B * algorithmic_bytes(R, C) / kavg / 1e9
        traffic = load_traffic(name) if not (args.taps or args.batch) and args.scaling == "weak" else None
        streamed = R * C > 4096
        kernel = ("streamed path: all kernels of one wn_train_step" if streamed else
                  "k_train64" if (R == 64 and C == 64) else
                  "k_train1d1024" if (R == 1024 and C == 1) else "k_train")
        out = {
            "metric": f"training examples/sec (fwd+backprop+update), {'2D' if C > 1 else '1D'} {R}x{C}",
            "value": value, "unit": "examples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{'2D' if C > 1 else '1D'} Wavenet, {R}x{C} {gen} inputs, {N}-tap filter, "
                                   f"batch {B} on this GPU ({global_batch} over {world}), Coach batch loop (lambda {lam}, alpha {alpha})",
                       "name": name, "shape": [R, C], "taps": N, "batch_per_gpu": B, "global_batch": global_batch,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel + (" (one launch per step: training, batch reduction and update)" if not streamed and world == 1 else ""),
                         "kernel_ms": device_ms,
                         "algorithmic_bytes_per_launch": B * algorithmic_bytes(R, C),
                         "measured_hbm_GBs": (traffic / kavg / 1e9) if traffic else None,
                         "limiter": LIMITER.get(name)},
            "nonfinite": nonfinite,
            "final_filter": [float(v) for v in final_filter],
            "resident_batches_rotated": nrot,
            "resident_bytes": nrot * B * R * C * 8,
        }
        if single_ms is not None:
            out["ms_per_step_one_resident_batch"] = single_ms
        if exchange is not None:
            out["exchange_overhead"] = exchange
        if world == 1 and not args.no_cpu_baseline:
            if X is None:
                # the same sequence of examples, on the host: no more of them than the sample can use
                X = make_inputs(gen, min(B, 4096, max(4, (1 << 23) // (R * C))), R, C, seed=seed)
            out["cpu_baseline"] = cpu_baseline(X, a0, lam, alpha, cpu_budget * args.cpu_scale)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if world == 1 and name == HEADLINE and not args.no_coach_probe:
            out["coach_cpp"] = coach_cpp_probe(R, C, N, B, a0)
    ctx.close()
    del dX, batches
    torch.cuda.empty_cache()
    return out


def workload_names(args, world):
    if args.workload == "all" or (args.workload == "default" and world == 1 and not (args.taps or args.batch)):
        return ALL_ORDER
    if args.workload in ("default", "headline"):
        return [HEADLINE]
    return [args.workload]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="default", choices=sorted(WORKLOADS) + ["all", "headline", "default"],
                    help="default: every configuration on one GPU (headline last), the headline alone on several")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the per-GPU batch is fixed; strong: the global batch is fixed and split over the ranks")
    ap.add_argument("--taps", type=int, default=0, help="filter length N (default: per workload)")
    ap.add_argument("--batch", type=int, default=0, help="examples per GPU per step (weak) / in all (strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rotate", type=int, default=0, help="distinct resident batches the steps rotate over (default: per workload)")
    ap.add_argument("--no-exchange-probe", action="store_true", help="skip the split-step / one-rank all-reduce timing of the headline")
    ap.add_argument("--no-single-batch", action="store_true", help="skip the extra timing loop on one resident batch (counter passes: every step of the run is then a rotating one)")
    ap.add_argument("--no-coach-probe", action="store_true", help="skip the C++ Coach::run timing of the headline")
    ap.add_argument("--cpu-scale", type=float, default=1.0, help="scales the seconds spent on the CPU baselines")
    ap.add_argument("--dry-run", action="store_true", help="print what each rank would run (no GPU is touched)")
    ap.add_argument("--rehearse", action="store_true",
                    help="several ranks on ONE GPU (every rank on device 0, sums exchanged by gloo): the multi-rank code path "
                         "of this script where only one GPU is at hand; the rate it prints is not a scaling measurement")
    args = ap.parse_args()

    from wavenet_amd.parallel import dist_env, shard_range

    rank, local_rank, world = dist_env()
    if args.gpus != world:
        if world == 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                     f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py ...")
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        plan = []
        for name in workload_names(args, world):
            B = args.batch or WORKLOADS[name][2]
            if args.scaling == "strong":
                lo, hi = shard_range(B, rank, world)
                plan.append({"name": name, "first": lo, "batch": hi - lo, "global_batch": B})
            else:
                plan.append({"name": name, "first": rank * B, "batch": B, "global_batch": B * world})
        print(json.dumps({"rank": rank, "world": world, "scaling": args.scaling, "plan": plan}), flush=True)
        return
    import torch
    device_index = 0 if args.rehearse else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))

    for name in workload_names(args, world):
        line = run_workload(name, args, rank, device_index, world, dist)
        if rank == 0:
            if args.rehearse:
                line["rehearsal"] = f"{world} ranks on one GPU, gloo exchange: not a scaling measurement"
            print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
