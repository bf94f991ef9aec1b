"""The exchange step of the data-parallel loop on one GPU: the C ABI's RCCL communicator
(wn_comm_unique_id / wn_comm_init / wn_allreduce_sums) and torch.distributed's "nccl" backend, each with a
single rank -- the sums must pass through unchanged and the update must match the plain one.  More ranks
are covered on the CPU with gloo (tests/test_distributed_cpu.py); 2, 4 and 8 GPUs are the driver's runs.
"""
import numpy as np
import pytest

import filters as F
from gpu_util import to_dev

pytestmark = pytest.mark.gpu


def _inputs(B=12, R=64, C=64, seed=3):
    rng = np.random.default_rng(seed)
    X = np.where(rng.random((B, R, C)) < 0.9, 0.0, rng.random((B, R, C)))
    X[:, 0, 0] += 0.25
    return X


def test_rccl_communicator_of_one_rank():
    import torch
    from wavenet_amd import Context
    X = _inputs()
    B, R, C = X.shape
    a = F.perturbed(F.daubechies(4), 1e-2, 1)
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_filter(a)
        ctx.train_batch(to_dev(X), B, R, C)
        before = ctx.sums().copy()
        ctx.comm_init(ctx.comm_unique_id(), 1, 0)
        ctx.allreduce_sums()
        after = ctx.sums()
        assert np.array_equal(before, after)
        ctx.apply_update(B, B * 10.0, 1e-3, 0.0, True)
        f1 = ctx.filter()
    finally:
        ctx.close()          # destroys the communicator with the context
    ref = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        ref.set_filter(a)
        ref.train_batch(to_dev(X), B, R, C)
        ref.apply_update(B, B * 10.0, 1e-3, 0.0, True)
        assert np.array_equal(f1, ref.filter())
    finally:
        ref.close()


def test_trainer_through_a_nccl_group_of_one_rank():
    """DataParallelTrainer with its all-reduce forced on (world size 2 declared, group of one rank, half the
    batch "elsewhere" contributing nothing): the sums tensor the context writes into goes through
    torch.distributed's nccl backend and comes back unchanged."""
    import torch
    import torch.distributed as dist
    from wavenet_amd import Context
    from wavenet_amd.parallel import DataParallelTrainer, allreduce_sums
    X = _inputs(B=8)
    B, R, C = X.shape
    a = F.perturbed(F.daubechies(4), 1e-2, 2)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29641", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_filter(a)
        tr = DataParallelTrainer(ctx, 1)
        ctx.train_batch(to_dev(X), B, R, C)
        torch.cuda.synchronize()
        before = tr.sums().copy()
        n = len(before)
        allreduce_sums(tr._buf[:n], 2)          # world size > 1: the collective runs
        torch.cuda.synchronize()
        assert np.array_equal(before, tr.sums())
        tr.step(to_dev(X), B, R, C, 10.0, 1e-3)
        assert np.all(np.isfinite(ctx.filter()))
    finally:
        ctx.close()
        if created:
            dist.destroy_process_group()


def test_trainer_follows_torchs_current_stream():
    """A step() issued under `torch.cuda.stream(s)` runs the context's kernels on `s`, the stream the
    all-reduce would be ordered on: the trainer rebinds the context when the current stream has changed."""
    import torch
    from wavenet_amd import Context
    from wavenet_amd.parallel import DataParallelTrainer
    X = _inputs(B=8)
    B, R, C = X.shape
    a = F.perturbed(F.daubechies(4), 1e-2, 2)
    dX = to_dev(X)
    ctx, ref = Context(0), Context(0, stream=torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream()
    try:
        ctx.set_filter(a)
        ref.set_filter(a)
        tr = DataParallelTrainer(ctx, 1)
        assert tr._stream == torch.cuda.current_stream().cuda_stream
        tr.step(dX, B, R, C, 10.0, 1e-3)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            tr.step(dX, B, R, C, 10.0, 1e-3)
            assert tr._stream == side.cuda_stream
        tr.step(dX, B, R, C, 10.0, 1e-3)          # back on the first stream; wn_set_stream waits for `side`
        assert tr._stream == torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            ref.train_step(dX, B, R, C, B, 10.0 * B, 1e-3, 0.0, True)
        assert np.array_equal(ctx.filter(), ref.filter())
    finally:
        ctx.close()
        ref.close()


def test_single_process_communicators_of_one_gpu():
    """wn_comm_init_all / wn_allreduce_sums_all (one host thread driving every GPU, as the host layer's data-parallel
    Coach does) with the one GPU there is: the grouped all-reduce returns the sums unchanged, wn_train_step on a
    context that holds a communicator lands on the plain update's filter, and two contexts on one GPU are refused."""
    import ctypes as C
    import torch
    from wavenet_amd import Context
    from wavenet_amd.capi import WN_ERR_ARG, WN_OK
    X = _inputs()
    B, R, Cc = X.shape
    a = F.perturbed(F.daubechies(4), 1e-2, 1)
    dX = to_dev(X)
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    other = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ref = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        arr2 = (C.c_void_p * 2)(ctx._h, other._h)
        assert ctx.lib.wn_comm_init_all(arr2, 2) == WN_ERR_ARG
        arr1 = (C.c_void_p * 1)(ctx._h)
        assert ctx.lib.wn_comm_init_all(arr1, 1) == WN_OK
        for c in (ctx, ref):
            c.set_filter(a)
        ctx.train_batch(dX, B, R, Cc)
        before = ctx.sums().copy()
        assert ctx.lib.wn_allreduce_sums_all(arr1, 1) == WN_OK
        assert np.array_equal(before, ctx.sums())
        for _ in range(3):
            ctx.train_step(dX, B, R, Cc, B, B * 10.0, 1e-3, 0.0, True)
            ref.train_batch(dX, B, R, Cc)
            ref.apply_update(B, B * 10.0, 1e-3, 0.0, True)
        assert np.array_equal(ctx.filter(), ref.filter())
    finally:
        ctx.close()
        other.close()
        ref.close()
