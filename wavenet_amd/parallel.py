"""Data-parallel training over the GPUs of one node: one process per GPU, examples of a batch
sharded in contiguous ranges, one all-reduce of N + 2 doubles per batch (RCCL via
torch.distributed's "nccl" backend), identical update on every replica.

The only coupling between examples in the reference is the batch mean of the gradients and of
the costs (src/Wavenet.cxx:723-727, 733), so the exchange is the sum of
[gradient sum (N), sparse-term sum, non-finite count].
"""
import os


def dist_env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total, rank, world_size):
    """Contiguous [begin, end) of `total` examples owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(total, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def allreduce_sums(sums, world_size, group=None):
    """In-place sum over ranks of the per-rank sums tensor (any device torch.distributed supports)."""
    if world_size > 1:
        import torch.distributed as dist
        if sums.is_cuda and dist.get_backend(group) == "gloo":
            # gloo reduces host memory: over the host and back, in stream order (the copy out waits for the kernels
            # that wrote the sums, the copy back is queued before whatever reads them)
            host = sums.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            sums.copy_(host)
        else:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


class DataParallelTrainer:
    """Drives wn_train_batch -> all-reduce -> wn_apply_update for one rank.

    The all-reduce is ordered on torch's current stream, so the context's kernels have to run on that
    stream too, or the collective could read the sums before the batch reduction has written them.
    The trainer therefore REBINDS `ctx` (wn_set_stream) to torch's current stream: once when it is made,
    and again in step() whenever the current stream differs from the one bound last (a step() under
    `torch.cuda.stream(s)` moves the context to `s`; wn_set_stream waits for the work on the former
    stream first).  The binding, like the sums buffer set here, stays with the context after the
    trainer is gone: a caller that goes on using the context on a stream of its own sets it again.
    """

    def __init__(self, ctx, world_size=1, group=None):
        import torch
        from .capi import WN_MAX_TAPS
        self.ctx = ctx
        self.world_size = world_size
        self.group = group
        self._stream = None
        self._bind_stream()
        self._buf = torch.zeros(WN_MAX_TAPS + 2, dtype=torch.float64, device="cuda")
        ctx.set_sums_buffer(self._buf)

    def _bind_stream(self):
        import torch
        stream = torch.cuda.current_stream().cuda_stream
        if stream != self._stream:
            self.ctx.set_stream(stream)
            self._stream = stream

    def step(self, X_dev, B_local, R, C, lam, alpha, inertia_eff=0.0, wavelet=True, B_total=None):
        """One batch: B_local examples here, B_total over all ranks (default B_local * world_size)."""
        if B_total is None:
            B_total = B_local * self.world_size
        self._bind_stream()
        if self.world_size == 1:
            # one submission; up to 4096 samples per example one kernel launch (wn_train_step)
            self.ctx.train_step(X_dev, B_local, R, C, B_total, lam * B_total, alpha, inertia_eff, wavelet)
            return
        self.ctx.train_batch(X_dev, B_local, R, C)
        if self.world_size > 1:
            n = len_sums(self.ctx)
            allreduce_sums(self._buf[:n], self.world_size, self.group)
        self.ctx.apply_update(B_total, lam * B_total, alpha, inertia_eff, wavelet)

    def sums(self):
        return self._buf[:len_sums(self.ctx)].cpu().numpy()


def len_sums(ctx):
    return ctx.sums_ptr()[1]
