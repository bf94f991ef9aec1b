"""wn_train_batch_static (the entry point that allocates nothing: partial sums in a fixed array of the
library's code object) against wn_train_batch, through the C ABI: same kernels, same grid, so the N + 2
sums must be bit-identical.  Two calls in a row per shape, the second with a smaller batch (fewer slots
of the shared array are written and read), then the first size again."""
import numpy as np
import pytest

import filters as F
from gpu_util import dev_empty, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,C,N,B1,B2", [(64, 64, 4, 1500, 37), (64, 64, 16, 300, 5), (1024, 1, 16, 2000, 3),
                                         (512, 1, 4, 900, 64), (16, 32, 8, 700, 2), (8, 8, 2, 5000, 11)])
def test_static_entry_point_matches_train_batch(R, C, N, B1, B2):
    import torch
    from wavenet_amd import Context
    from wavenet_amd.capi import WavenetError
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    a = F.perturbed(F.daubechies(N), 1e-2, 3)
    ctx.set_filter(a)
    rng = np.random.default_rng(R + C + N)
    X = np.where(rng.random((B1, R, C)) < 0.7, 0.0, rng.random((B1, R, C)))
    X[:, 0, 0] += 0.5
    dX = to_dev(X)
    out = torch.full((N + 2,), float("nan"), dtype=torch.float64, device="cuda")
    for B in (B1, B2, B1):
        ctx.train_batch(dX, B, R, C)
        want = ctx.sums().copy()
        out.fill_(float("nan"))
        ctx.train_batch_static(dX, B, R, C, out)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert want[N + 1] == 0 and np.isfinite(want).all()
        assert np.array_equal(got, want), (B, got, want)
    # the context's own sums are untouched by the static call
    assert np.array_equal(ctx.sums(), want)
    ctx.close()


def test_static_entry_point_refuses_streamed_shapes():
    import torch
    from wavenet_amd import Context
    from wavenet_amd.capi import WN_ERR_UNSUPPORTED, WavenetError
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_filter(F.DB4)
    out = dev_empty(6)
    dX = dev_empty(128 * 64)
    with pytest.raises(WavenetError) as err:
        ctx.train_batch_static(dX, 1, 128, 64, out)
    assert err.value.status == WN_ERR_UNSUPPORTED and "8192 samples exceed the 4096" in str(err.value)
    ctx.close()


def test_static_calls_of_two_contexts_on_two_streams_do_not_overlap():
    """Two contexts of one GPU, each on a stream of its own, share the GPU's one array.  Calls issued one after
    the other through the entry point are ordered by the library (the second stream waits for the event the
    first call recorded), so alternating calls without any synchronisation in between still give each context
    the sums wn_train_batch gives it."""
    import torch
    from wavenet_amd import Context
    R, C, N = 64, 64, 4
    rng = np.random.default_rng(8)
    shapes = [(3000, 1), (2500, 2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs, Xs, outs, wants = [], [], [], []
    for (B, seed), st in zip(shapes, streams):
        ctx = Context(0, stream=st.cuda_stream)
        ctx.set_filter(F.perturbed(F.daubechies(N), 1e-2, seed))
        X = np.where(rng.random((B, R, C)) < 0.7, 0.0, rng.random((B, R, C)))
        dX = to_dev(X)
        torch.cuda.synchronize()
        ctx.train_batch(dX, B, R, C)
        wants.append(ctx.sums().copy())
        ctxs.append(ctx)
        Xs.append(dX)
        outs.append([torch.full((N + 2,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(8)])
    torch.cuda.synchronize()
    for rep in range(8):
        for k in (0, 1):
            ctxs[k].train_batch_static(Xs[k], shapes[k][0], R, C, outs[k][rep])
    torch.cuda.synchronize()
    for k in (0, 1):
        for rep in range(8):
            assert np.array_equal(outs[k][rep].cpu().numpy(), wants[k]), (k, rep)
    for ctx in ctxs:
        ctx.close()
