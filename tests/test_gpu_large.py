"""The streamed path (R*C > 4096: line and level kernels; Gini ranks by histogram + per-bucket LDS sort, or by
device-wide radix sort with WAVENET_HIP_RANK=radix and whenever a bucket would overflow) against the CPU oracle,
including the full sizes of the 512x512 and 2^20-sample configurations."""
import numpy as np
import pytest

import filters as F
from gpu_util import dev_empty, from_dev, relerr, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    from wavenet_amd import Context
    c = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    torch.cuda.synchronize()
    c.close()


def _ctx_with_env(name, value):
    import os
    import torch
    from wavenet_amd import Context
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return Context(0, stream=torch.cuda.current_stream().cuda_stream)
    finally:
        if old is None:
            del os.environ[name]
        else:
            os.environ[name] = old


@pytest.fixture(scope="module", params=["buckets", "radix"])
def ctx_rank(request):
    """A context per ranking method of the streamed path."""
    import torch
    c = _ctx_with_env("WAVENET_HIP_RANK", request.param)
    yield c
    torch.cuda.synchronize()
    c.close()


def filt_for(N, seed=0):
    if N in (2, 4, 8, 16):
        return F.perturbed(F.daubechies(N), 1e-2, seed)
    return F.random_filter(N, seed)


@pytest.mark.parametrize("R,C,N", [(128, 64, 4), (64, 128, 2), (8192, 1, 16), (1, 8192, 6), (256, 128, 32), (2, 4096, 8),
                                   (512, 512, 4), (512, 16, 8), (512, 32, 32),
                                   (1 << 17, 1, 32), (1 << 16, 1, 16), (1 << 16, 2, 16),      # long lines: several levels per launch
                                   (1 << 16, 1, 28), (1 << 16, 1, 30), (1 << 15, 1, 18)])     # 32 compiled taps, 2 / 1 / 7 of padding in front
def test_forward_inverse_backprop(ctx, oracle, R, C, N):
    import torch
    B = 2
    a = filt_for(N, 1)
    rng = np.random.default_rng(R + C + N)
    X = rng.random((B, R, C))
    D = rng.standard_normal((B, R, C))
    ctx.set_filter(a)
    n = R * C
    dX = to_dev(X)
    dY, dPr, dY1, dPc, dXr = (dev_empty(B * n) for _ in range(5))
    ctx.forward(dX, B, R, C, dY, dPr, dY1, dPc)
    ctx.inverse(dY, B, R, C, dXr)
    dG, dDin = dev_empty(B * N), dev_empty(B * n)
    ctx.backprop(to_dev(D), dX, dPr, dY1, dPc, B, R, C, dG, dDin)
    torch.cuda.synchronize()
    Y, Pr, Y1, Pc, Xr = (from_dev(t, B, R, C) for t in (dY, dPr, dY1, dPc, dXr))
    G = dG.cpu().numpy().reshape(B, N)
    for e in range(B):
        oPr, oY1, oPc, oY = oracle.band_forward_2d(a, X[e])
        assert relerr(Y[e], oY) < 1e-11
        assert relerr(Y1[e], oY1) < 1e-11
        assert np.abs(Pr[e] - oPr).max() < 1e-11 * max(1.0, np.abs(oPr).max())
        assert np.abs(Pc[e] - oPc).max() < 1e-11 * max(1.0, np.abs(oPc).max())
        assert relerr(Xr[e], oracle.band_inverse_2d(a, oY)) < 1e-10
        og = oracle.band_backprop_2d(a, D[e], X[e], oPr, oY1, oPc)
        assert relerr(G[e], og) < 1e-10


@pytest.mark.parametrize("n", [8192, 65536, 1 << 20])
def test_sparse_term_and_deriv(ctx, oracle, n):
    import torch
    B = 3
    rng = np.random.default_rng(n)
    Y = rng.standard_normal((B, n))
    Y[1, ::3] = 0.0
    Y[2, :] = np.where(rng.random(n) < 0.5, 0.5, -0.5) * (1 + (np.arange(n) % 5))  # heavy ties
    dY = torch.from_numpy(Y.reshape(-1)).cuda()
    dD, dS = dev_empty(B * n), dev_empty(B)
    ctx.sparse_term_deriv(dY, B, n, dD, dS)
    torch.cuda.synchronize()
    D = dD.cpu().numpy().reshape(B, n)
    S = dS.cpu().numpy()
    for e in range(B):
        assert S[e] == pytest.approx(oracle.sparse_term(Y[e]), rel=1e-11)
        assert relerr(D[e], oracle.sparse_term_deriv(Y[e])) < 1e-9
    onehot = np.zeros((1, n))
    onehot[0, n // 3] = 2.0
    ctx.sparse_term(torch.from_numpy(onehot.reshape(-1)).cuda(), 1, n, dS)
    assert dS.cpu().numpy()[0] == pytest.approx(1.0 / n, rel=1e-12)


@pytest.mark.parametrize("R,C,N,B", [(128, 64, 4, 5), (8192, 1, 16, 3), (512, 512, 4, 2), (512, 512, 16, 2),
                                     (1 << 20, 1, 32, 2), (512, 16, 6, 3), (512, 64, 32, 2), (512, 32, 2, 70),
                                     (512, 128, 8, 2), (16, 512, 4, 3), (64, 512, 16, 2), (128, 512, 32, 2),
                                     (1024, 512, 2, 2), (32, 512, 6, 40), (128, 128, 4, 3), (256, 256, 8, 2),
                                     (1024, 128, 2, 2), (128, 1024, 16, 2), (256, 64, 32, 2), (1024, 1024, 4, 1),
                                     (2048, 256, 4, 2), (8, 1024, 6, 5), (1 << 17, 1, 28, 2), (1 << 17, 1, 30, 2)])
def test_train_batch(ctx, oracle, R, C, N, B):
    """Batch sums and per-example values of the full step; 512x512 and 2^20 x 1 are the full sizes of the
    data-parallel configurations."""
    import torch
    from wavenet_amd import generators as G
    a = filt_for(N, 2)
    X = G.calorimeter(B, R, C, seed=3) if C > 1 else G.bumps(B, R, seed=3)
    ctx.set_filter(a)
    dG, dS = dev_empty(B * N), dev_empty(B)
    ctx.train_batch(to_dev(X), B, R, C, dG, dS)
    sums = ctx.sums()
    Gd = dG.cpu().numpy().reshape(B, N)
    Sd = dS.cpu().numpy()
    og, os_, bad = oracle.band_train_batch(a, X)
    assert bad == 0 and sums[N + 1] == 0
    assert relerr(sums[:N], og) < 1e-9
    assert sums[N] == pytest.approx(os_, rel=1e-11)
    ge, se, _ = oracle.band_train_batch(a, X[:1])
    assert relerr(Gd[0], ge) < 1e-9
    assert Sd[0] == pytest.approx(se, rel=1e-11)
    # without per-example outputs the sums are the same
    ctx.train_batch(to_dev(X), B, R, C)
    assert np.array_equal(ctx.sums(), sums)


@pytest.mark.parametrize("R,C,N", [(16384, 1, 2), (256, 128, 2)])
def test_train_batch_when_the_bucket_ranking_gives_up(oracle, capfd, monkeypatch, R, C, N):
    """An example whose coefficients crowd into one bucket (a constant signal with alternating sign under the
    two-tap averaging filter: thousands of coefficients of one magnitude) makes the bucket ranking give up on its chunk.  The
    training step has queued the whole chunk by the time the host sees that, and computes it again with the radix
    ranking; the sums must be those of the oracle, with no trace of the abandoned attempt -- also when the chunk
    is not the first one of the batch."""
    import torch
    from wavenet_amd import generators as G
    monkeypatch.setenv("WAVENET_HIP_VERBOSE", "1")
    B = 5
    # taps that are powers of two: every product and sum of the crowded example is exact, so its thousands of equal
    # magnitudes are equal bit for bit here and in the oracle, and both order them by index (with rounding in play the
    # two would break the near-ties differently, and the Gini gradient is discontinuous there)
    a = np.array([0.5, 0.5])
    X = G.calorimeter(B, R, C, seed=5) if C > 1 else G.bumps(B, R, seed=5).reshape(B, R, C)
    sign = np.where((np.arange(R)[:, None] + np.arange(C)[None, :]) % 2 == 0, 1.0, -1.0)
    X[3] = 0.75 * sign
    c = _ctx_with_env("WAVENET_HIP_CHUNK", "2")      # chunks of two examples: the crowded one sits in the second chunk
    try:
        c.set_filter(a)
        dG, dS = dev_empty(B * N), dev_empty(B)
        c.train_batch(to_dev(X), B, R, C, dG, dS)
        torch.cuda.synchronize()
        sums = c.sums()
        err = capfd.readouterr().err
        assert "bucket ranking gave up" in err
        og, os_, bad = oracle.band_train_batch(a, X)
        assert bad == 0 and sums[N + 1] == 0
        assert relerr(sums[:N], og) < 1e-9
        assert sums[N] == pytest.approx(os_, rel=1e-11)
        Sd = dS.cpu().numpy()
        for e in range(B):
            assert Sd[e] == pytest.approx(oracle.band_train_batch(a, X[e:e + 1])[1], rel=1e-11)
        # the same batch again gives the same sums (nothing of the abandoned attempt is left behind)
        c.train_batch(to_dev(X), B, R, C)
        assert np.array_equal(c.sums(), sums)
    finally:
        torch.cuda.synchronize()
        c.close()


def test_full_size_properties_512(ctx):
    """Size-independent properties at 512x512: perfect reconstruction for an orthonormal filter, linearity of
    the forward transform, and energy conservation (Parseval) of the coefficients."""
    import torch
    from wavenet_amd import generators as G
    R = C = 512
    B = 4
    X = G.calorimeter(B, R, C, seed=9)
    ctx.set_filter(F.daubechies(8))
    n = R * C
    dX = to_dev(X)
    dY, dXr, dY2 = dev_empty(B * n), dev_empty(B * n), dev_empty(B * n)
    ctx.forward(dX, B, R, C, dY)
    ctx.inverse(dY, B, R, C, dXr)
    ctx.forward(2.5 * dX, B, R, C, dY2)
    torch.cuda.synchronize()
    assert torch.allclose(dXr, dX, rtol=0, atol=1e-9)
    assert torch.allclose(dY2, 2.5 * dY, rtol=1e-12, atol=1e-12)
    ex = (dX.view(B, n) ** 2).sum(dim=1)
    ey = (dY.view(B, n) ** 2).sum(dim=1)
    assert torch.allclose(ex, ey, rtol=1e-9)


@pytest.mark.parametrize("n,B", [(8192, 5), (65536, 3), (1 << 18, 2)])
def test_sparse_term_32bit_keys_against_64bit_sort(ctx_rank, oracle, n, B):
    """The streamed Gini ranks with one sort of 32-bit keys (example number, leading magnitude bits) and
    re-ranks groups of equal keys by exact magnitude and index.  Inputs: smooth random values (small
    groups everywhere), zeros, and pairs that differ only in their last mantissa bits and stand in the
    opposite order of their indices.  Yardsticks: the 64-bit sort of the same library and the oracle."""
    import torch
    rng = np.random.default_rng(n + B)
    Y = rng.standard_normal((B, n)) * np.exp(rng.standard_normal((B, n)))
    Y[0, ::7] = 0.0
    base = 1.0 + rng.random(n // 8)
    Y[1, 0:n // 8] = base * (1 + 2.0 ** -50)      # the larger one first ...
    Y[1, n // 2:n // 2 + n // 8] = -base          # ... its near-twin later, with the other sign
    Y[B - 1, 5::11] = Y[B - 1, 3::11][:len(Y[B - 1, 5::11])]   # exact ties
    dY = torch.from_numpy(Y.reshape(-1)).cuda()
    dD, dS = dev_empty(B * n), dev_empty(B)
    ctx_rank.sparse_term_deriv(dY, B, n, dD, dS)
    torch.cuda.synchronize()
    D, S = dD.cpu().numpy().reshape(B, n), dS.cpu().numpy()
    exact = _ctx_with_env("WAVENET_HIP_EXACT_SORT", "1")
    try:
        dD2, dS2 = dev_empty(B * n), dev_empty(B)
        exact.sparse_term_deriv(dY, B, n, dD2, dS2)
        torch.cuda.synchronize()
        D2, S2 = dD2.cpu().numpy().reshape(B, n), dS2.cpu().numpy()
    finally:
        exact.close()
    assert np.abs(S - S2).max() < 1e-13
    for e in range(B):
        assert relerr(D[e], D2[e]) < 1e-11
        assert S[e] == pytest.approx(oracle.sparse_term(Y[e]), rel=1e-11)
        assert relerr(D[e], oracle.sparse_term_deriv(Y[e])) < 1e-10


@pytest.mark.parametrize("n", [8192, 1 << 18])
@pytest.mark.parametrize("in_place", [False, True])
def test_sparse_term_zeros_left_out_of_the_sort(ctx_rank, oracle, n, in_place):
    """Exact zeros are compacted away before the sort and rank first: an all-zero example (NaN cost, as in
    the reference), a mostly-zero one with its non-zeros at workgroup boundaries, a dense one, a single
    non-zero coefficient, and signed zeros; errors written in place and to a separate array."""
    import torch
    B = 5
    rng = np.random.default_rng(n)
    Y = np.zeros((B, n))
    Y[1] = np.where(rng.random(n) < 0.05, rng.standard_normal(n), 0.0)
    Y[1, [0, 2047, 2048, 4095, 4096, n - 1]] = [3.0, -2.0, 1.0, 1.0, -1.0, 0.5]
    Y[2] = rng.standard_normal(n)
    Y[3, n // 2 + 1] = -4.0
    Y[4] = np.where(rng.random(n) < 0.5, rng.standard_normal(n), -0.0)
    dY = torch.from_numpy(Y.reshape(-1).copy()).cuda()
    dD = dY if in_place else dev_empty(B * n).fill_(7.0)
    dS = dev_empty(B)
    ctx_rank.sparse_term_deriv(dY, B, n, dD, dS)
    torch.cuda.synchronize()
    D = dD.cpu().numpy().reshape(B, n)
    S = dS.cpu().numpy()
    assert np.isnan(S[0])
    for e in range(1, B):
        assert S[e] == pytest.approx(oracle.sparse_term(Y[e]), rel=1e-11)
        want = oracle.sparse_term_deriv(Y[e])
        assert relerr(D[e], want) < 1e-9
        assert np.all(D[e][Y[e] == 0] == 0)
    assert S[3] == pytest.approx(1.0 / n, rel=1e-12)


@pytest.mark.parametrize("n", [8192, 1 << 18])
def test_sparse_term_magnitudes_outside_the_key_range(ctx_rank, oracle, n):
    """The sort key keeps 7 exponent bits (2^-95 ... 2^33) and the leading mantissa bits; smaller and larger
    magnitudes share the end keys and are ordered by the exact re-ranking of equal keys, as are the
    several coefficients per key of a dense example."""
    import torch
    B = 3
    rng = np.random.default_rng(n + 1)
    Y = rng.standard_normal((B, n))
    Y[0, :300] = rng.standard_normal(300) * 1e-40        # below the range, few enough for the short-key path
    Y[0, 300:500] = rng.standard_normal(200) * 1e20      # above it
    Y[1] *= np.exp(8 * rng.standard_normal(n))           # wide but inside
    Y[2] = 1.0 + rng.random(n) * 2.0 ** -8               # one binade, many per key
    dY = torch.from_numpy(Y.reshape(-1)).cuda()
    dD, dS = dev_empty(B * n), dev_empty(B)
    ctx_rank.sparse_term_deriv(dY, B, n, dD, dS)
    torch.cuda.synchronize()
    D, S = dD.cpu().numpy().reshape(B, n), dS.cpu().numpy()
    for e in range(B):
        assert S[e] == pytest.approx(oracle.sparse_term(Y[e]), rel=1e-11)
        assert relerr(D[e], oracle.sparse_term_deriv(Y[e])) < 1e-10


@pytest.mark.parametrize("R,C,N", [(512, 512, 4), (64, 512, 8), (512, 32, 16), (256, 256, 4), (128, 1024, 8), (1024, 128, 32)])
def test_forward_coefficients_only(ctx, R, C, N):
    """wn_forward asked for the coefficients alone takes the one-wave-per-line kernels for lines of 128 ... 1024
    samples; asked for the activations as well it takes the level kernels.  Same coefficients."""
    import torch
    from wavenet_amd import generators as G
    B = 3
    X = G.calorimeter(B, R, C, seed=5)
    ctx.set_filter(filt_for(N, 3))
    n = R * C
    dX = to_dev(X)
    dY, dY2, dPr, dY1, dPc = (dev_empty(B * n) for _ in range(5))
    ctx.forward(dX, B, R, C, dY)
    ctx.forward(dX, B, R, C, dY2, dPr, dY1, dPc)
    torch.cuda.synchronize()
    assert torch.allclose(dY, dY2, rtol=1e-13, atol=1e-15)


def test_slices_of_the_chunk_do_not_change_the_examples(ctx):
    """260 images of 512x512 are more than one sort covers (2^26 coefficients per slice): every example's
    gradient and cost must come out the same as when the batch is handed over in two halves of one slice
    each, and the batch sums must agree."""
    import torch
    from wavenet_amd import generators as G
    R = C = 512
    N, B = 4, 260
    ctx.set_filter(filt_for(N, 5))
    n = R * C
    dX = torch.empty(B * n, dtype=torch.float64, device="cuda")
    ctx.generate_calorimeter(dX, B, R, C, seed=77, first=0)
    dG, dS = dev_empty(B * N), dev_empty(B)
    ctx.train_batch(dX, B, R, C, dG, dS)
    sums = ctx.sums().copy()
    G1, S1 = dG.cpu().numpy().copy(), dS.cpu().numpy().copy()
    half = B // 2
    dG2, dS2 = dev_empty(B * N), dev_empty(B)
    ctx.train_batch(dX, half, R, C, dG2, dS2)
    s_a = ctx.sums().copy()
    ctx.train_batch(dX[half * n:], B - half, R, C, dG2[half * N:], dS2[half:])
    s_b = ctx.sums().copy()
    assert np.array_equal(G1, dG2.cpu().numpy())
    assert np.array_equal(S1, dS2.cpu().numpy())
    assert sums[-1] == 0
    assert relerr(sums[:-1], (s_a + s_b)[:-1]) < 1e-12


@pytest.mark.parametrize("R,C,N", [(512, 512, 4), (256, 128, 8), (512, 256, 16), (128, 1024, 2), (1024, 256, 6)])
def test_line_kernels_against_level_kernels(ctx, R, C, N):
    """The same batch through the one-wave-per-line kernels (default) and, with the one-wave kernels switched
    off, level by level: per-example gradients and costs agree to rounding, the batch sums too."""
    import torch
    from wavenet_amd import generators as G
    B = 5
    X = G.calorimeter(B, R, C, seed=R + C + N)
    a = filt_for(N, 6)
    dX = to_dev(X)
    ctx.set_filter(a)
    dG, dS = dev_empty(B * N), dev_empty(B)
    ctx.train_batch(dX, B, R, C, dG, dS)
    sums = ctx.sums().copy()
    plain = _ctx_with_env("WAVENET_HIP_WAVE64", "0")
    try:
        plain.set_filter(a)
        dG2, dS2 = dev_empty(B * N), dev_empty(B)
        plain.train_batch(dX, B, R, C, dG2, dS2)
        sums2 = plain.sums().copy()
        torch.cuda.synchronize()
    finally:
        plain.close()
    assert relerr(dG.cpu().numpy(), dG2.cpu().numpy()) < 1e-11
    assert np.abs(dS.cpu().numpy() / dS2.cpu().numpy() - 1).max() < 1e-12
    assert relerr(sums[:N + 1], sums2[:N + 1]) < 1e-11 and sums[N + 1] == sums2[N + 1] == 0


def _capture_ranking(ctx, capfd, dY, B, n):
    """sparse_term_deriv with the library's remarks on; returns (D, S, gave_up)."""
    import os
    import torch
    dD, dS = dev_empty(B * n), dev_empty(B)
    capfd.readouterr()
    ctx.sparse_term_deriv(dY, B, n, dD, dS)
    torch.cuda.synchronize()
    err = capfd.readouterr().err
    return dD.cpu().numpy().reshape(B, n), dS.cpu().numpy(), "bucket ranking gave up" in err


@pytest.mark.parametrize("n,B", [(8192, 700), (16384, 3), (65536, 9), (1 << 18, 3), (1 << 20, 2), (1 << 21, 1)])
def test_bucket_ranking(oracle, capfd, monkeypatch, n, B):
    """Ranks by histogram and per-bucket LDS sort (wn_rank.h) against the oracle and against the radix-sort path:
    many examples (one workgroup per example in the histogram and partition kernels) and few (several workgroups
    per example, cursors in global memory); stragglers hundreds of octaves below the rest (the lowest bin collects
    what lies 32 octaves below the largest; in long examples its bucket sorts magnitudes and indices side by side); zeros;
    exact ties and near-ties in the opposite order of their indices; a constant-magnitude example that overflows
    every bucket and must take the radix path, alone in its chunk."""
    import torch
    monkeypatch.setenv("WAVENET_HIP_VERBOSE", "1")
    rng = np.random.default_rng(n + B)
    Y = rng.standard_normal((B, n)) * np.exp(rng.standard_normal((B, n)))
    Y[0, ::5] = 0.0
    if B > 1:
        Y[1] = rng.standard_normal(n) * 2.0 ** rng.integers(-20, 1, size=n)
        far = rng.choice(n, size=1500, replace=False)                                # stragglers hundreds of octaves below
        Y[1, far] = rng.standard_normal(1500) * 2.0 ** rng.integers(-440, -40, size=1500)
    last = B - 1
    m = min(n // 8, 8192)
    base = 1.0 + rng.random(m)
    Y[last, 0:m] = base * (1 + 2.0 ** -50)
    Y[last, n // 2:n // 2 + m] = -base
    Y[last, 5::11] = Y[last, 3::11][:len(Y[last, 5::11])]
    dY = torch.from_numpy(Y.reshape(-1)).cuda()
    buckets = _ctx_with_env("WAVENET_HIP_RANK", "buckets")
    radix = _ctx_with_env("WAVENET_HIP_RANK", "radix")
    try:
        D, S, gave_up = _capture_ranking(buckets, capfd, dY, B, n)
        assert not gave_up
        D2, S2, _ = _capture_ranking(radix, capfd, dY, B, n)
        assert np.abs(S - S2).max() < 1e-13
        for e in sorted({0, 1 % B, last, B // 2}):
            assert relerr(D[e], D2[e]) < 1e-11
            assert S[e] == pytest.approx(oracle.sparse_term(Y[e]), rel=1e-11)
            assert relerr(D[e], oracle.sparse_term_deriv(Y[e])) < 1e-9
        # every magnitude one of five values: each bin holds a fifth of the example
        ties = np.where(rng.random(n) < 0.5, 0.5, -0.5) * (1 + (np.arange(n) % 5))
        dT = torch.from_numpy(ties).cuda()
        Dt, St, gave_up = _capture_ranking(buckets, capfd, dT, 1, n)
        assert gave_up == (n >= 32768)     # 3277 equal magnitudes still fit a bucket, 6554 do not
        assert St[0] == pytest.approx(oracle.sparse_term(ties), rel=1e-11)
        assert relerr(Dt[0], oracle.sparse_term_deriv(ties)) < 1e-9
    finally:
        buckets.close()
        radix.close()


@pytest.mark.parametrize("R,C,N,B", [(128, 128, 4, 300), (256, 128, 8, 260), (128, 512, 2, 256), (512, 512, 4, 256),
                                     (1024, 128, 4, 256), (256, 256, 16, 256)])
def test_histograms_built_by_the_column_kernel(oracle, R, C, N, B):
    """With an example per CU or more and both passes on the line kernels, the column kernel leaves the ranking's
    histograms and largest magnitudes itself (kc_fwd_cols_hist) instead of kr_hist reading the coefficients
    back.  The histograms are the same numbers, so every sum and per-example value is bit for bit the one of the
    path with WAVENET_HIP_FUSED_HIST=0, and of the path that stores ranks at home positions (WAVENET_HIP_RANK_PAIRS=0)
    instead of handing (home index, rank) pairs to the column backward kernel; three examples are checked against the oracle.  Examples scaled by 1e-30 and
    1e30 sit far from the others' magnitudes; in the second batch the taps are scaled by 1e4, which puts the
    kernel's bound on the largest magnitude so far above the truth that it hands every example to kr_hist; a NaN
    in one example goes the same way, and an image of zeros has nothing to count."""
    import torch
    from wavenet_amd import generators as G
    a = filt_for(N, 2)
    X = G.calorimeter(B, R, C, seed=4)
    X[1] *= 1e-30
    X[2] *= 1e30
    fused = _ctx_with_env("WAVENET_HIP_FUSED_HIST", "1")
    plain = _ctx_with_env("WAVENET_HIP_FUSED_HIST", "0")
    home = _ctx_with_env("WAVENET_HIP_RANK_PAIRS", "0")     # ranks stored at home positions instead of (home, rank) pairs per region
    try:
        dX = to_dev(X)
        out = {}
        for name, c in (("fused", fused), ("plain", plain), ("home", home)):
            c.set_filter(a)
            dG, dS = dev_empty(B * N), dev_empty(B)
            c.train_batch(dX, B, R, C, dG, dS)
            out[name] = (c.sums().copy(), dG.cpu().numpy().reshape(B, N), dS.cpu().numpy())
        for k in range(3):
            assert np.array_equal(out["fused"][k], out["plain"][k]), k
        # the region-wise kernel adds a wave's window sums over other columns than the home-rank kernel: same ranks, same
        # errors, filter gradients equal to rounding; the sparsity terms come from the ranking alone and are the same bits
        assert np.array_equal(out["fused"][2], out["home"][2])
        assert relerr(out["fused"][1], out["home"][1]) < 1e-12
        assert relerr(out["fused"][0][:N + 1], out["home"][0][:N + 1]) < 1e-12
        assert out["fused"][0][N + 1] == 0
        for e in (0, 1, 2):
            ge, se, _ = oracle.band_train_batch(a, X[e:e + 1])
            assert relerr(out["fused"][1][e], ge) < 1e-9
            assert out["fused"][2][e] == pytest.approx(se, rel=1e-11)
        # the bound too far above the truth, and a non-finite example
        X[5, 3, 2] = np.nan
        X[7] = 0.0                 # and an image of zeros: no magnitude to count, no largest one
        dX = to_dev(X)
        for name, c in (("fused", fused), ("plain", plain)):
            c.set_filter(1e4 * a)
            dG, dS = dev_empty(B * N), dev_empty(B)
            c.train_batch(dX, B, R, C, dG, dS)
            out[name] = (c.sums().copy(), dG.cpu().numpy().reshape(B, N), dS.cpu().numpy())
        for k in range(3):
            assert np.array_equal(out["fused"][k], out["plain"][k], equal_nan=True), k
        assert out["fused"][0][N + 1] > 0
    finally:
        torch.cuda.synchronize()
        fused.close()
        plain.close()
        home.close()


def test_full_batch_of_512x512_against_its_parts(ctx):
    """The bench's full batch -- 2048 images of 512x512 (one GPU's share of the 16384-image configuration), generated
    on the device -- against the same images handed over in eight parts of 256: every per-example cost and gradient
    is the same bit for bit (an example's result does not depend on its neighbours in the batch, on the chunking of
    the workspace or on which kernel built its histogram), the batch sums agree to rounding, nothing is non-finite,
    and the sparsity costs lie in (0, 1)."""
    import torch
    R = C = 512
    N, B, part = 4, 2048, 256
    n = R * C
    ctx.set_filter(filt_for(N, 5))
    dX = torch.empty(B * n, dtype=torch.float64, device="cuda")
    ctx.generate_calorimeter(dX, B, R, C, seed=1000, first=0)
    dG, dS = dev_empty(B * N), dev_empty(B)
    ctx.train_batch(dX, B, R, C, dG, dS)
    sums = ctx.sums().copy()
    G1, S1 = dG.cpu().numpy().copy(), dS.cpu().numpy().copy()
    dG2, dS2 = dev_empty(B * N), dev_empty(B)
    acc = np.zeros_like(sums)
    for e0 in range(0, B, part):
        ctx.train_batch(dX[e0 * n:], part, R, C, dG2[e0 * N:], dS2[e0:])
        acc += ctx.sums()
    assert np.array_equal(G1, dG2.cpu().numpy())
    assert np.array_equal(S1, dS2.cpu().numpy())
    assert sums[-1] == 0 and acc[-1] == 0
    assert relerr(sums[:-1], acc[:-1]) < 1e-12
    assert np.all((S1 > 0) & (S1 < 1))
    assert sums[N] == pytest.approx(S1.sum(), rel=1e-12)
    del dX
    torch.cuda.empty_cache()


def test_batch_of_long_signals_against_its_parts(ctx):
    """Eight signals of 2^20 samples with 32 taps (one GPU's share of the 64-signal configuration) against the same
    signals one at a time: several workgroups share an example in the ranking there, one alone here."""
    import torch
    from wavenet_amd import generators as G
    n, N, B = 1 << 20, 32, 8
    ctx.set_filter(filt_for(N, 3))
    X = G.bumps(B, n, seed=1000).reshape(B, n, 1)
    dX = to_dev(X)
    dG, dS = dev_empty(B * N), dev_empty(B)
    ctx.train_batch(dX, B, n, 1, dG, dS)
    sums = ctx.sums().copy()
    G1, S1 = dG.cpu().numpy().copy(), dS.cpu().numpy().copy()
    dG2, dS2 = dev_empty(B * N), dev_empty(B)
    acc = np.zeros_like(sums)
    for e in range(B):
        ctx.train_batch(dX[e * n:], 1, n, 1, dG2[e * N:], dS2[e:])
        acc += ctx.sums()
    assert np.array_equal(S1, dS2.cpu().numpy())
    assert relerr(G1.reshape(B, N), dG2.cpu().numpy().reshape(B, N)) < 1e-12
    assert sums[-1] == 0
    assert relerr(sums[:-1], acc[:-1]) < 1e-12
    assert np.all((S1 > 0) & (S1 < 1))
