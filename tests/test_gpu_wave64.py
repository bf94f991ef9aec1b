"""The one-wave-per-example training kernel of 64x64 examples (wavenet_amd/csrc/wn_wave64.h) against
the oracle and against the workgroup-per-example kernel, through the C ABI.

Covers what is particular to it: batch sizes that do not fill the 4-example workgroups or exceed
one pass of the grid, exact ties (ranked by ascending index), near-ties that differ only in the 12
mantissa bits the sort key gives to the index, zeros, and magnitudes outside the range the fast
sign / ranking paths accept.
"""
import os

import numpy as np
import pytest

import filters as F
from gpu_util import dev_empty, relerr, to_dev

pytestmark = pytest.mark.gpu

R = C = 64


def _context(wave64):
    import torch
    from wavenet_amd import Context
    old = os.environ.get("WAVENET_HIP_WAVE64")
    os.environ["WAVENET_HIP_WAVE64"] = "1" if wave64 else "0"
    try:
        c = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    finally:
        if old is None:
            del os.environ["WAVENET_HIP_WAVE64"]
        else:
            os.environ["WAVENET_HIP_WAVE64"] = old
    return c


@pytest.fixture(scope="module")
def ctx():
    import torch
    c = _context(True)
    yield c
    torch.cuda.synchronize()
    c.close()


@pytest.fixture(scope="module")
def ctx_block():
    import torch
    c = _context(False)
    yield c
    torch.cuda.synchronize()
    c.close()


def filt_for(N, seed=0):
    if N in (2, 4, 8, 16):
        return F.perturbed(F.daubechies(N), 1e-2, seed)
    return F.random_filter(N, seed)


def run(c, a, X):
    B, N = X.shape[0], len(a)
    c.set_filter(a)
    dG, dS = dev_empty(B * N), dev_empty(B)
    c.train_batch(to_dev(X), B, R, C, dG, dS)
    sums = c.sums()
    return dG.cpu().numpy().reshape(B, N), dS.cpu().numpy(), sums


def sparse_inputs(B, seed):
    rng = np.random.default_rng(seed)
    X = np.where(rng.random((B, R, C)) < 0.9, 0.0, rng.random((B, R, C)))
    X[:, 0, 0] += 0.25
    return X


@pytest.mark.parametrize("N", [2, 4, 6, 8, 12, 16, 22, 32])
@pytest.mark.parametrize("B", [1, 3, 4, 37])
def test_against_oracle(ctx, oracle, N, B):
    a = filt_for(N, 1)
    X = sparse_inputs(B, seed=10 * N + B) if B != 3 else np.random.default_rng(N).standard_normal((B, R, C))
    G, S, sums = run(ctx, a, X)
    og, os_, bad = oracle.band_train_batch(a, X)
    assert bad == 0 and sums[N + 1] == 0
    assert relerr(sums[:N], og) < 1e-10
    assert sums[N] == pytest.approx(os_, rel=1e-12)
    for e in range(min(B, 5)):
        ge, se, _ = oracle.band_train_batch(a, X[e:e + 1])
        assert relerr(G[e], ge) < 1e-10
        assert S[e] == pytest.approx(se, rel=1e-12)


@pytest.mark.parametrize("N", [2, 4, 8, 16, 32])
def test_many_passes_against_block_kernel(ctx, ctx_block, N):
    """More examples than one pass of the grid (1024 waves) holds; every example compared."""
    B = 2500
    a = filt_for(N, 2)
    X = sparse_inputs(B, seed=N)
    G, S, sums = run(ctx, a, X)
    Gb, Sb, sumsb = run(ctx_block, a, X)
    assert sums[N + 1] == 0 and sumsb[N + 1] == 0
    assert np.abs(S - Sb).max() < 1e-13
    assert relerr(G, Gb) < 1e-10
    assert relerr(sums, sumsb) < 1e-11


def test_exact_ties_and_zeros(ctx, ctx_block, oracle):
    """Haar on needles and on constant blocks: most coefficients are zero and the rest come in groups of
    equal magnitude.  Magnitudes that are equal in exact arithmetic may differ in the last bit between
    two evaluation orders (the oracle multiplies and adds, the kernels use fused multiply-adds), and
    the Gini gradient jumps when two of them swap ranks -- so the gradient is compared with the
    workgroup-per-example kernel, which evaluates the same expressions, and only the sparsity term,
    which is continuous, with the oracle."""
    a = F.HAAR
    B = 6
    X = np.zeros((B, R, C))
    rng = np.random.default_rng(0)
    for e in range(B):
        for _ in range(1 + e):
            X[e, rng.integers(R), rng.integers(C)] = 1.0
    X[B - 1, :32, :16] = 2.0
    G, S, sums = run(ctx, a, X)
    Gb, Sb, sumsb = run(ctx_block, a, X)
    assert sums[3] == 0
    assert relerr(G, Gb) < 1e-12
    assert np.abs(S - Sb).max() < 1e-14
    for e in range(B):
        _, se, _ = oracle.band_train_batch(a, X[e:e + 1])
        assert S[e] == pytest.approx(se, rel=1e-10)


@pytest.mark.parametrize("taps", [(0.5, 0.5), (0.5, 0.25, 0.5, -0.25), (0.75, 0.5, 0.25, -0.5, 1.0, 0.125, -0.25, 0.5)])
@pytest.mark.parametrize("wave64", [True, False])
def test_exact_ties_against_oracle(ctx, ctx_block, oracle, taps, wave64):
    """Tie handling checked against the oracle itself: with dyadic taps and small-integer inputs every
    product and sum on the path is exact, fused or not, so equal magnitudes are equal in both
    evaluations, no rank can flip, and the gradient must agree with the oracle's (ties in ascending
    index order in both)."""
    a = np.array(taps)
    B = 5
    rng = np.random.default_rng(len(taps))
    X = np.zeros((B, R, C))
    for e in range(B):
        for _ in range(3 + 4 * e):
            X[e, rng.integers(R), rng.integers(C)] = float(rng.integers(1, 5))
    X[B - 1, 8:40, 16:32] = 2.0
    c = ctx if wave64 else ctx_block
    G, S, sums = run(c, a, X)
    assert sums[len(a) + 1] == 0
    for e in range(B):
        ge, se, _ = oracle.band_train_batch(a, X[e:e + 1])
        assert relerr(G[e], ge) < 1e-10
        assert S[e] == pytest.approx(se, rel=1e-12)


def test_near_ties_take_exact_path(ctx, ctx_block, oracle):
    """Coefficients that differ only in the last mantissa bits and stand in the opposite order of their
    indices: the sort key's index bits would misorder them, the neighbour check must catch it and rank
    by exact magnitude, as the workgroup-per-example kernel does by construction (same expressions,
    full 64-bit keys).  The gradient is compared with that kernel, the continuous sparsity term also
    with the oracle."""
    a = F.HAAR
    rng = np.random.default_rng(5)
    B = 4
    X = np.zeros((B, R, C))
    for e in range(B):
        blocks = 1.0 + np.arange(32 * 32)[::-1].reshape(32, 32) * 2.0 ** -50   # descending with the index
        X[e] = np.kron(blocks, np.ones((2, 2))) * (1 + e)
        X[e, ::2, ::2] += rng.integers(0, 3, size=(32, 32)) * 2.0 ** -49
    G, S, sums = run(ctx, a, X)
    Gb, Sb, _ = run(ctx_block, a, X)
    assert sums[3] == 0
    assert relerr(G, Gb) < 1e-12
    assert np.abs(S - Sb).max() < 1e-14
    for e in range(B):
        _, se, _ = oracle.band_train_batch(a, X[e:e + 1])
        assert S[e] == pytest.approx(se, rel=1e-10)


@pytest.mark.parametrize("scale", [1e-200, 1e200, 1e-310])
def test_magnitudes_outside_fast_range(ctx, ctx_block, scale):
    """Tiny / huge / denormal inputs go down the exact ranking path with the reference's sign
    expression; the workgroup-per-example kernel is the yardstick."""
    a = filt_for(4, 3)
    X = sparse_inputs(5, seed=3) * scale
    G, S, sums = run(ctx, a, X)
    Gb, Sb, sumsb = run(ctx_block, a, X)
    assert np.array_equal(np.isfinite(S), np.isfinite(Sb))
    fin = np.isfinite(Sb)
    assert np.abs(S[fin] - Sb[fin]).max() < 1e-12 if fin.any() else True
    assert np.array_equal(np.isfinite(G), np.isfinite(Gb))
    m = np.isfinite(Gb)
    if m.any():
        assert relerr(G[m], Gb[m]) < 1e-10
    assert sums[-1] == sumsb[-1]


def test_all_zero_example_counts_nonfinite(ctx):
    a = F.DB4
    X = np.zeros((5, R, C))
    X[1] = np.random.default_rng(0).random((R, C))
    X[4] = X[1]
    G, S, sums = run(ctx, a, X)
    assert np.isnan(S[0]) and np.isnan(S[2]) and np.isfinite(S[1])
    assert S[1] == S[4] and np.array_equal(G[1], G[4])
    assert np.isnan(sums[-2]) and sums[-1] > 0


def test_no_per_example_outputs(ctx, oracle):
    a = filt_for(4, 7)
    X = sparse_inputs(9, seed=2)
    ctx.set_filter(a)
    ctx.train_batch(to_dev(X), 9, R, C)
    sums = ctx.sums()
    og, os_, _ = oracle.band_train_batch(a, X)
    assert relerr(sums[:4], og) < 1e-10
    assert sums[4] == pytest.approx(os_, rel=1e-12)
