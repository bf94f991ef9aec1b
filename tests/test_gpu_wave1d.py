"""The one-wave-per-example training kernel of 1D signals of 1024 samples
(wavenet_amd/csrc/wn_wave1d.h) against the oracle and against the workgroup-per-example kernel,
through the C ABI: every compiled tap count and the padded ones in between, batches smaller and
larger than one pass of the grid, exact ties and zeros.
"""
import os

import numpy as np
import pytest

import filters as F
from gpu_util import dev_empty, relerr, to_dev

pytestmark = pytest.mark.gpu

R, C = 1024, 1


def _context(wave):
    import torch
    from wavenet_amd import Context
    old = os.environ.get("WAVENET_HIP_WAVE64")
    os.environ["WAVENET_HIP_WAVE64"] = "1" if wave else "0"
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


@pytest.mark.parametrize("N", [2, 4, 6, 8, 10, 16, 22, 30, 32])
@pytest.mark.parametrize("B", [1, 3, 37])
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
    """More examples than one pass of the grid holds; every example compared."""
    B = 9000
    a = filt_for(N, 2)
    X = sparse_inputs(B, seed=N)
    G, S, sums = run(ctx, a, X)
    Gb, Sb, sumsb = run(ctx_block, a, X)
    assert sums[N + 1] == 0 and sumsb[N + 1] == 0
    assert np.abs(S - Sb).max() < 1e-13
    assert relerr(G, Gb) < 1e-11
    assert relerr(sums, sumsb) < 1e-11


def test_exact_ties_and_zeros(ctx, ctx_block, oracle):
    """Haar on needles and constant runs: groups of equal magnitudes (ranked by ascending index) and
    zeros.  Magnitudes equal in exact arithmetic may differ in the last bit between the oracle's
    multiply-and-add and the kernels' fused multiply-add, and the Gini gradient jumps when two swap
    ranks, so the gradient is compared with the workgroup-per-example kernel and only the sparsity
    term, which is continuous, with the oracle."""
    a = F.HAAR
    B = 6
    X = np.zeros((B, R, C))
    rng = np.random.default_rng(0)
    for e in range(B):
        for _ in range(1 + e):
            X[e, rng.integers(R), 0] = 1.0
    X[B - 1, :256, 0] = 2.0
    G, S, sums = run(ctx, a, X)
    Gb, Sb, sumsb = run(ctx_block, a, X)
    assert sums[3] == 0
    assert relerr(G, Gb) < 1e-12
    assert np.abs(S - Sb).max() < 1e-14
    for e in range(B):
        _, se, _ = oracle.band_train_batch(a, X[e:e + 1])
        assert S[e] == pytest.approx(se, rel=1e-10)


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


@pytest.mark.parametrize("n", [256, 512])
@pytest.mark.parametrize("N", [2, 4, 6, 8])
def test_shorter_signals_against_oracle_and_block_kernel(ctx, ctx_block, oracle, n, N):
    """Signals of 256 and 512 samples with up to 8 taps take the one-wave kernel built from the line
    pieces (k_train_line): against the oracle, and example by example against the workgroup kernel on
    sparse inputs with exact ties."""
    B = 70
    a = filt_for(N, 3)
    rng = np.random.default_rng(n + N)
    X = np.where(rng.random((B, n, 1)) < 0.8, 0.0, rng.random((B, n, 1)))
    X[:, 0, 0] += 0.25
    X[3] = np.round(X[3] * 4) / 4          # few distinct values: ties
    def go(c):
        c.set_filter(a)
        dG, dS = dev_empty(B * N), dev_empty(B)
        c.train_batch(to_dev(X), B, n, 1, dG, dS)
        return dG.cpu().numpy().reshape(B, N), dS.cpu().numpy(), c.sums()
    G, S, sums = go(ctx)
    Gb, Sb, sumsb = go(ctx_block)
    assert sums[N + 1] == 0
    assert np.abs(S - Sb).max() < 1e-13 and relerr(G, Gb) < 1e-11 and relerr(sums, sumsb) < 1e-11
    og, os_, bad = oracle.band_train_batch(a, X)
    assert bad == 0
    assert sums[N] == pytest.approx(os_, rel=1e-12)
    for e in (0, 1, 2):
        ge, se, _ = oracle.band_train_batch(a, X[e:e + 1])
        assert relerr(G[e], ge) < 1e-10 and S[e] == pytest.approx(se, rel=1e-12)
