"""Known-answer tests that pin the CPU oracle (the reference ships no fixtures).

Closed-form answers, the identity of the two operator constructions and the
banded formula, perfect reconstruction, finite differences, and the fixed
point of the 2-tap training (reference examples/Example01.cxx:100-113).
"""
import numpy as np
import pytest

import filters as F


def banded_matrix(f, size):
    """M[r, (N/2 + 2r - j) mod 2^(size+1)] += f[j]."""
    N = len(f)
    nr, nc = 1 << size, 2 << size
    M = np.zeros((nr, nc))
    for r in range(nr):
        for j in range(N):
            M[r, (N // 2 + 2 * r - j) % nc] += f[j]
    return M


@pytest.mark.parametrize("N", [2, 4, 6, 16, 32])
@pytest.mark.parametrize("size", [0, 1, 2, 3, 4, 5])
def test_operator_constructions_agree(oracle, N, size):
    f = np.random.default_rng(N * 100 + size).standard_normal(N)
    by_rows = oracle.operator(f, size, "rows")
    by_idx = oracle.operator(f, size, "indices")
    ref = banded_matrix(f, size)
    assert np.array_equal(by_rows, by_idx) or np.allclose(by_rows, by_idx, rtol=0, atol=1e-15)
    assert np.allclose(by_rows, ref, rtol=0, atol=1e-15)
    assert np.allclose(oracle.operator(f, size), ref, rtol=0, atol=1e-15)


def test_highpass_taps(oracle):
    a = np.arange(1.0, 7.0)
    assert np.array_equal(oracle.highpass_filter(a), [6, -5, 4, -3, 2, -1])


@pytest.mark.parametrize("a", [F.HAAR, F.DB4, F.DB8, F.daubechies(16)])
def test_lowpass_highpass_orthogonal(oracle, a):
    for size in range(0, 5):
        L = oracle.lowpass_operator(a, size)
        H = oracle.highpass_operator(a, size)
        assert np.abs(L @ H.T).max() < 1e-12
        if (2 << size) >= len(a):  # no folding: rows orthonormal
            assert np.abs(L @ L.T - np.eye(1 << size)).max() < 1e-10
            assert np.abs(H @ H.T - np.eye(1 << size)).max() < 1e-10


@pytest.mark.parametrize("n", [2, 8, 64, 4096])
def test_sparse_term_closed_forms(oracle, n):
    onehot = np.zeros(n)
    onehot[n // 3] = -2.5
    assert oracle.sparse_term(onehot) == pytest.approx(1.0 / n, rel=1e-14)
    assert oracle.sparse_term(np.full(n, 0.7)) == pytest.approx(1.0, abs=1e-14)


def test_sparse_term_deriv_finite_difference(oracle):
    rng = np.random.default_rng(3)
    c = rng.standard_normal(64)
    g = oracle.sparse_term_deriv(c)
    h = 1e-7
    for j in range(0, 64, 5):
        cp, cm = c.copy(), c.copy()
        cp[j] += h
        cm[j] -= h
        fd = (oracle.sparse_term(cp) - oracle.sparse_term(cm)) / (2 * h)
        assert g[j] == pytest.approx(fd, abs=2e-8)


def test_sparse_term_deriv_ties_by_index(oracle):
    # equal magnitudes get ascending ranks in index order; zeros get gradient 0
    c = np.array([1.0, -1.0, 0.0, 1.0])
    g = oracle.sparse_term_deriv(c)
    n, s = 4, 3.0
    gg = n * s
    f = sum((2 * (i + 1) - n - 1) * v for i, v in enumerate([0.0, 1.0, 1.0, 1.0]))
    ranks = {0: 2, 1: 3, 3: 4}
    for j, rk in ranks.items():
        sgn = np.sign(c[j])
        expect = -((2 * rk - n - 1) * sgn * gg - f * n * sgn) / gg ** 2
        assert g[j] == pytest.approx(expect, rel=1e-14)
    assert g[2] == 0.0


@pytest.mark.parametrize("a", [F.HAAR, F.DB4, F.DB8])
def test_reg_term_zero_for_wavelets(oracle, a):
    assert abs(oracle.reg_term(a)) < 1e-13      # float accumulators: ~1e-7 relative on O(1e-7) deviations
    assert np.abs(oracle.reg_term_deriv(a)).max() < 1e-10  # DB8 literals carry ~1e-12 error
    assert abs(oracle.reg_term(a, wavelet=False)) < 1e-13


def test_reg_term_deriv_finite_difference(oracle):
    # RegTerm returns R/2 (CostFunctions.cxx:256) while RegTermDeriv carries the factor 2 of
    # d(...)^2 (CostFunctions.cxx:303): the reference's gradient is twice dRegTerm/da.
    a = F.random_filter(6, 1)
    g = oracle.reg_term_deriv(a, wavelet=False)
    h = 1e-4  # RegTerm is accumulated in float: use a wide step, loose tolerance
    for i in range(6):
        ap, am = a.copy(), a.copy()
        ap[i] += h
        am[i] -= h
        fd = (oracle.reg_term(ap, wavelet=False) - oracle.reg_term(am, wavelet=False)) / (2 * h)
        assert g[i] == pytest.approx(2 * fd, abs=5e-3)


def test_reg_term_float_rounding(oracle):
    # The value is accumulated in float: it differs from an all-double evaluation at ~1e-7 relative.
    a = F.random_filter(8, 5)
    N = len(a)

    def dbl():
        R = 0.0
        for m in range(-N // 2, N // 2 + 1):
            t = sum(a[k] * a[k + 2 * m] for k in range(N) if 0 <= k + 2 * m < N)
            R += (t - (m == 0)) ** 2
        return R / 2
    v = oracle.reg_term(a, wavelet=False)
    assert v == pytest.approx(dbl(), rel=1e-5)
    assert v != dbl()


@pytest.mark.parametrize("a", [F.HAAR, F.DB4, F.DB8])
@pytest.mark.parametrize("n", [2, 16, 64])
def test_perfect_reconstruction_1d(oracle, a, n):
    w = oracle.Wavenet()
    w.set_filter(a)
    x = np.random.default_rng(n).standard_normal(n)
    LO, HI = w.forward_1d(x)
    y = w.coeffs_1d(LO, HI)
    assert np.allclose(w.inverse_1d(y), x, atol=1e-12)
    assert np.allclose(oracle.band_inverse_1d(a, y), x, atol=1e-12)


def test_perfect_reconstruction_2d(oracle):
    w = oracle.Wavenet()
    w.set_filter(F.DB4)
    X = np.random.default_rng(0).standard_normal((8, 16))
    _, _, Y = w.forward_2d(X)
    assert np.allclose(w.inverse_2d(Y), X, atol=1e-12)
    assert np.allclose(oracle.band_inverse_2d(F.DB4, Y), X, atol=1e-12)


@pytest.mark.parametrize("N", [2, 4, 6, 16, 32])
@pytest.mark.parametrize("n", [1, 2, 4, 32, 128])
def test_dense_equals_banded_1d(oracle, N, n):
    a = F.random_filter(N, N + n)
    w = oracle.Wavenet()
    w.set_filter(a)
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    LO, HI = w.forward_1d(x)
    y = w.coeffs_1d(LO, HI)
    P, yb = oracle.band_forward_1d(a, x)
    assert np.allclose(y, yb, rtol=1e-13, atol=1e-13)
    m = int(np.log2(n))
    for i in range(m):
        assert np.allclose(P[(1 << i):(2 << i)], LO[(1 << i):(2 << i)], rtol=1e-13, atol=1e-13)
    delta = rng.standard_normal(n)
    g, din = w.backprop_1d(delta, LO, HI)
    gb, dinb = oracle.band_backprop_1d(a, delta, x, P)
    assert np.allclose(g, gb, rtol=1e-12, atol=1e-12)
    assert np.allclose(din, dinb, rtol=1e-12, atol=1e-12)
    assert np.allclose(w.inverse_1d(y), oracle.band_inverse_1d(a, y), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("N,R,C", [(2, 8, 8), (4, 8, 16), (16, 16, 4), (6, 32, 1), (4, 1, 1)])
def test_dense_equals_banded_2d(oracle, N, R, C):
    a = F.random_filter(N, 7)
    w = oracle.Wavenet()
    w.set_filter(a)
    rng = np.random.default_rng(R * C)
    X = rng.standard_normal((R, C))
    AR, AC, Y = w.forward_2d(X)
    Prow, Y1, Pcol, Yb = oracle.band_forward_2d(a, X)
    assert np.allclose(Y, Yb, rtol=1e-12, atol=1e-12)
    D = rng.standard_normal((R, C))
    g = w.backprop_2d(D, AR, AC)
    gb = oracle.band_backprop_2d(a, D, X, Prow, Y1, Pcol)
    assert np.allclose(g, gb, rtol=1e-11, atol=1e-11)
    assert np.allclose(w.inverse_2d(Y), oracle.band_inverse_2d(a, Y), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("N,shape", [(2, (32, 1)), (4, (32, 1)), (16, (32, 1)), (4, (8, 16))])
def test_backprop_matches_finite_difference(oracle, N, shape):
    """d/da SparseTerm(forward(X; a)) == backpropagate(SparseTermDeriv(Y))."""
    a = F.random_filter(N, 11)
    rng = np.random.default_rng(5)
    X = rng.standard_normal(shape)
    w = oracle.Wavenet()
    w.set_filter(a)
    AR, AC, Y = w.forward_2d(X)
    g = w.backprop_2d(oracle.sparse_term_deriv(Y), AR, AC)

    def S(af):
        ww = oracle.Wavenet()
        ww.set_filter(af)
        return oracle.sparse_term(ww.forward_2d(X)[2])
    h = 1e-6
    for k in range(N):
        ap, am = a.copy(), a.copy()
        ap[k] += h
        am[k] -= h
        assert g[k] == pytest.approx((S(ap) - S(am)) / (2 * h), abs=5e-9)


def test_train_bookkeeping(oracle):
    """Batch queue, cost log and filter log follow Wavenet.cxx:184-242, 717-765."""
    rng = np.random.default_rng(2)
    a0 = F.perturbed(F.DB4, 1e-2, 1)
    w = oracle.Wavenet(lam=10.0, alpha=1e-3)
    w.set_batch_size(4)
    w.set_inertia(0.5)
    w.set_filter(a0)
    Xs = rng.random((8, 8, 8))
    grads, costs = [], []
    for e in range(4):
        ww = oracle.Wavenet(lam=10.0)
        ww.set_filter(a0)
        AR, AC, Y = ww.forward_2d(Xs[e])
        gS = ww.backprop_2d(oracle.sparse_term_deriv(Y), AR, AC)
        grads.append(gS + 10.0 * oracle.reg_term_deriv(a0))
        costs.append(oracle.sparse_term(Y) + 10.0 * oracle.reg_term(a0))
        assert w.train(Xs[e])
    assert w.batch_queue_size() == 0
    gbar = np.sum(grads, axis=0) / 4.0
    a1 = a0 + (-1e-3 * gbar)
    assert np.allclose(w.filter(), a1, rtol=1e-14, atol=0)
    cl = w.cost_log()
    assert len(cl) == 2 and cl[1] == 0
    assert cl[0] == pytest.approx(np.sum(costs) / 4.0, rel=1e-14)
    assert len(w.filter_log()) == 2
    # second batch exercises the momentum term
    for e in range(4, 8):
        w.train(Xs[e])
    p1 = -1e-3 * gbar
    assert not np.allclose(w.momentum(), p1)
    assert np.allclose(w.filter() - a1, w.momentum(), rtol=1e-12, atol=1e-15)


def test_two_tap_training_converges_to_haar(oracle):
    """reference examples/Example01.cxx:100-113: N=2 training ends at [1/sqrt2, 1/sqrt2]."""
    rng = np.random.default_rng(0)
    E = 600
    Xs = np.zeros((E, 16, 1))
    for e in range(E):  # needle-like inputs
        Xs[e, rng.integers(0, 16, 2), 0] = rng.random(2) + 0.5
    w = oracle.Wavenet(lam=10.0, alpha=0.01)
    w.set_batch_size(10)
    init = np.array([0.6, 0.8])
    w.coach_run(init, Xs, E)
    assert np.allclose(w.filter(), F.HAAR, atol=2e-2)
    cl = w.cost_log()
    assert len(cl) == E // 10
    assert cl[-1] < cl[0]


def test_coach_annealing_lambda_schedule(oracle):
    """lambda = lambda0 f/(2-f)^2 with f through a float (Coach.cxx:176-180)."""
    Xs = np.random.default_rng(1).random((10, 8, 1))
    w = oracle.Wavenet(lam=10.0, alpha=1e-3)
    w.set_batch_size(1)
    w.coach_run(F.perturbed(F.DB4, 1e-2, 0), Xs, 10, annealing=True)
    f = np.float64(np.float32(9) / np.float32(10))  # int / float is a float division
    assert w.lam() == pytest.approx(10.0 * f / (2 - f) ** 2, rel=1e-15)
    assert len(w.cost_log()) == 10
