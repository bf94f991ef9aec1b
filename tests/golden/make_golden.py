"""Generates the golden vectors under tests/golden/ (run from the repository root:
`python tests/golden/make_golden.py`).

The reference cannot run in this image (Armadillo/BLAS absent) and holds no fixtures, so these
vectors do NOT come from the reference: they are produced by an independent, literal numpy
restatement of its dense algorithm written below (dense operator matrices, matrix-vector cascade,
outer product + Frobenius product against 0/+-1 weight matrices, sort-based Gini) -- a second
implementation next to oracle/wavenet_oracle.c.  tests/test_golden.py checks the C oracle against
them on the CPU and the HIP path against them on the GPU.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import filters as F  # noqa: E402


def operator(f, size):
    """MatrixOperator::constructByIndices_ (src/MatrixOperator.cxx:63-113) in numpy."""
    N = len(f)
    nr, nc = 1 << size, 2 << size
    M = np.zeros((nr, nc))
    cols = (N // 2 + 2 * np.arange(nr)) % nc
    rows = np.arange(nr)
    for i in range(N):
        np.add.at(M, (rows, cols), f[i])
        cols = np.where(cols == 0, cols + nc, cols) - 1
    return M


def highpass(a):
    N = len(a)
    return np.array([(-1.0) ** i * a[N - 1 - i] for i in range(N)])


def forward_1d(a, x):
    """Wavenet::forward_(Col) (src/Wavenet.cxx:404-445): returns {level: (lo, hi)} and coefficients."""
    n = len(x)
    m = int(np.log2(n))
    b = highpass(a)
    act = {m: (x.copy(), np.zeros(n))}
    cur = x.copy()
    for i in range(m - 1, -1, -1):
        lo = operator(a, i) @ cur
        hi = operator(b, i) @ cur
        act[i] = (lo, hi)
        cur = lo
    y = np.zeros(n)
    y[0] = act[0][0][0] if m > 0 else x[0]
    for i in range(m):
        y[(1 << i):(2 << i)] = act[i][1]
    return act, y


def backprop_1d(a, delta, act):
    """Wavenet::backpropagate_(Col, Activations1D_t) (src/Wavenet.cxx:474-556)."""
    N = len(a)
    n = len(delta)
    m = int(np.log2(n))
    b = highpass(a)
    grad = np.zeros(N)
    dLP = np.array([delta[0]])
    dHP = np.array([delta[1]]) if n > 1 else np.zeros(1)
    for i in range(m):
        activ = act[i + 1][0]
        eLP = np.outer(dLP, activ)
        eHP = np.outer(dHP, activ)
        for k in range(N):
            t = np.zeros(N)
            t[k] = 1.0
            grad[k] += np.sum(operator(t, i) * eLP)
            grad[k] += np.sum(operator(highpass(t), i) * eHP)
        dLP = operator(a, i).T @ dLP + operator(b, i).T @ dHP
        dHP = delta[(2 << i):(4 << i)] if i != m - 1 else np.zeros(len(dLP))
    return grad, dLP


def forward_2d(a, X):
    R, C = X.shape
    Y = X.copy()
    rows, cols = [], []
    for r in range(R):
        act, y = forward_1d(a, Y[r, :].copy())
        Y[r, :] = y
        rows.append(act)
    for c in range(C):
        act, y = forward_1d(a, Y[:, c].copy())
        Y[:, c] = y
        cols.append(act)
    return rows, cols, Y


def backprop_2d(a, Delta, rows, cols):
    R, C = Delta.shape
    X = Delta.copy()
    grad = np.zeros(len(a))
    for c in range(C):
        g, din = backprop_1d(a, X[:, c].copy(), cols[c])
        grad += g
        X[:, c] = din
    for r in range(R):
        g, _ = backprop_1d(a, X[r, :].copy(), rows[r])
        grad += g
    return grad


def sparse_term(c):
    c = np.sort(np.abs(np.ravel(c, order="F")))
    n = len(c)
    idx = np.arange(1, n + 1)
    return 1.0 - np.dot(2 * idx - n - 1, c) / (n * c.sum())


def sparse_term_deriv(C):
    c = np.ravel(C, order="F")
    n = len(c)
    order = np.lexsort((np.arange(n), np.abs(c)))  # ascending |c|, ties by index
    idx = np.arange(1, n + 1)
    cs = np.abs(c)[order]
    f = np.dot(2 * idx - n - 1, cs)
    g = n * cs.sum()
    df = np.zeros(n)
    df[order] = 2 * idx - n - 1
    sgn = np.sign(c)
    out = -(df * sgn * g - f * n * sgn) / g ** 2
    return out.reshape(C.shape, order="F")


def seqsum(v):
    """Left-to-right sum (numpy's pairwise sum rounds differently)."""
    s = 0.0
    for x in v:
        s += float(x)
    return s


def reg_term_deriv(a):
    """RegTermDeriv with doWavelet (src/CostFunctions.cxx:259-415), all double."""
    N = len(a)
    b = highpass(a)
    g = np.zeros(N)

    def block(v):
        out = np.zeros(N)
        for m in range(-N // 2, N // 2 + 1):
            pre = sum(v[k] * v[k + 2 * m] for k in range(N) if 0 <= k + 2 * m < N) - (m == 0)
            inner = np.array([(a[i + 2 * m] if 0 <= i + 2 * m < N else 0.0) +
                              (a[i - 2 * m] if 0 <= i - 2 * m < N else 0.0) for i in range(N)])
            out += 2 * pre * inner
        return out
    g += block(a)
    g += 2 * (seqsum(a) - np.sqrt(2.0))
    g += block(b)
    g += 2 * seqsum(b) * np.array([(-1.0) ** (N - i - 1) for i in range(N)])
    return g


def reg_term(a):
    """RegTerm with the float accumulators of src/CostFunctions.cxx:140-160, 188-208."""
    N = len(a)
    b = highpass(a)

    def block(v):
        R = np.float32(0)
        for m in range(-N // 2, N // 2 + 1):
            term = np.float32(0)
            for k in range(N):
                if 0 <= k + 2 * m < N:
                    term = np.float32(np.float64(term) + v[k] * v[k + 2 * m])
            dev = np.float32(term - np.float32(m == 0))
            R = np.float32(R + np.float32(dev * dev))
        return np.float64(R)
    R = 0.0
    R += block(a)
    R += (seqsum(a) - np.sqrt(2.0)) ** 2
    R += block(b)
    R += seqsum(b) ** 2
    return R / 2.0


CASES = [  # name, R, C, N, seed
    ("1d64_n2", 64, 1, 2, 0), ("1d64_n4", 64, 1, 4, 1), ("1d64_n16", 64, 1, 16, 2),
    ("2d8x16_n4", 8, 16, 4, 0), ("2d8x16_n16", 8, 16, 16, 1), ("2d16x16_n2", 16, 16, 2, 2),
    ("2d16x8_n6", 16, 8, 6, 3),
]


def filt(N, seed):
    base = F.daubechies(N) if N in (2, 4, 8, 16) else F.random_filter(N, seed)
    return F.perturbed(base, 1e-2, seed)


def main():
    for name, R, C, N, seed in CASES:
        rng = np.random.default_rng(100 + seed)
        a0 = filt(N, seed)
        B = 4
        X = rng.random((B, R, C))
        Y = np.zeros_like(X)
        D = np.zeros_like(X)
        G = np.zeros((B, N))
        S = np.zeros(B)
        for e in range(B):
            rows, cols, Ye = forward_2d(a0, X[e])
            De = sparse_term_deriv(Ye)
            Y[e], D[e], S[e] = Ye, De, sparse_term(Ye)
            G[e] = backprop_2d(a0, De, rows, cols)
        # 5 updates of batch 4: lambda 10, alpha 1e-3, no inertia
        steps, bs, lam, alpha = 5, 4, 10.0, 1e-3
        Xt = rng.random((steps * bs, R, C))
        a = a0.copy()
        flog, clog = [], []
        for s in range(steps):
            gsum = np.zeros(N)
            cost = 0.0
            for e in range(bs):
                rows, cols, Ye = forward_2d(a, Xt[s * bs + e])
                gsum += backprop_2d(a, sparse_term_deriv(Ye), rows, cols) + lam * reg_term_deriv(a)
                cost += sparse_term(Ye) + lam * reg_term(a)
            a = a + (-alpha * (gsum / np.float64(np.float32(bs))))
            flog.append(a.copy())
            clog.append(cost / np.float64(np.float32(bs)))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), filter=a0, X=X, Y=Y, Delta=D, grad=G, S=S,
                            reg=reg_term(a0), reg_deriv=reg_term_deriv(a0), train_X=Xt,
                            filter_log=np.array(flog), cost_log=np.array(clog),
                            lam=lam, alpha=alpha, batch=bs)
        print("wrote", name)


if __name__ == "__main__":
    main()
