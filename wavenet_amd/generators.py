"""Deterministic synthetic inputs for the Wavenet training path.

The reference's generators (include/Wavenet/Generators.h) reseed Armadillo's RNG from entropy
on every reset (Generators.h:66,100,170), so "identical seeds" cannot be had from them.  These
are counter-based: every value is a pure function of (seed, example, item) through splitmix64,
uses no libm call, and can therefore be regenerated bit-for-bit anywhere.

Arrays are returned as (B, R, C) float64 in natural (row, col) indexing.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _hash(seed, *counters):
    """64-bit hash of the seed and broadcastable integer counters."""
    with np.errstate(over="ignore"):
        h = _splitmix64(np.uint64(seed))
        for c in counters:
            h = _splitmix64(h ^ (np.asarray(c).astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93) & _M64))
        return h


def _unit(h):
    """uint64 -> float64 in [0, 1) by exact scaling of the top 53 bits."""
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(B, R, C, seed=1):
    """UniformGenerator-like (Generators.h:87-96): independent U[0,1) pixels."""
    e, r, c = np.ogrid[0:B, 0:R, 0:C]
    return _unit(_hash(seed, e, r * C + c))


def needle(B, R, C, seed=1, threshold=0.995):
    """NeedleGenerator-like (Generators.h:50-62): U[0,1) pixels, everything below the threshold zeroed;
    one pixel is kept per image if that leaves it empty."""
    X = uniform(B, R, C, seed)
    X = np.where(X < threshold, 0.0, X)
    empty = X.reshape(B, -1).max(axis=1) == 0
    if empty.any():
        idx = np.nonzero(empty)[0]
        pos = (_hash(seed ^ 0x5EED, idx) % np.uint64(R * C)).astype(np.int64)
        X.reshape(B, -1)[idx, pos] = 0.995 + 0.005 * _unit(_hash(seed ^ 0xBEEF, idx))
    return X


def _poisson_table(mean, size=4096):
    """Inverse CDF of Poisson(mean) sampled at (i + 0.5)/size, built with exact rational-free recurrences."""
    kmax = int(mean + 12 * np.sqrt(mean) + 20)
    logp = np.empty(kmax + 1)
    logp[0] = -mean
    for k in range(1, kmax + 1):
        logp[k] = logp[k - 1] + np.log(mean / k)
    cdf = np.cumsum(np.exp(logp))
    u = (np.arange(size) + 0.5) / size
    return np.searchsorted(cdf, u).astype(np.int64)


_EXP_TABLE = -np.log((np.arange(1024) + 0.5) / 1024.0)          # unit-mean exponential spectrum
_GAUSS_OFFSETS = np.array([-3, -2, -2, -1, -1, -1, -1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3])


def calorimeter(B, R, C, seed=1, mean_deposits=None, jets=2, jet_deposits=30):
    """Calorimeter-style eta-phi images in the spirit of HepMCGenerator (Generators.h:377-393: a TH2F over
    eta x phi filled with pT/1000 of visible particles).  Each image has K ~ Poisson(mean_deposits) soft
    deposits in uniformly drawn bins with exponentially distributed energies, plus `jets` clusters of
    `jet_deposits` harder deposits spread over +-3 bins.  Deposits in one bin add.  Bin contents are
    rounded to float as TH2F contents are.  Default occupancy is about 5 %.
    """
    if mean_deposits is None:
        mean_deposits = max(1.0, 200.0 * (R * C) / 4096.0)
    ptab = _poisson_table(float(mean_deposits))
    dmax = int(ptab.max())
    e = np.arange(B)[:, None]
    K = ptab[(_hash(seed, e[:, 0], 1) % np.uint64(len(ptab))).astype(np.int64)]
    d = np.arange(dmax)[None, :]
    live = d < K[:, None]
    h = _hash(seed, e, 2 + d)
    rows = (h % np.uint64(R)).astype(np.int64)
    cols = ((h >> np.uint64(20)) % np.uint64(C)).astype(np.int64)
    en = 0.5 * _EXP_TABLE[((h >> np.uint64(40)) % np.uint64(1024)).astype(np.int64)]
    X = np.zeros((B, R * C))
    flat = rows * C + cols
    eb = np.broadcast_to(e, flat.shape)
    np.add.at(X, (eb[live], flat[live]), en[live])
    for j in range(jets):
        hc = _hash(seed, e[:, 0], 1000 + j)
        r0 = (hc % np.uint64(R)).astype(np.int64)[:, None]
        c0 = ((hc >> np.uint64(20)) % np.uint64(C)).astype(np.int64)[:, None]
        dd = np.arange(jet_deposits)[None, :]
        hj = _hash(seed, e, 2000 + 100 * j + dd)
        dr = _GAUSS_OFFSETS[(hj % np.uint64(len(_GAUSS_OFFSETS))).astype(np.int64)]
        dc = _GAUSS_OFFSETS[((hj >> np.uint64(16)) % np.uint64(len(_GAUSS_OFFSETS))).astype(np.int64)]
        ej = 4.0 * _EXP_TABLE[((hj >> np.uint64(40)) % np.uint64(1024)).astype(np.int64)]
        rr = (r0 + dr) % R
        cc = (c0 + dc) % C
        ebj = np.broadcast_to(e, rr.shape)
        np.add.at(X, (ebj.reshape(-1), (rr * C + cc).reshape(-1)), ej.reshape(-1))
    X = X.astype(np.float32).astype(np.float64)
    return X.reshape(B, R, C)


GAUSS_EXP_TABLE = np.exp(-np.arange(4097) / 128.0)       # exp(-t) at t = i / 128, interpolated linearly


def _table_exp_neg(t):
    x = t * 128.0
    inside = x < 4096.0
    xi = np.where(inside, x, 0.0)
    i = xi.astype(np.int64)
    fr = xi - i
    return np.where(inside, GAUSS_EXP_TABLE[i] + fr * (GAUSS_EXP_TABLE[i + 1] - GAUSS_EXP_TABLE[i]), 0.0)


def gaussian(B, R, C, seed=1):
    """GaussianGenerator-like (Generators.h:122-166): one to three unit-height Gaussian bumps of random centre and
    width per example; exp from a table with linear interpolation, as on the device and in the C++ host layer."""
    X = np.zeros((B, R, C))
    r = np.arange(R, dtype=np.float64)[:, None]
    c = np.arange(C, dtype=np.float64)[None, :]
    for e in range(B):
        nb = 1 + int(_hash(seed, e, 0) % np.uint64(3))
        for b in range(nb):
            mr = float(_unit(_hash(seed, e, 10 + b))) * R
            mc = float(_unit(_hash(seed, e, 20 + b))) * C
            sr = 0.5 + 0.15 * R * float(_unit(_hash(seed, e, 30 + b)))
            sc = 0.5 + 0.15 * C * float(_unit(_hash(seed, e, 40 + b)))
            dr = (r - mr) / sr
            dc = (c - mc) / sc if C > 1 else np.zeros_like(c)
            X[e] += _table_exp_neg(0.5 * (dr * dr + dc * dc))
    return X


def bumps(B, n, seed=1):
    """1D signals: U[0,1) noise plus one raised-cosine bump of random position, width and height."""
    X = uniform(B, n, 1, seed)[:, :, 0]
    e = np.arange(B)
    h = _hash(seed ^ 0xB0B, e)
    centre = (h % np.uint64(n)).astype(np.int64)[:, None]
    width = (8 + (h >> np.uint64(24)) % np.uint64(max(1, n // 8))).astype(np.int64)[:, None]
    height = 1.0 + 4.0 * _unit(_hash(seed ^ 0xCAFE, e))[:, None]
    t = np.arange(n)[None, :]
    dist = np.minimum((t - centre) % n, (centre - t) % n)
    # raised cosine from a lookup so that no libm call is needed at regeneration time elsewhere
    table = 0.5 * (1.0 + np.cos(np.pi * np.arange(1025) / 1024.0))
    idx = np.minimum(1024, (dist * 1024) // width)
    X = X + height * table[idx] * (dist < width)
    return X.reshape(B, n, 1)


def to_device_layout(X):
    """(B, R, C) natural indexing -> flat float64 vector, column-major per example (the C ABI's layout)."""
    X = np.asarray(X, dtype=np.float64)
    return np.ascontiguousarray(np.transpose(X, (0, 2, 1))).reshape(-1)
