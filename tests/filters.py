"""Filters and synthetic inputs shared by the tests."""
import numpy as np

SQ2 = np.sqrt(2.0)
HAAR = np.array([1.0, 1.0]) / SQ2
S3 = np.sqrt(3.0)
DB4 = np.array([1 + S3, 3 + S3, 3 - S3, 1 - S3]) / (4 * SQ2)
# Daubechies 8-tap (db4 in pywt naming), low-pass decomposition taps in the
# a_k convention sum(a) = sqrt(2).
DB8 = np.array([0.23037781330885523, 0.7148465705525415, 0.6308807679295904,
                -0.02798376941698385, -0.18703481171888114, 0.030841381835986965,
                0.032883011666982945, -0.010597401784997278])


def perturbed(base, eps, seed):
    rng = np.random.default_rng(seed)
    return base * (1.0 + eps * rng.standard_normal(base.shape))


def random_filter(N, seed):
    """A generic (non-orthonormal) filter near a point on the unit sphere scaled to sum ~ sqrt(2)."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(N)
    a /= np.linalg.norm(a)
    return a


def daubechies(N):
    """Minimum-phase Daubechies low-pass filter with N taps (N/2 vanishing moments), sum = sqrt(2).

    Spectral factorisation of P(y) = sum_k C(p-1+k, k) y^k with y = (2 - z - 1/z)/4.
    """
    from math import comb
    p = N // 2
    if p == 1:
        return HAAR.copy()
    coeffs = [comb(p - 1 + k, k) for k in range(p)]          # ascending powers of y
    yroots = np.roots(coeffs[::-1])
    zs = []
    for y in yroots:
        # z^2 - (2 - 4y) z + 1 = 0
        b = 2 - 4 * y
        d = np.sqrt(b * b - 4 + 0j)
        z1, z2 = (b + d) / 2, (b - d) / 2
        zs.append(z1 if abs(z1) < 1 else z2)
    h = np.array([1.0 + 0j])
    for _ in range(p):
        h = np.convolve(h, [1.0, 1.0])
    for z in zs:
        h = np.convolve(h, [1.0, -z])
    h = np.real(h)
    return h * (SQ2 / h.sum())
