"""Initial filters: Daubechies low-pass taps in the reference's convention (sum a = sqrt 2)."""
from math import comb

import numpy as np


def daubechies(N):
    """Minimum-phase Daubechies filter with N taps (N/2 vanishing moments).

    Spectral factorisation of P(y) = sum_k C(p-1+k, k) y^k, y = (2 - z - 1/z)/4, p = N/2.
    RegTerm of the result is zero to rounding (it satisfies conditions C1-C4 of
    reference src/CostFunctions.cxx:134-218).
    """
    if N < 2 or N % 2:
        raise ValueError("N must be even and >= 2")
    p = N // 2
    if p == 1:
        return np.array([1.0, 1.0]) / np.sqrt(2.0)
    coeffs = [comb(p - 1 + k, k) for k in range(p)]
    zs = []
    for y in np.roots(coeffs[::-1]):
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
    return h * (np.sqrt(2.0) / h.sum())


def perturbed(base, eps, seed):
    """base * (1 + eps * xi) with seeded standard-normal xi: keeps |coefficients| tie-free."""
    rng = np.random.default_rng(seed)
    return np.asarray(base) * (1.0 + eps * rng.standard_normal(len(base)))
