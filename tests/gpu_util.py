"""Helpers for the GPU parity tests: torch is used for device memory only."""
import numpy as np


def to_dev(arr_brc):
    """(B, R, C) numpy, natural indexing -> flat device tensor, column-major per example."""
    import torch
    a = np.ascontiguousarray(np.transpose(np.asarray(arr_brc, dtype=np.float64), (0, 2, 1))).reshape(-1)
    return torch.from_numpy(a).cuda()


def from_dev(t, B, R, C):
    a = t.cpu().numpy().reshape(B, C, R)
    return np.transpose(a, (0, 2, 1)).copy()


def dev_empty(n):
    import torch
    return torch.empty(n, dtype=torch.float64, device="cuda")


def relerr(a, b):
    """Largest element-wise relative error of a against b, with an absolute floor of 1e-4 of b's
    largest magnitude: relerr(a, b) <= t  <=>  |a - b| <= t |b| + 1e-4 t max|b| for every element.
    At t = 1e-9 that is the stated tolerance on coefficients and activations, with 1e-13 max|b| of
    slack for entries that are small only by cancellation."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    return (np.abs(a - b) / (np.abs(b) + 1e-4 * scale)).max()
