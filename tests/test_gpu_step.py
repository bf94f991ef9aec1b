"""wn_train_step (a whole batch of Coach's loop in one submission; for examples of up to 4096 samples the
batch reduction and the filter update run inside the training kernel, in the workgroup that finishes last)
against wn_train_batch + wn_apply_update, through the C ABI: same arithmetic in the same order, so filters,
costs and momentum must be bit-identical over a run of updates."""
import numpy as np
import pytest

import filters as F
from gpu_util import to_dev

pytestmark = pytest.mark.gpu


def _ctx():
    import torch
    from wavenet_amd import Context
    return Context(0, stream=torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("R,C,N,B", [(64, 64, 4, 1500), (64, 64, 16, 700), (64, 64, 32, 130), (1024, 1, 16, 1200), (512, 1, 4, 900),
                                     (256, 1, 8, 1100), (16, 32, 8, 700), (8, 8, 2, 3000), (64, 32, 6, 300), (128, 64, 4, 5)])
def test_train_step_matches_three_calls(R, C, N, B):
    import torch
    a = F.perturbed(F.daubechies(N), 1e-2, 3) if N in (2, 4, 8, 16) else F.random_filter(N, 3)
    rng = np.random.default_rng(R * 3 + C + N)
    X = np.where(rng.random((B, R, C)) < 0.6, 0.0, rng.random((B, R, C)))
    X[:, 0, 0] += 0.5
    dX = to_dev(X)
    n = R * C
    legacy, fused = _ctx(), _ctx()
    for c in (legacy, fused):
        c.set_filter(a)
    lam, alpha = 10.0, 2e-3
    sizes = [B, max(1, B // 3), 1, B, max(1, B // 7), B, 2, B, B]
    for step, Bs in enumerate(sizes):
        off = (step * 13) % (B - Bs + 1)
        inertia = 0.5 * (1 - np.exp(-step / 3.0))
        legacy.train_batch(dX[off * n:], Bs, R, C)
        legacy.apply_update(Bs, lam * Bs, alpha, inertia, True)
        fused.train_step(dX[off * n:], Bs, R, C, Bs, lam * Bs, alpha, inertia, True)
    torch.cuda.synchronize()
    assert fused.num_updates() == legacy.num_updates() == len(sizes)
    Fl, Cl = legacy.logs()
    Ff, Cf = fused.logs()
    assert np.isfinite(Fl).all() and np.isfinite(Cl).all()
    assert np.array_equal(Ff, Fl) and np.array_equal(Cf, Cl)
    assert np.array_equal(fused.momentum(), legacy.momentum())
    assert np.array_equal(fused.filter(), legacy.filter())
    assert np.array_equal(fused.sums(), legacy.sums())
    legacy.close()
    fused.close()


def test_train_step_many_in_a_row_under_load():
    """Several hundred one-launch steps queued back to back (each kernel's last workgroup reads what all the
    others wrote, and the next kernel starts right behind it) against the three-call form."""
    import torch
    R = C = 64
    N, B = 4, 4096
    a = F.perturbed(F.DB4, 1e-2, 1)
    legacy, fused = _ctx(), _ctx()
    X = torch.empty(B * R * C, dtype=torch.float64, device="cuda")
    legacy.generate_calorimeter(X, B, R, C, seed=3)
    for c in (legacy, fused):
        c.set_filter(a)
    steps = 300
    for s in range(steps):
        Bs = B - (s % 5) * 100
        fused.train_step(X, Bs, R, C, Bs, 10.0 * Bs, 1e-3, 0.0, True)
    for s in range(steps):
        Bs = B - (s % 5) * 100
        legacy.train_batch(X, Bs, R, C)
        legacy.apply_update(Bs, 10.0 * Bs, 1e-3, 0.0, True)
    torch.cuda.synchronize()
    Fl, Cl = legacy.logs()
    Ff, Cf = fused.logs()
    assert Ff.shape == (steps, N)
    assert np.array_equal(Ff, Fl) and np.array_equal(Cf, Cl)
    legacy.close()
    fused.close()


def test_train_step_rejects_a_diverging_batch():
    import torch
    from wavenet_amd.capi import WN_ERR_NONFINITE, WavenetError
    R = C = 64
    B = 64
    c = _ctx()
    c.set_filter(F.DB4)
    X = torch.empty(B * R * C, dtype=torch.float64, device="cuda")
    c.generate_uniform(X, B, R, C, seed=1)
    for _ in range(12):
        c.train_step(X, B, R, C, B, 10.0 * B, 1e6, 0.0, True)
    with pytest.raises(WavenetError) as err:
        c.num_updates()
    assert err.value.status == WN_ERR_NONFINITE
    assert np.isfinite(c.filter()).all()
    c.set_filter(F.DB4)          # takes the flag down
    c.train_step(X, B, R, C, B, 10.0 * B, 1e-3, 0.0, True)
    assert c.num_updates() >= 1
    c.close()


def test_get_updates_is_the_logs_the_momentum_and_the_verdict_in_one_readback():
    """wn_get_updates against wn_num_updates + wn_get_logs + wn_get_momentum: a window of the log that starts after
    the first updates, a window larger than what exists, the log ring wrapped around (more than 16384 updates), and
    the refusal of a non-finite batch reported with the data still filled in."""
    import torch
    R, C, N, B = 8, 8, 4, 16
    a = F.perturbed(F.DB4, 1e-2, 3)
    rng = np.random.default_rng(5)
    dX = to_dev(rng.random((B, R, C)))
    c = _ctx()
    c.set_filter(a)
    for _ in range(40):
        c.train_step(dX, B, R, C, B, 10.0 * B, 1e-3, 0.3, True)
    Fl, Cl = c.logs()
    steps, Fw, Cw, P, rejected = c.updates(first=25, max_count=100)
    assert steps == 40 and not rejected
    assert np.array_equal(Fw, Fl[25:]) and np.array_equal(Cw, Cl[25:]) and np.array_equal(P, c.momentum())
    steps, Fw, Cw, P, _ = c.updates(first=0, max_count=7)
    assert steps == 40 and np.array_equal(Fw, Fl[:7]) and np.array_equal(Cw, Cl[:7])
    # the ring of 16384 entries wrapped: the newest entries are still there
    for _ in range(16400):
        c.train_step(dX, B, R, C, B, 10.0 * B, 1e-6, 0.0, True)
    steps, Fw, Cw, P, _ = c.updates(first=16430, max_count=10)
    assert steps == 16440
    F2, C2 = c.logs(first=16430, count=10)
    assert np.array_equal(Fw, F2) and np.array_equal(Cw, C2) and np.array_equal(Fw[-1], c.filter())
    # a batch that is refused
    bad = rng.random((B, R, C))
    bad[3, 2, 2] = np.inf
    c.train_step(to_dev(bad), B, R, C, B, 10.0 * B, 1e-3, 0.0, True)
    steps, Fw, Cw, P, rejected = c.updates(first=16438, max_count=10)
    assert rejected and steps == 16440 and len(Fw) == 2 and np.array_equal(Fw[-1], c.filter())
    c.close()
