"""The committed golden vectors (tests/golden/*.npz, produced by the independent numpy restatement in
tests/golden/make_golden.py) against the C oracle on the CPU and against the HIP path on the GPU."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
IDS = [os.path.basename(c)[:-4] for c in CASES]


def load(path):
    return np.load(path, allow_pickle=False)


def test_fixtures_present():
    assert len(CASES) >= 7


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_oracle_matches_golden(oracle, path):
    g = load(path)
    a = g["filter"]
    w = oracle.Wavenet()
    w.set_filter(a)
    for e in range(len(g["X"])):
        AR, AC, Y = w.forward_2d(g["X"][e])
        assert np.allclose(Y, g["Y"][e], rtol=1e-12, atol=1e-13)
        D = oracle.sparse_term_deriv(Y)
        assert np.allclose(D, g["Delta"][e], rtol=1e-10, atol=1e-13)
        assert oracle.sparse_term(Y) == pytest.approx(float(g["S"][e]), rel=1e-12)
        assert np.allclose(w.backprop_2d(D, AR, AC), g["grad"][e], rtol=1e-10, atol=1e-13)
    assert oracle.reg_term(a) == float(g["reg"])          # float accumulators reproduced bit for bit
    assert np.allclose(oracle.reg_term_deriv(a), g["reg_deriv"], rtol=1e-13, atol=1e-15)
    w = oracle.Wavenet(lam=float(g["lam"]), alpha=float(g["alpha"]))
    w.set_batch_size(int(g["batch"]))
    w.set_filter(a)
    for X in g["train_X"]:
        w.train(X)
    assert np.allclose(w.filter_log()[1:], g["filter_log"], rtol=1e-10, atol=0)
    assert np.allclose(w.cost_log()[:-1], g["cost_log"], rtol=1e-10, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_hip_matches_golden(path):
    import torch
    from gpu_util import dev_empty, from_dev, relerr, to_dev
    from wavenet_amd import Context
    g = load(path)
    a = g["filter"]
    X = g["X"]
    B, R, C = X.shape
    N = len(a)
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_filter(a)
    dX = to_dev(X)
    dY, dD, dG, dS = dev_empty(B * R * C), dev_empty(B * R * C), dev_empty(B * N), dev_empty(B)
    ctx.forward(dX, B, R, C, dY)
    ctx.sparse_term_deriv(dY, B, R * C, dD)
    ctx.train_batch(dX, B, R, C, dG, dS)
    torch.cuda.synchronize()
    assert relerr(from_dev(dY, B, R, C), g["Y"]) < 1e-11
    assert relerr(from_dev(dD, B, R, C), g["Delta"]) < 1e-10
    assert relerr(dG.cpu().numpy().reshape(B, N), g["grad"]) < 1e-10
    assert np.allclose(dS.cpu().numpy(), g["S"], rtol=1e-12)
    # the 5-update training log
    bs, lam, alpha = int(g["batch"]), float(g["lam"]), float(g["alpha"])
    ctx.set_filter(a)
    ctx.set_momentum(np.zeros(N))
    ctx.clear_logs()
    dT = to_dev(g["train_X"])
    n = R * C
    for s in range(len(g["filter_log"])):
        ctx.train_batch(dT[s * bs * n:].data_ptr(), bs, R, C)
        ctx.apply_update(bs, bs * lam, alpha, 0.0, True)
    fl, cl = ctx.logs()
    assert np.allclose(fl, g["filter_log"], rtol=1e-9, atol=0)
    assert np.allclose(cl, g["cost_log"], rtol=1e-9, atol=0)
    ctx.close()
