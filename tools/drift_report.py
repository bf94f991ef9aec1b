"""Drift of the learned filter over many updates: the device path against the banded CPU restatement
(oracle/), both free-running from the same start on the same examples, and the one-step error when the
device is put back on the CPU filter before every update ("per step").  Writes one JSON object per
configuration to stdout; profiles/r01_drift.json holds a committed run.

    python tools/drift_report.py [updates]

The update of the CPU side follows flushBatchQueue_ / update_ (reference src/Wavenet.cxx:717-765) with
no inertia: gradient = (sum of example gradients + B lambda RegTermDeriv) / float(B); a -= alpha gradient.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(R, C, N, batch, updates, marks, seed=5):
    import torch
    from oracle import oracle as orc   # checker
    from wavenet_amd import Context
    from wavenet_amd import generators as G
    from wavenet_amd.filters import daubechies, perturbed
    from wavenet_amd.generators import to_device_layout

    lam, alpha = 10.0, 1e-3
    pool = 64                                    # examples, reused cyclically
    X = G.calorimeter(pool, R, C, seed=seed) if C > 1 else G.bumps(pool, R, seed=seed).reshape(pool, R, 1)
    dX = torch.from_numpy(to_device_layout(X)).cuda()
    n = R * C
    a0 = perturbed(daubechies(N), 1e-2, seed=7)

    free = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    forced = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    free.set_filter(a0)
    a = a0.copy()
    drift, step_err = {}, 0.0
    for u in range(1, updates + 1):
        e0 = ((u - 1) * batch) % pool
        idx = [(e0 + i) % pool for i in range(batch)]
        sel = X[idx]
        dsel = torch.cat([dX[i * n:(i + 1) * n] for i in idx])
        # CPU update
        g, s, bad = orc.band_train_batch(a, sel)
        assert bad == 0
        gbar = (g + batch * lam * orc.reg_term_deriv(a)) / np.float32(batch)
        a_next = a - alpha * gbar
        # device, put on the CPU filter first: one-step error
        forced.set_filter(a)
        forced.set_momentum(np.zeros(N))
        forced.train_batch(dsel, batch, R, C)
        forced.apply_update(batch, batch * lam, alpha, 0.0, True)
        step_err = max(step_err, float(np.abs(forced.filter() - a_next).max() / np.abs(a_next).max()))
        # device, free-running
        free.train_batch(dsel, batch, R, C)
        free.apply_update(batch, batch * lam, alpha, 0.0, True)
        a = a_next
        if u in marks:
            f = free.filter()
            drift[str(u)] = float(np.abs(f - a).max() / np.abs(a).max())
    out = {"shape": [R, C], "taps": N, "batch": batch, "updates": updates, "lambda": lam, "alpha": alpha,
           "max_one_step_relative_error": step_err, "free_running_relative_drift_after_updates": drift,
           "final_filter_cpu": [float(v) for v in a]}
    free.close()
    forced.close()
    return out


if __name__ == "__main__":
    updates = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    marks = [m for m in (10, 100, 1000) if m <= updates]
    for R, C, N, batch in ((64, 64, 4, 8), (1024, 1, 16, 8), (512, 512, 4, 2)):
        ups = updates if R * C <= 4096 else min(updates, 100)
        print(json.dumps(run(R, C, N, batch, ups, [m for m in marks if m <= ups])), flush=True)
