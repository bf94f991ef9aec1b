"""Times the staged entry points on one batch to see which phase of the fused step dominates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wavenet_amd import Context, generators as G
from wavenet_amd.filters import daubechies, perturbed

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    n = R * C
    X = G.calorimeter(B, R, C, seed=1000) if C > 1 else G.bumps(B, R, seed=1)
    dX = torch.from_numpy(G.to_device_layout(X)).cuda()
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    e = lambda k=B * n: torch.empty(k, dtype=torch.float64, device="cuda")
    dY, dPr, dY1, dPc, dD, dXr = e(), e(), e(), e(), e(), e()
    dS = e(B)
    for N in (2, 4, 16, 32):
        ctx.set_filter(perturbed(daubechies(N), 1e-2, 7))
        dG = e(B * N)
        ctx.forward(dX, B, R, C, dY, dPr, dY1, dPc)
        ctx.sparse_term_deriv(dY, B, n, dD, dS)
        t_f = timeit(lambda: ctx.forward(dX, B, R, C, dY))
        t_fa = timeit(lambda: ctx.forward(dX, B, R, C, dY, dPr, dY1, dPc))
        t_s = timeit(lambda: ctx.sparse_term_deriv(dY, B, n, dD, dS))
        t_s0 = timeit(lambda: ctx.sparse_term(dY, B, n, dS))
        t_b = timeit(lambda: ctx.backprop(dD, dX, dPr, dY1, dPc, B, R, C, dG))
        t_i = timeit(lambda: ctx.inverse(dY, B, R, C, dXr))
        t_t = timeit(lambda: ctx.train_batch(dX, B, R, C))
        print(f"N={N:2d} {R}x{C} B={B}: forward {t_f:.3f} (+acts {t_fa:.3f})  sparse_deriv {t_s:.3f} (term only {t_s0:.3f})  "
              f"backprop {t_b:.3f}  inverse {t_i:.3f}  train {t_t:.3f} ms", flush=True)

main()
