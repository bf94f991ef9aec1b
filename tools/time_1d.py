"""Times wn_train_batch for 1D 1024-sample signals on different inputs (GPU)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from wavenet_amd import Context, generators as G
from wavenet_amd.filters import daubechies, perturbed
from wavenet_amd.generators import to_device_layout
B, n = 4096, 1024
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
inputs = {"normal": rng.standard_normal((B, n, 1)), "bumps": G.bumps(B, n, seed=1000).reshape(B, n, 1), "uniform": rng.random((B, n, 1))}
for N in (int(v) for v in (sys.argv[1:] or ["4", "8", "16", "32"])):
    ctx.set_filter(perturbed(daubechies(N), 1e-2, seed=7))
    for name, X in inputs.items():
        dX = torch.from_numpy(to_device_layout(X)).cuda()
        for _ in range(5):
            ctx.train_batch(dX, B, n, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            ctx.train_batch(dX, B, n, 1)
        torch.cuda.synchronize()
        print(N, name, "%.4f ms" % ((time.perf_counter() - t0) / 50 * 1e3), flush=True)
