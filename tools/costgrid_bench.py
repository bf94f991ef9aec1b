"""Times wn_sparse_cost_grid on the grid of examples/Example02.cxx:178 (300 x 300 two-tap filters)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wavenet_amd import Context, generators as G

B, R, C, Ndiv, rng_ = 10, 64, 64, 300, 1.2
X = G.calorimeter(B, R, C, seed=5)
dX = torch.from_numpy(G.to_device_layout(X)).cuda()
g = (2 * np.arange(Ndiv) / (Ndiv - 1) - 1) * rng_
filters = np.stack([np.tile(g, Ndiv), np.repeat(g, Ndiv)], axis=1)   # (a1 = grid[j], a2 = grid[i]), i major
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
dS = torch.empty(len(filters) * B, dtype=torch.float64, device="cuda")
ctx.sparse_cost_grid(dX, B, R, C, filters[:100], dS)
t0 = time.perf_counter()
ctx.sparse_cost_grid(dX, B, R, C, filters, dS)
dt = time.perf_counter() - t0
print(f"cost grid {Ndiv}x{Ndiv} filters x {B} examples of {R}x{C}: {dt*1e3:.1f} ms ({len(filters)*B/dt/1e6:.2f} M forward+Gini / s)")
ctx.set_filter(filters[0])
dY = torch.empty(B * R * C, dtype=torch.float64, device="cuda"); dS1 = torch.empty(B, dtype=torch.float64, device="cuda")
t0 = time.perf_counter()
for f in range(2000):
    ctx.set_filter(filters[f]); ctx.forward(dX, B, R, C, dY); ctx.sparse_term(dY, B, R * C, dS1); dS1.cpu()
dt1 = (time.perf_counter() - t0) / 2000 * len(filters)
print(f"point by point (set_filter, forward, sparse_term, download): {dt1*1e3:.0f} ms extrapolated")
