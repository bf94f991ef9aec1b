"""Ranks implied by wn_sparse_term_deriv against numpy, for one shape (debugging aid)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wavenet_amd import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(n)
Y = rng.standard_normal((B, n))
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
dY = torch.from_numpy(Y.reshape(-1)).cuda()
dD = torch.empty(B * n, dtype=torch.float64, device="cuda")
dS = torch.empty(B, dtype=torch.float64, device="cuda")
ctx.sparse_term_deriv(dY, B, n, dD, dS)
torch.cuda.synchronize()
D = dD.cpu().numpy().reshape(B, n)
S = dS.cpu().numpy()
for e in range(B):
    a = np.abs(Y[e])
    order = np.argsort(a, kind="stable")
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(1, n + 1)
    g = n * a.sum()
    f = ((2 * np.arange(1, n + 1) - n - 1) * a[order]).sum()
    Sref = 1 - f / g
    # delta = -((2 rank - n - 1) g - f n) sign / g^2  ->  implied rank from D using the reference f, g
    sg = np.sign(Y[e])
    implied = ((-D[e] * g * g / sg + f * n) / g + n + 1) / 2
    bad = np.nonzero(np.abs(implied - rank) > 0.25)[0]
    print(f"example {e}: S {S[e]:.6f} ref {Sref:.6f}; wrong ranks {len(bad)} of {n}")
    for i in bad[:12]:
        print("   idx", i, "rank", rank[i], "implied", round(float(implied[i]), 2), "|y|", a[i])
ctx.close()
