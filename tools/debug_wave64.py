"""Compare the wave kernel, the workgroup kernel and the oracle on tie-heavy Haar inputs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import filters as F
from gpu_util import dev_empty, to_dev
from oracle import oracle
from wavenet_amd import Context

def mk(w):
    os.environ["WAVENET_HIP_WAVE64"] = w
    return Context(0, stream=torch.cuda.current_stream().cuda_stream)
cw, cb = mk("1"), mk("0")
R = C = 64
a = F.HAAR
B = 6
X = np.zeros((B, R, C))
rng = np.random.default_rng(0)
for e in range(B):
    for _ in range(1 + e):
        X[e, rng.integers(R), rng.integers(C)] = 1.0
X[B - 1, :32, :16] = 2.0
def run(c):
    c.set_filter(a)
    dG, dS = dev_empty(B * 2), dev_empty(B)
    c.train_batch(to_dev(X), B, R, C, dG, dS)
    c.sums()
    return dG.cpu().numpy().reshape(B, 2), dS.cpu().numpy()
G, S = run(cw); Gb, Sb = run(cb)
for e in range(B):
    ge, se, _ = oracle.band_train_batch(a, X[e:e+1])
    print(e, 'S', S[e], Sb[e], se, 'G', G[e], Gb[e], ge)
