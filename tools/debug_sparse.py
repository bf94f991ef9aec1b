import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wavenet_amd import Context
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
n = 4096
def run(Y, deriv):
    B = Y.shape[0]
    dY = torch.from_numpy(Y.reshape(-1)).cuda()
    dS = torch.full((B,), -7.0, dtype=torch.float64, device="cuda")
    dD = torch.empty(B * n, dtype=torch.float64, device="cuda")
    if deriv: ctx.sparse_term_deriv(dY, B, n, dD, dS)
    else: ctx.sparse_term(dY, B, n, dS)
    torch.cuda.synchronize()
    return dS.cpu().numpy()
Y = np.zeros((2, n)); Y[0, 1234] = -3.0; Y[1, :] = 0.7
print("both, no deriv", run(Y, False))
print("both, deriv   ", run(Y, True))
print("onehot only, no deriv", run(Y[:1], False))
print("const only, no deriv", run(Y[1:], False))
print("const only, deriv", run(Y[1:], True))
Y2 = np.zeros((1, n)); Y2[0, :] = 0.7; Y2[0, 5] = 0.9
print("const+1, no deriv", run(Y2, False), "expect", 1 - (4095*0.9 + 0.7*(0-4095))/ (4096*(0.7*4095+0.9)))
