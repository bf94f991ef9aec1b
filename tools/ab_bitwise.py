"""Are two builds of the library bit-for-bit the same on the training step?  On the GPU box:

    python tools/ab_bitwise.py wavenet_amd/lib/exp/a.so wavenet_amd/lib/exp/b.so "1024x1:6,16,30,32" "64x64:4,6,16"

Every shape is R x C : filter lengths; each build runs in a process of its own (the library is chosen by
WAVENET_HIP_LIBRARY when the package is imported) and leaves the batch sums, the per-example gradients and
sparsity terms of a batch of 5; the parent compares the bytes.
"""
import hashlib
import json
import os
import subprocess
import sys


def child(lib, out, shapes):
    os.environ["WAVENET_HIP_LIBRARY"] = os.path.abspath(lib)
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import numpy as np
    import torch
    from wavenet_amd import Context, generators as G
    import filters as F
    res = {}
    for spec in shapes:
        shape, taps = spec.split(":")
        R, C = (int(v) for v in shape.split("x"))
        for N in (int(v) for v in taps.split(",")):
            ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
            ctx.set_filter(F.random_filter(N, 2))
            B = 5
            X = G.calorimeter(B, R, C, seed=3) if C > 1 else G.bumps(B, R, seed=3)
            dX = torch.from_numpy(np.ascontiguousarray(X).reshape(-1)).cuda()
            dG = torch.empty(B * N, dtype=torch.float64, device="cuda")
            dS = torch.empty(B, dtype=torch.float64, device="cuda")
            ctx.train_batch(dX, B, R, C, dG, dS)
            sums = np.asarray(ctx.sums())
            torch.cuda.synchronize()
            h = hashlib.sha256(sums.tobytes() + dG.cpu().numpy().tobytes() + dS.cpu().numpy().tobytes()).hexdigest()
            res["%s N=%d" % (shape, N)] = h
            ctx.close()
    json.dump(res, open(out, "w"))


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3], sys.argv[4:])
    libs, shapes = sys.argv[1:3], sys.argv[3:]
    os.makedirs("gpurun_out", exist_ok=True)
    outs = []
    for i, lib in enumerate(libs):
        out = "gpurun_out/ab_bitwise_%d.json" % i
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, out] + shapes, check=True, timeout=300)
        outs.append(json.load(open(out)))
    same = True
    for k in outs[0]:
        ok = outs[0][k] == outs[1][k]
        same &= ok
        print(k, "identical" if ok else "DIFFERENT")
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
