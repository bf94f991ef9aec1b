"""Training-step throughput for arbitrary image shapes (calorimeter inputs generated on the device).
    python tools/shape_bench.py R C taps batch [steps]
WAVENET_HIP_WAVE64=0 switches the one-wave-per-example / per-line kernels off for comparison."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wavenet_amd import Context  # noqa: E402
from wavenet_amd.filters import daubechies, perturbed  # noqa: E402

R, C, N, B = (int(v) for v in sys.argv[1:5])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
dX = torch.empty(B * R * C, dtype=torch.float64, device="cuda")
for e0 in range(0, B, 32768):
    ctx.generate_calorimeter(dX[e0 * R * C:], min(32768, B - e0), R, C, seed=1, first=e0)
ctx.set_filter(perturbed(daubechies(N), 1e-2, seed=7))
for _ in range(2):
    ctx.train_batch(dX, B, R, C)
    ctx.apply_update(B, B * 10.0, 1e-3, 0.0, True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    ctx.train_batch(dX, B, R, C)
    ctx.apply_update(B, B * 10.0, 1e-3, 0.0, True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print(f"{R}x{C} N={N} batch {B}: {dt * 1e3:.3f} ms per step, {B / dt:.0f} examples/s, "
      f"{B * (120 if (R > 1 and C > 1) else 64) * R * C / dt / 8e12:.3f} of the roofline", flush=True)
ctx.close()
