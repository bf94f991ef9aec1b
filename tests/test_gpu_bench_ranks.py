"""The multi-rank path of bench.py on the one GPU a test box has: two ranks under torch.distributed.run, both on
device 0, the sums of every batch exchanged by gloo (--rehearse).  Everything but RCCL itself is the code the
driver's 2-, 4- and 8-GPU runs execute: rank environment, contiguous shares of one sequence of examples, the
barriers and the maximum over ranks around the timed region, train_batch -> all-reduce -> apply_update on every
rank, one JSON line from rank 0.  The learned filter must be the one a single rank learns from the whole batch
(shares are added in another order: agreement to rounding, not to the bit)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.join(os.path.dirname(__file__), "..")


def _line(cmd, env=None):
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert res.returncode == 0, (res.stdout[-1000:], res.stderr[-3000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_on_one_gpu_learn_what_one_rank_learns(scaling):
    common = ["--steps", "4", "--warmup", "2", "--workload", "headline", "--batch", "512", "--no-cpu-baseline",
              "--no-exchange-probe", "--no-coach-probe", "--scaling", scaling]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", "29655", "bench.py", "--gpus", "2", "--rehearse"] + common, env=env)
    assert two["n_gpus"] == 2 and two["scaling"] == scaling and "rehearsal" in two
    global_batch = two["config"]["global_batch"]
    assert global_batch == (1024 if scaling == "weak" else 512)
    one = _line([sys.executable, "bench.py", "--gpus", "1"] + common[:common.index("--batch")] +
                ["--batch", str(global_batch)] + common[common.index("--batch") + 2:-2] + ["--scaling", "weak"])
    assert one["config"]["global_batch"] == global_batch
    f2, f1 = np.array(two["final_filter"]), np.array(one["final_filter"])
    assert np.all(np.isfinite(f2)) and two["nonfinite"] == 0
    assert np.allclose(f2, f1, rtol=1e-12, atol=1e-15), (f2, f1)
    assert two["value"] > 0 and two["ms_per_step"] > 0
