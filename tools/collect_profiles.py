"""Copies the summaries of tools/measure_round.sh from gpurun_out/ into profiles/ (file names prefixed with the
tag given) and recomputes profiles/traffic.json: HBM-side bytes per training step of every bench workload from
the FETCH_SIZE and WRITE_SIZE passes (counters in KB; raw figures, see the comment in the file).

    python tools/collect_profiles.py r02
"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
WORKLOADS = ["2d64", "2d64n16", "1d1024", "2d512", "2d512n16", "1d1m", "1d1mb64"]
PMC_STEPS = 6          # tools/pmc_workload.sh runs bench.py with 5 timed steps and 1 warm-up step
# kernels that are not part of a training step (input generation, torch's fills, copies of the set-up)
NOT_STEP = ("kg_", "at::native", "k_reset_steps")


# kernels whose reads are 16-byte-per-lane streaming loads: FETCH_SIZE reports half of their bytes on gfx950
# (checked here against their known input volumes: kr_fwd_rows, kc_fwd_cols and kr_hist each read the 4295 MB of
# a 512x512 chunk once and report 2147 - 2212 MB)
WIDE_READS = ("kr_fwd_rows", "kr_grad_rows", "kc_fwd_cols", "kc_bwd_cols", "kr_hist", "kr_part", "kr_count", "kr_max", "k_train1d1024")


def step_bytes(path, counter):
    """Sum of the counter over the dispatches of the training steps, per step, in bytes (the counter is in KB)."""
    total = 0.0
    per_kernel = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if row["Counter_Name"] != counter or any(s in name for s in NOT_STEP):
                continue
            v = float(row["Counter_Value"]) * 1024
            if counter == "FETCH_SIZE" and any(k in name for k in WIDE_READS):
                v *= 2
            total += v
            short = name.replace("void ", "").split("(")[0]
            per_kernel[short] = per_kernel.get(short, 0.0) + v
    return total / PMC_STEPS, {k: v / PMC_STEPS for k, v in per_kernel.items()}


shutil.copy(os.path.join(GO, "measure", f"{tag}_bench_all.jsonl"), os.path.join(DST, f"{tag}_bench_all.jsonl"))
traffic = {"_comment": (
    "HBM-side bytes per training step (wn_train_step: every kernel of the step) of each bench workload: rocprofv3 --pmc "
    "FETCH_SIZE and WRITE_SIZE, separate passes, KB x 1024, summed over the step's dispatches "
    f"(profiles/{tag}_pmc_<workload>.csv holds the per-kernel means, uncorrected).  FETCH_SIZE is doubled, as the "
    "guide prescribes for gfx950, for the kernels that read with 16-byte-per-lane streaming loads (the line kernels of the "
    "streamed path, kr_max, kr_hist, kr_part, k_train1d1024: their known input volumes confirm the factor); the other kernels load 8 bytes "
    "per lane, where the raw figure brackets the known volume (k_train64 reads its 134.2 MB batch twice and reports 226 MB), "
    "and are taken raw.  '_detail' gives fetch and write separately.")}
detail = {}
for w in WORKLOADS:
    for suffix in ("kernel_stats.csv", "last_step.csv"):
        src = os.path.join(GO, "trace", f"{tag}_bench_{w}_{suffix}")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(DST, f"{tag}_bench_{w}_{suffix}"))
    parts = sorted(p for p in os.listdir(os.path.join(GO, "pmc")) if p.startswith(f"{tag}_{w}_") and p.endswith(".csv"))
    if not parts:
        continue
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] +
                          [os.path.join(GO, "pmc", p) for p in parts] + ["--out", os.path.join(DST, f"{tag}_pmc_{w}.csv")])
    fpath, wpath = os.path.join(GO, "pmc", f"{tag}_{w}_fetch.csv"), os.path.join(GO, "pmc", f"{tag}_{w}_write.csv")
    if os.path.exists(fpath) and os.path.exists(wpath):
        fetch, fk = step_bytes(fpath, "FETCH_SIZE")
        write, wk = step_bytes(wpath, "WRITE_SIZE")
        traffic[w] = int(round(fetch + write))
        detail[w] = {"fetch": int(round(fetch)), "write": int(round(write))}
        print(f"{w}: fetch {fetch / 1e6:.1f} MB, write {write / 1e6:.1f} MB per step")
traffic["_detail"] = detail
with open(os.path.join(DST, "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
