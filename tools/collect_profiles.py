"""Copies the summaries of tools/measure_round.sh from gpurun_out/measure/ into profiles/ (file names
prefixed with the tag given) and recomputes profiles/traffic.json: HBM-side bytes per launch of the training kernel from the
FETCH_SIZE and WRITE_SIZE passes (the counters are in KB; raw figures, see the comment in the file).

    python tools/collect_profiles.py r01
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "measure")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
KERNEL = "k_train64"


def per_launch(path, counter):
    total, launches = 0.0, 0
    with open(path) as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                total += float(row["Counter_Value"])
                launches += 1
    return total / launches if launches else None, launches


def keep_kernel_rows(src, dst):
    """The counter files hold a row per dispatch of every kernel; keep the training kernel's."""
    with open(src) as f, open(dst, "w", newline="") as g:
        r = csv.reader(f)
        w = csv.writer(g)
        head = next(r)
        w.writerow(head)
        name = head.index("Kernel_Name")
        for row in r:
            if KERNEL in row[name] or "k_update" in row[name] or "k_reduce_partials" in row[name]:
                w.writerow(row)


names = {"pmc1.csv": "pmc_fetch.csv", "pmc2.csv": "pmc_write.csv", "pmc3.csv": "pmc_sq1.csv", "pmc4.csv": "pmc_sq2.csv"}
for src, dst in names.items():
    keep_kernel_rows(os.path.join(SRC, src), os.path.join(DST, f"{tag}_{dst}"))
for name in ("bench_2d64_n4.json", "bench_2d64_n4_kernel_stats.csv", "bench_1d1024.json", "bench_2d512.json", "bench_1d1m.json"):
    shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{tag}_{name}"))

fetch, n1 = per_launch(os.path.join(SRC, "pmc1.csv"), "FETCH_SIZE")
write, n2 = per_launch(os.path.join(SRC, "pmc2.csv"), "WRITE_SIZE")
path = os.path.join(DST, "traffic.json")
with open(path) as f:
    traffic = json.load(f)
traffic["2d64_n4"] = int(round((fetch + write) * 1024))
with open(path, "w") as f:
    json.dump(traffic, f, indent=1)
print(f"FETCH_SIZE {fetch * 1024 / 1e6:.1f} MB over {n1} launches, WRITE_SIZE {write * 1024 / 1e6:.1f} MB over {n2}; "
      f"traffic {traffic['2d64_n4'] / 1e6:.1f} MB per launch")
