"""Per-kernel averages of rocprofv3 counter tables (one row per dispatch and counter):

    python tools/pmc_summary.py gpurun_out/pmc/r02_1d1024_*.csv [--out profiles/r02_pmc_1d1024.csv]

Prints kernel, launches, counter, mean value per launch; FETCH_SIZE / WRITE_SIZE (reported in KB) are also
given in bytes.
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


def main(argv):
    out = None
    if "--out" in argv:
        i = argv.index("--out")
        out = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    acc = defaultdict(lambda: [0.0, 0])
    meta = {}
    for path in argv:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = (short(row["Kernel_Name"]), row["Counter_Name"])
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
                meta[k[0]] = (row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"], row["Scratch_Size"],
                              row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"])
    rows = [("kernel", "grid", "workgroup", "lds", "scratch", "vgpr", "agpr", "sgpr", "launches", "counter", "mean_per_launch", "bytes")]
    for (kern, ctr), (total, n) in sorted(acc.items()):
        mean = total / n
        b = int(round(mean * 1024)) if ctr in ("FETCH_SIZE", "WRITE_SIZE") else ""
        rows.append((kern,) + meta[kern] + (n, ctr, f"{mean:.1f}", b))
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
    w.writerows(rows)


if __name__ == "__main__":
    main(sys.argv[1:])
