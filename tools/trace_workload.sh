#!/bin/bash
# rocprofv3 --kernel-trace --stats over one bench workload, on the GPU box:
#   tools/trace_workload.sh <workload> <tag>
# leaves gpurun_out/trace/<tag>_bench_<workload>_kernel_stats.csv and the bench line beside it.
set -u
W=$1
TAG=$2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$W
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$W -o r -- python $ROOT/bench.py --workload $W --no-cpu-baseline > $OUT/${TAG}_bench_${W}_traced.json 2> $OUT/${TAG}_${W}.err || { echo "trace of $W failed"; tail -5 $OUT/${TAG}_${W}.err; exit 1; }
cp $(find /tmp/trace_$W -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_${W}_kernel_stats.csv
# the last step's dispatches, one row each (name, start, end)
python3 - "$(find /tmp/trace_$W -name "*kernel_trace.csv" | head -1)" $OUT/${TAG}_bench_${W}_last_step.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-80:]
t0 = int(tail[0]["Start_Timestamp"])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "workgroup", "start_us", "duration_us"])
    for r in tail:
        name = r["Kernel_Name"]
        cut = name.find("(")
        w.writerow([name if cut < 0 else name[:cut], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")),
                    "%.1f" % ((int(r["Start_Timestamp"]) - t0) / 1e3), "%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])
PY
echo "trace of $W done"
