#!/bin/bash
# Per-kernel average times of one bench workload for every library variant under wavenet_amd/lib/exp
# (rocprofv3 --kernel-trace --stats), on the GPU box:
#   [WORKLOAD=2d512] [PATTERN='kr_|kc_'] [EXTRA='--taps 4'] tools/ab_trace.sh
set -u
ROOT=$(pwd)
WORKLOAD=${WORKLOAD:-2d512}
PATTERN=${PATTERN:-'kr_|kc_|kb_'}
OUT=$ROOT/gpurun_out/ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in $ROOT/wavenet_amd/lib/exp/*.so; do
  name=$(basename $so .so)
  rm -rf /tmp/ab_$name
  WAVENET_HIP_LIBRARY=$so timeout -k 10 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$name -o r -- python $ROOT/bench.py --workload $WORKLOAD --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline ${EXTRA:-} > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; exit 1; }
  python3 - "$(find /tmp/ab_$name -name '*kernel_stats.csv' | head -1)" "$name" "$PATTERN" "$OUT/$name.json" <<'PY' | tee -a $OUT/summary.txt
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[3])
line = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
parts = []
for r in rows:
    n = r["Name"].replace("void ", "")
    n = n.replace("(anonymous namespace)::", ""); n = n[:n.find("(")] if "(" in n else n
    n = n.replace("wn::", "").replace("rk::", "").replace("(anonymous namespace)::", "")
    if pat.search(n):
        parts.append("%s %.0f" % (n, float(r["AverageNs"]) / 1e3))
print("%-14s step %.3f ms | %s" % (sys.argv[2], line["roofline"]["kernel_ms"], " | ".join(parts)))
PY
done
