#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of one bench workload: bash scripts/r03_trace.sh <tag> <bench flags...>
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/trace_$TAG
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-multi-inprocess --no-scaling-4k32 "$@" > $O/bench.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rpf" in r["Name"]: print("%-100s calls %4s avg_us %10.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
