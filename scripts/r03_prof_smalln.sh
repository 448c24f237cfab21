#!/bin/bash
# per-kernel times of the small-N route (packed) at 1080p x 8 spp
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
for c in 0 2; do
  CASE=$c PACKED=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03c/trace_$c -- python3 scripts/smalln.py > gpurun_out/r03c/trace_$c.log 2>&1
  f=$(find gpurun_out/r03c/trace_$c -name "*kernel_stats.csv" | head -1)
  echo "== case $c"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-90s calls %4s avg_us %10.1f total_ms %9.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
