#!/bin/bash
# quick loop: packed / small-N tests, then the small-N timing (8 spp) with per-kernel trace of case 0 and 2
export TMPDIR=/tmp
mkdir -p gpurun_out/r03q
python -m pytest tests -m gpu -q -x -k "packed or flat_quad or config1 or small_neigh or independent" > gpurun_out/r03q/tests.log 2>&1; tail -3 gpurun_out/r03q/tests.log
for c in ${CASES:-0 2 3}; do
  CASE=$c PACKED=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03q/trace_$c -- python3 scripts/smalln.py > gpurun_out/r03q/trace_$c.log 2>&1
  grep kernel_ms gpurun_out/r03q/trace_$c.log | cut -c1-420
  f=$(find gpurun_out/r03q/trace_$c -name "*kernel_stats.csv" | head -1)
  echo "== case $c"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rpf" in r["Name"]: print("%-90s calls %4s avg_us %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
