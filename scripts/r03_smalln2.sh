#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r03e
python -m pytest tests -m gpu -q > gpurun_out/r03e/tests.log 2>&1; tail -3 gpurun_out/r03e/tests.log
SPP=8 python scripts/smalln.py > gpurun_out/r03e/smalln_8.txt 2>gpurun_out/r03e/smalln_8.err; cat gpurun_out/r03e/smalln_8.txt
SPP=16 CASE=2 python scripts/smalln.py > gpurun_out/r03e/smalln_16.txt 2>gpurun_out/r03e/smalln_16.err; cat gpurun_out/r03e/smalln_16.txt
for c in 0 2; do
  CASE=$c PACKED=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03e/trace_$c -- python3 scripts/smalln.py > gpurun_out/r03e/trace_$c.log 2>&1
  f=$(find gpurun_out/r03e/trace_$c -name "*kernel_stats.csv" | head -1)
  echo "== case $c"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rpf" in r["Name"]: print("%-90s calls %4s avg_us %10.1f total_ms %9.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-4k32 > gpurun_out/r03e/bench_cfg2.json 2> gpurun_out/r03e/bench_cfg2.err; python -c "
import json; d=json.load(open('gpurun_out/r03e/bench_cfg2.json')); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
