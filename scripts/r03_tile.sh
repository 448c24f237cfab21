#!/bin/bash
# count-first route (two-phase nbhd_count_kernel ahead of the filter kernels) vs the fused route: parity subset, small-N timing
# per route, per-kernel trace, headline A/B
export TMPDIR=/tmp
O=gpurun_out/r03t
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x ${TESTSEL:--k "count_first or packed or flat_quad or config1 or small_neigh or independent or row_slab or layout27_fp16 or multi_pass"} > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for cf in ${ROUTES:--1 0 1}; do
  echo "== COUNT_FIRST $cf"
  for c in ${CASES:-0 1 2 3}; do CASE=$c PACKED=1 COUNT_FIRST=$cf timeout -k 10 200 python3 scripts/smalln.py 2>&1 | grep kernel_ms | cut -c1-60,225-262; done
done
CASE=0 PACKED=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_0 -- python3 scripts/smalln.py > $O/trace_0.log 2>&1
f=$(find $O/trace_0 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rpf" in r["Name"]: print("%-90s calls %4s avg_us %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for cf in ${HROUTES:--1 0}; do
  echo "== headline count_first $cf"
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-scaling-4k32 --no-multi-inprocess --option count_first=$cf 2> $O/bench_$cf.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
