#!/bin/bash
# large-neighbourhood kernels: suite, spp sweep, one rank's 4K x 32 slab with per-kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out/r03l
if [ -z "$NOTESTS" ]; then python -m pytest tests -m gpu -q > gpurun_out/r03l/tests.log 2>&1; tail -3 gpurun_out/r03l/tests.log; fi
bash scripts/sweep.sh > gpurun_out/r03l/sweep.txt 2>&1; cat gpurun_out/r03l/sweep.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03l/trace_slab -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --width 3840 --rows-per-gpu 270 --spp 32 > gpurun_out/r03l/slab.json 2> gpurun_out/r03l/slab.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03l/slab.json')); print('slab 3840x270x32', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
f=$(find gpurun_out/r03l/trace_slab -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rpf" in r["Name"]: print("%-100s calls %4s avg_us %10.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
