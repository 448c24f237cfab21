#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c8
python -m pytest tests -m gpu -q -x -k "split or config4 or config5 or layout27 or far_pair or waves_per" > gpurun_out/r03c8/tests.log 2>&1; tail -3 gpurun_out/r03c8/tests.log
for cw in 4 8; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03c8/trace_$cw -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --width 3840 --rows-per-gpu 270 --spp 32 --option chain_waves=$cw > gpurun_out/r03c8/slab_$cw.json 2> gpurun_out/r03c8/slab_$cw.err
f=$(find gpurun_out/r03c8/trace_$cw -name "*kernel_stats.csv" | head -1)
echo "== chain_waves $cw"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "filter_pixel_kernel" in r["Name"]: print("%-100s calls %4s avg_us %10.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --spp 64 --rows-per-gpu 270 --option chain_waves=$cw 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('64 spp', 'kernel_ms %.1f'%d['roofline']['kernel_ms'], 'Msamples/s %.1f'%d['value'])"
done
