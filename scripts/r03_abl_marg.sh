#!/bin/bash
# what the 19 marginal histograms of the large-K MI stage cost: stage_mask 15 skips them (results wrong), slab 3840x270x32 + 16 spp
export TMPDIR=/tmp
mkdir -p gpurun_out/r03m
for m in -1 15; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03m/trace_$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --width 3840 --rows-per-gpu 270 --spp 32 --allow-nonfinite --option stage_mask=$m > gpurun_out/r03m/slab_$m.json 2> gpurun_out/r03m/slab_$m.err
f=$(find gpurun_out/r03m/trace_$m -name "*kernel_stats.csv" | head -1)
echo "== stage_mask $m"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "filter_pixel_kernel" in r["Name"]: print("%-100s calls %4s avg_us %10.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --spp 16 --allow-nonfinite --option stage_mask=$m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('16 spp', 'kernel_ms %.1f'%d['roofline']['kernel_ms'])"
done
