#!/bin/bash
# strip width of the XCD pixel walk vs kernel time and fetched bytes on one rank's 4K x 32 spp slab (option strip_w)
export TMPDIR=/tmp
OUT=gpurun_out/strip_sweep_${SPP:-32}; rm -rf $OUT; mkdir -p $OUT
for w in ${STRIPS:-16 24 40 64 128}; do
  CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --spp ${SPP:-32} --width ${WIDTH:-3840} --rows-per-gpu ${ROWS:-270} --option strip_w=$w"
  ms=$($CMD 2>/dev/null | python3 -c "import sys,json; print('%.1f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/w$w -- $CMD > $OUT/w$w.log 2>&1
  f=$(python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("$OUT/w$w/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "filter_pixel_kernel<${KSEL:-25}" in r["Kernel_Name"]]
print("%.3g" % (sum(v)/max(len(v),1)*2048/1e9))
PY
)
  echo "strip_w $w kernel_ms $ms fetched_GB(x2-corrected) $f"
done
