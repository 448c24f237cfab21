#!/bin/bash
# review item 8(e): the split route's three launches chunk by chunk (option split_chunk) on one rank's 3840x270x32 slab:
# pass time and fetched bytes per chunk size (0 = whole list)
export TMPDIR=/tmp
O=gpurun_out/r03sc
mkdir -p $O
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "split_weight" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
F="--width 3840 --rows-per-gpu 270 --spp 32 --no-scaling-4k32 --no-cpu-baseline --no-multi-inprocess"
for c in ${CHUNKS:-0 16384 32768 65536 131072 262144}; do
  timeout -k 10 200 python bench.py $F --steps 4 --warmup 1 --option split_chunk=$c 2> $O/bench_$c.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('split_chunk $c: Msamples/s %.1f  ms/step %.2f  kernel_ms %.2f launches %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel_launches_per_step')))"
done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_0 -- python3 bench.py $F --steps 2 --warmup 1 > $O/write_0.log 2>&1 || { echo "pmc write pass failed"; exit 3; }
python3 - $O/write_0 <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_pixel_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(acc): print("  WRITE_SIZE per launch  %-42s %.3f GB" % (k, acc[k] * 1024 / 1e9 / n[k]))
PY
for c in ${FETCH_CHUNKS:-0 65536}; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$c -- python3 bench.py $F --steps 2 --warmup 1 --option split_chunk=$c > $O/fetch_$c.log 2>&1 || { echo "pmc pass $c failed"; tail -3 $O/fetch_$c.log; exit 3; }
  python3 - $O/fetch_$c $c <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_pixel_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k] += float(r["Counter_Value"]); n[k] += 1
tot = 0.0
for k in sorted(acc):
    print("  split_chunk %s  %-42s dispatches %4d  FETCH_SIZE sum %.3f GB (KiB -> bytes, x2: gfx950 tallies 128-B requests at 64 B)" % (sys.argv[2], k, n[k], acc[k] * 2048 / 1e9))
    tot += acc[k] * 2048 / 1e9
print("  split_chunk %s total over 3 passes (1 warm-up + 2 steps): %.2f GB -> per pass %.2f GB" % (sys.argv[2], tot, tot / 3))
PY
done
