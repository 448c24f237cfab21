#!/bin/bash
# per-kernel traces of the four small-N generators at 1080p x 8 spp (default route)
export TMPDIR=/tmp
O=gpurun_out/r03sn
mkdir -p $O
for c in 0 1 2 3; do
  CASE=$c PACKED=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -- python3 scripts/smalln.py > $O/trace_$c.log 2>&1 || { echo "case $c failed"; exit 3; }
  echo "case $c done"
done
