#!/bin/bash
# round-3 measurement set on the GPU box (summaries are copied into profiles/ afterwards).  PART selects a subset.
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
part() { [ -z "$PART" ] || [[ " $PART " == *" $1 "* ]]; }
if part bench; then
python bench.py --steps 10 --warmup 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --workload cfg1 --steps 50 --warmup 5 > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "cfg1 rc=$?"
python bench.py --workload cfg3 --steps 5 --warmup 1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
python bench.py --workload cfg5 --steps 2 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
fi
if part sweep; then
bash scripts/sweep.sh > $O/spp_sweep.txt 2>&1; cat $O/spp_sweep.txt
(SPP=8 python scripts/smalln.py; SPP=16 python scripts/smalln.py; SPP=32 ROWS=540 python scripts/smalln.py; SPP=64 ROWS=270 python scripts/smalln.py) 2>&1 | grep -v amdgpu > $O/smalln.txt; grep -c kernel_ms $O/smalln.txt
fi
if part prof; then
bash scripts/profile.sh r03_1080p8 > $O/prof_1080p8.log 2>&1 || { cat $O/prof_1080p8.log; exit 3; }; tail -3 $O/prof_1080p8.log
BENCH_FLAGS="--no-scaling-4k32 --width 3840 --rows-per-gpu 270 --spp 32" bash scripts/profile.sh r03_4k32slab > $O/prof_4k32slab.log 2>&1 || { cat $O/prof_4k32slab.log; exit 3; }; tail -3 $O/prof_4k32slab.log
BENCH_FLAGS="--no-scaling-4k32 --spp 16" bash scripts/profile.sh r03_16spp > $O/prof_16spp.log 2>&1 || { cat $O/prof_16spp.log; exit 3; }; tail -3 $O/prof_16spp.log
BENCH_FLAGS="--workload cfg3" bash scripts/profile.sh r03_cfg3 > $O/prof_cfg3.log 2>&1 || { cat $O/prof_cfg3.log; exit 3; }; tail -3 $O/prof_cfg3.log
fi
if part prof5; then
BENCH_FLAGS="--workload cfg5 --steps 1" bash scripts/profile.sh r03_cfg5 > $O/prof_cfg5.log 2>&1 || { cat $O/prof_cfg5.log; exit 3; }; tail -3 $O/prof_cfg5.log
fi
if part prof5s; then
# (counter passes of the full 8192x512x64 workload hung inside the profiler: a 70-row slab of the same layout instead)
BENCH_FLAGS="--workload cfg5 --rows-per-gpu 70 --steps 1" bash scripts/profile.sh r03_cfg5slab > $O/prof_cfg5slab.log 2>&1 || { cat $O/prof_cfg5slab.log; exit 3; }; tail -3 $O/prof_cfg5slab.log
fi
if part marker; then
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/marker -- python3 bench.py --workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline > $O/marker.log 2>&1
find $O/marker -name "*marker*stats*.csv" -o -name "*domain_stats*.csv" | head; f=$(find $O/marker -name "*marker_api_stats.csv" | head -1); [ -n "$f" ] && cat "$f"
fi
if part fuzz; then
python scripts/fuzz_parity.py 250 7 > $O/fuzz1.txt 2>&1; tail -1 $O/fuzz1.txt
python scripts/fuzz_parity.py 250 11 > $O/fuzz2.txt 2>&1; tail -1 $O/fuzz2.txt
fi
