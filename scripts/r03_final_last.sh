#!/bin/bash
# last measurement set of round 3 (final build): bench lines, spp sweep, small-N table, profiles
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
python bench.py --steps 10 --warmup 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --workload cfg3 --steps 5 --warmup 1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
python bench.py --workload cfg5 --steps 2 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
bash scripts/sweep.sh > $O/spp_sweep.txt 2>&1; cat $O/spp_sweep.txt
(SPP=8 python scripts/smalln.py; SPP=16 python scripts/smalln.py; SPP=32 ROWS=540 python scripts/smalln.py; SPP=64 ROWS=270 python scripts/smalln.py) 2>&1 | grep -v amdgpu > $O/smalln.txt; grep -c kernel_ms $O/smalln.txt
BENCH_FLAGS="--no-scaling-4k32 --width 3840 --rows-per-gpu 270 --spp 32" bash scripts/profile.sh r03_4k32slab > $O/prof_4k32slab.log 2>&1 || { cat $O/prof_4k32slab.log; exit 3; }; tail -2 $O/prof_4k32slab.log
BENCH_FLAGS="--no-scaling-4k32 --spp 16" bash scripts/profile.sh r03_16spp > $O/prof_16spp.log 2>&1 || { cat $O/prof_16spp.log; exit 3; }; tail -2 $O/prof_16spp.log
