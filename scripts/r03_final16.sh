#!/bin/bash
# re-measure what the last K = 13 changes touch: cfg3 bench line, spp sweep, small-N table, 16-spp profile
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
python bench.py --workload cfg3 --steps 5 --warmup 1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
bash scripts/sweep.sh > $O/spp_sweep.txt 2>&1; cat $O/spp_sweep.txt
(SPP=8 python scripts/smalln.py; SPP=16 python scripts/smalln.py; SPP=32 ROWS=540 python scripts/smalln.py; SPP=64 ROWS=270 python scripts/smalln.py) 2>&1 | grep -v amdgpu > $O/smalln.txt; grep -c kernel_ms $O/smalln.txt
BENCH_FLAGS="--no-scaling-4k32 --spp 16" bash scripts/profile.sh r03_16spp > $O/prof_16spp.log 2>&1 || { cat $O/prof_16spp.log; exit 3; }; tail -3 $O/prof_16spp.log
