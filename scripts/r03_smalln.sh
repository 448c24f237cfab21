#!/bin/bash
mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -q > gpurun_out/r03b/tests.log 2>&1; tail -3 gpurun_out/r03b/tests.log
SPP=8 python scripts/smalln.py > gpurun_out/r03b/smalln_8.txt 2>gpurun_out/r03b/smalln_8.err; cat gpurun_out/r03b/smalln_8.txt; tail -3 gpurun_out/r03b/smalln_8.err
SPP=16 python scripts/smalln.py > gpurun_out/r03b/smalln_16.txt 2>gpurun_out/r03b/smalln_16.err; cat gpurun_out/r03b/smalln_16.txt
SPP=32 ROWS=540 python scripts/smalln.py > gpurun_out/r03b/smalln_32.txt 2>gpurun_out/r03b/smalln_32.err; cat gpurun_out/r03b/smalln_32.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-4k32 > gpurun_out/r03b/bench_cfg2.json 2> gpurun_out/r03b/bench_cfg2.err; python -c "
import json; d=json.load(open('gpurun_out/r03b/bench_cfg2.json')); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
