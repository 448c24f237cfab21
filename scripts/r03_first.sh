#!/bin/bash
# round-3 first measurement set: suite, the default bench line with its new legs, cfg1 / cfg3, binned route on the headline
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -q > gpurun_out/r03a/tests.log 2>&1; tail -3 gpurun_out/r03a/tests.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r03a/bench_cfg2.json 2> gpurun_out/r03a/bench_cfg2.err; echo "cfg2 rc=$?"; tail -c 600 gpurun_out/r03a/bench_cfg2.err
python bench.py --workload cfg1 --steps 20 --warmup 3 > gpurun_out/r03a/bench_cfg1.json 2> gpurun_out/r03a/bench_cfg1.err; echo "cfg1 rc=$?"; tail -c 600 gpurun_out/r03a/bench_cfg1.err
python bench.py --workload cfg3 --steps 3 --warmup 1 > gpurun_out/r03a/bench_cfg3.json 2> gpurun_out/r03a/bench_cfg3.err; echo "cfg3 rc=$?"; tail -c 600 gpurun_out/r03a/bench_cfg3.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --option binning=1 > gpurun_out/r03a/bench_cfg2_binned.json 2> gpurun_out/r03a/bench_cfg2_binned.err; echo "binned rc=$?"
python bench.py --gpus 2 --dist-backend gloo --steps 2 --warmup 1 --no-scaling-4k32 > gpurun_out/r03a/bench_spawn2.json 2> gpurun_out/r03a/bench_spawn2.err; echo "spawn2 rc=$?"; tail -c 400 gpurun_out/r03a/bench_spawn2.err
python - <<'PY'
import json
for f in ('cfg2','cfg1','cfg3','cfg2_binned','spawn2'):
    try:
        d=json.load(open('gpurun_out/r03a/bench_%s.json'%f))
    except Exception as e:
        print(f, 'no json', e); continue
    print(f, 'value %.1f'%d['value'], 'ms/step %.2f'%d['ms_per_step'], 'kernel_ms %.2f'%d['roofline']['kernel_ms'], 'frac %.5f'%d['roofline']['frac'], 'n_gpus', d['n_gpus'], d['config'].get('nbhd'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('gpu_vs_oracle_rel_l2'))
    for k in ('scaling_4k32','multi_inprocess','parity_probe','parity_probe_32spp'):
        if k in d: print('   ', k, d[k])
PY
