#!/bin/bash
# round-2 measurement set on the GPU box: default bench line (cfg2 + scaling_4k32), cfg5 line, spp sweep
mkdir -p gpurun_out/r02
python bench.py --steps 10 --warmup 2 > gpurun_out/r02/bench_cfg2.json 2> gpurun_out/r02/bench_cfg2.err && echo "cfg2 done"
python bench.py --workload cfg5 --steps 2 --warmup 1 > gpurun_out/r02/bench_cfg5.json 2> gpurun_out/r02/bench_cfg5.err && echo "cfg5 done"
for cfg in "8 1080" "16 1080" "32 540" "64 270"; do set -- $cfg
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --spp $1 --rows-per-gpu $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('spp', $1, 'rows', $2, 'Msamples/s %.1f'%d['value'], 'kernel_ms %.1f'%d['roofline']['kernel_ms'], 'meanN %.0f'%c['mean_nbhd'], 'maxN', c['max_nbhd'], 'bad', c['nonfinite_pixels'])"
done > gpurun_out/r02/spp_sweep.txt
cat gpurun_out/r02/spp_sweep.txt
python -c "
import json
for f in ('cfg2','cfg5'):
    d=json.load(open('gpurun_out/r02/bench_%s.json'%f))
    print(f, 'value %.1f'%d['value'], 'ms/step %.2f'%d['ms_per_step'], 'kernel_ms %.2f'%d['roofline']['kernel_ms'], 'frac %.5f'%d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('gpu_vs_oracle_rel_l2'))
    if 'scaling_4k32' in d: print('   4k32', d['scaling_4k32'])
"
