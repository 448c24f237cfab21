#!/bin/bash
# kernel time of the profiling variants built by scripts/build_variants.sh (headline workload), then the stage ablation
# of the product library.  Output: one line per run.
one() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --allow-nonfinite "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms %.2f'%d['roofline']['kernel_ms'], 'ms_per_step %.2f'%d['ms_per_step'])"; }
for v in build/variants/librpf_hip_*.so; do
  n=$(basename $v .so); n=${n#librpf_hip_}
  RPF_HIP_LIB=$PWD/$v one "variant $n"
done
for m in -1 0 1 3 7; do one "stage_mask $m" --option stage_mask=$m; done
one "table_in_lds 1" --option table_in_lds=1
