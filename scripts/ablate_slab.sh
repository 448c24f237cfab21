#!/bin/bash
# stage ablation on one rank's 4K x 32 spp slab (and 64 spp): option stage_mask (results wrong unless -1)
for spp in 32 64; do
  rows=$((8640/spp))
  for m in -1 0 1 3 7; do
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --allow-nonfinite --spp $spp --width 3840 --rows-per-gpu $rows --option stage_mask=$m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spp $spp rows $rows stage_mask $m', 'kernel_ms %.1f'%d['roofline']['kernel_ms'])"
  done
done
