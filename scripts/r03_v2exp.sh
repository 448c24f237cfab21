#!/bin/bash
# experiment: mi_group's V2 form in the one-wave kernels (variant library v2all) -- MI against the oracle with one wave per pixel,
# run-to-run determinism of a 32-spp slab, 16-spp timing against the shipped build
export TMPDIR=/tmp
O=gpurun_out/r3p
mkdir -p $O
L=$PWD/raytracer-rpf_amd/lib/librpf_hip_v2all.so
RPF_HIP_LIB=$L NW=1 python scripts/r03_diag_mi.py 2>&1 | grep -v amdgpu | head -3 | tee $O/diag_nw1_v2all.txt
RPF_HIP_LIB=$L NW=1 SPP=64 python scripts/r03_diag_mi.py 2>&1 | grep -v amdgpu | head -1 | tee -a $O/diag_nw1_v2all.txt
RPF_HIP_LIB=$L NW=0 SPP=16 python scripts/r03_diag_mi.py 2>&1 | grep -v amdgpu | head -1 | tee -a $O/diag_nw1_v2all.txt
RPF_HIP_LIB=$L ROWS=48 python scripts/determinism.py 2>&1 | grep -v amdgpu | grep "run 0" | tee $O/det_v2all.txt
for lib in "" $L; do
  RPF_HIP_LIB=$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-multi-inprocess --no-scaling-4k32 --spp 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('16 spp', d['config']['library'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'])" | tee -a $O/spp16.txt
done
