#!/bin/bash
# (historical: the variant libraries were builds with bisecting macros -- build.py --variant <tag> -D... -- that the source no
# longer carries; the fault they chased was the missing scc clobber of lds_store_u64_if)
# the V2 form of mi_group in the one-wave K = 25 kernel (option waves_per_pixel 1), variant libraries: which ingredient breaks it
export TMPDIR=/tmp
O=gpurun_out/r3v
mkdir -p $O
for v in $VARIANTS; do echo "== $v"; RPF_HIP_LIB=$PWD/raytracer-rpf_amd/lib/librpf_hip_$v.so NW=1 python scripts/r03_diag_mi.py 2>&1 | grep -v amdgpu | head -1 | tee -a $O/diag.txt; done
