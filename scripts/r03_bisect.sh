#!/bin/bash
# (historical: the variant libraries were builds with bisecting macros -- build.py --variant <tag> -D... -- that the source no
# longer carries; the fault they chased was the missing scc clobber of lds_store_u64_if)
# determinism of the 32-spp split route on the shipped build and on variant libraries (bisecting a run-to-run difference)
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r3j}
mkdir -p $O
echo "== main"; ROWS=${ROWS:-48} python scripts/determinism.py 2>&1 | grep -v amdgpu | tee $O/det_main.txt
for v in $VARIANTS; do echo "== $v"; ROWS=${ROWS:-48} RPF_HIP_LIB=$PWD/raytracer-rpf_amd/lib/librpf_hip_$v.so python scripts/determinism.py 2>&1 | grep -v amdgpu | tee $O/det_$v.txt; done
