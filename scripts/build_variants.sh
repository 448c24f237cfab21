#!/bin/bash
# profiling variants of librpf_hip.so (same ABI): gpurun_out is scratch, so they are built into build/variants/
# usage: bash scripts/build_variants.sh name1:"-DFLAGS" name2:"-DFLAGS" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags \
      -I$ROOT/include -I$ROOT/raytracer-rpf_amd/csrc -o $OUT/librpf_hip_$name.so \
      $ROOT/raytracer-rpf_amd/csrc/rpf_kernels.hip $ROOT/raytracer-rpf_amd/csrc/rpf_api.hip && echo built $name ) &
done
wait
