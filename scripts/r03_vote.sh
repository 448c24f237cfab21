#!/bin/bash
# the exponent vote of the one-wave kernels: spp sweep with the vote (shipped), with its verdict dropped (option screen 0) and compiled out
export TMPDIR=/tmp
O=gpurun_out/r3r
mkdir -p $O
echo "== shipped"; bash scripts/sweep.sh 2>&1 | tee $O/sweep_main.txt
echo "== option screen=0"; SWEEP_FLAGS="--option screen=0" bash scripts/sweep.sh 2>&1 | head -2 | tee $O/sweep_screen0.txt
echo "== compiled without the vote"; RPF_HIP_LIB=$PWD/raytracer-rpf_amd/lib/librpf_hip_novote.so bash scripts/sweep.sh 2>&1 | head -2 | tee $O/sweep_novote.txt
