#!/bin/bash
# round 3, second session: the DPP own-row form of stage 4 -- probe, tests, spp sweep on the shipped build and on a variant library
export TMPDIR=/tmp
O=gpurun_out/r3g
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_bcast_probe scripts/microbench/dpp_bcast_probe.hip > /dev/null 2>&1 && /tmp/dpp_bcast_probe > $O/dpp_probe.txt; tail -1 $O/dpp_probe.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || exit $rc
bash scripts/sweep.sh > $O/sweep_main.txt 2>&1; cat $O/sweep_main.txt
for v in $VARIANTS; do
  RPF_HIP_LIB=$PWD/raytracer-rpf_amd/lib/librpf_hip_$v.so bash scripts/sweep.sh > $O/sweep_$v.txt 2>&1; echo "== variant $v"; cat $O/sweep_$v.txt
done
