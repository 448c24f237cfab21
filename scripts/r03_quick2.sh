#!/bin/bash
# quick loop of the second round-3 session: GPU tests (optionally a -k subset), then the spp sweep of the shipped build and of variants
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r3h}
mkdir -p $O
if [ -n "$TESTK" ]; then timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "$TESTK" > $O/tests.log 2>&1; else timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; fi
rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || exit $rc
SWEEP_FLAGS="$SWEEP_FLAGS" bash scripts/sweep.sh > $O/sweep_main.txt 2>&1; cat $O/sweep_main.txt
for v in $VARIANTS; do
  RPF_HIP_LIB=$PWD/raytracer-rpf_amd/lib/librpf_hip_$v.so bash scripts/sweep.sh > $O/sweep_$v.txt 2>&1; echo "== variant $v"; cat $O/sweep_$v.txt
done
