#!/bin/bash
# 16 spp (K = 13 class): one-wave kernel vs four waves per pixel vs the split route (options), with a same-bits check
export TMPDIR=/tmp
mkdir -p gpurun_out/r03s
python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import rpf_pkg; rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip
W, H, S = 20, 12, 16
planes = fb.synth_planes(W, H, S, seed=33, sigma_f=0.05, sigma_c=1e-3, mode="smooth")
desc = hip.make_desc(W, H, S, policy=hip.DEGEN_EPS)
with hip.Context(0) as c:
    a = c.filter_pass_debug(planes, desc, box=7)
with hip.Context(0) as c:
    c.set_option("waves_per_pixel", 4); c.set_option("split_weights", 1)
    b = c.filter_pass_debug(planes, desc, box=7)
print("max N", a["max_nbhd"], "split K=13 same bits:", all(np.array_equal(a[k], b[k], equal_nan=True) for k in ("colour", "mi", "alpha", "beta", "wrc", "mean", "stddev", "bin_hash", "member_hash", "nbhd_size")))
PY
for opts in "" "--option waves_per_pixel=4" "--option waves_per_pixel=4 --option split_weights=1"; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --spp 16 $opts 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('16 spp [$opts]', 'Msamples/s %.1f'%d['value'], 'kernel_ms %.1f'%d['roofline']['kernel_ms'], 'launches', d['roofline']['kernel_launches_per_step'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03s/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-4k32 --spp 16 --option waves_per_pixel=4 --option split_weights=1 > /dev/null 2>&1
f=$(find gpurun_out/r03s/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rpf" in r["Name"]: print("%-100s calls %4s avg_us %10.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
