#!/bin/bash
# PMC counters of the packed kernels (small-N route), case = flat-quad buffer at 1080p x 8 spp
export TMPDIR=/tmp
mkdir -p gpurun_out/r03d
CASE=${CASE:-2}
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/r03d/pmc_a -- python3 scripts/smalln.py > gpurun_out/r03d/pmc_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r03d/pmc_b -- python3 scripts/smalln.py > gpurun_out/r03d/pmc_b.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/r03d/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rpf::" not in k: continue
        short = k.split("(")[0].replace("void rpf::", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print("==", k, " VGPR", "")
    for c, v in sorted(d.items()):
        print("   %-24s mean %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
