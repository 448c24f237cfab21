# VALU / LDS / SALU / VMEM instruction counts of the fused kernel per stage, from differences between stage_mask runs
export TMPDIR=/tmp
for m in -1 0 1 3 7; do
  rm -rf gpurun_out/ibs_$m
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d gpurun_out/ibs_$m -- python3 bench.py --option stage_mask=$m --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
res = {}
for m in (-1, 0, 1, 3, 7):
    acc = defaultdict(list)
    for f in glob.glob("gpurun_out/ibs_%d/**/*counter_collection.csv" % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "filter_pixel_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[m] = {k: sum(v) / len(v) for k, v in acc.items()}
px = 1920 * 1080
names = ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
stages = [("gather+misc", 0, None), ("chain", 1, 0), ("bins", 3, 1), ("MI", 7, 3), ("weights", -1, 7), ("TOTAL", -1, None)]
print("per-pixel counts:  %-12s" % "stage" + "".join("%22s" % n for n in names))
for name, a, b in stages:
    row = [(res[a].get(n, 0) - (res[b].get(n, 0) if b is not None else 0)) / px for n in names]
    print("                   %-12s" % name + "".join("%22.0f" % v for v in row))
PY
