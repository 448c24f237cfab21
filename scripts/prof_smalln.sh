export TMPDIR=/tmp
OUT=gpurun_out/prof_smalln
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | head -40 > $OUT/counters_list.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH --output-format csv -d $OUT/pmc1 -- python3 scripts/smalln.py > $OUT/pmc1.log 2>&1 || echo pmc1 failed
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc2 -- python3 scripts/smalln.py > $OUT/pmc2.log 2>&1 || echo pmc2 failed
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("gpurun_out/prof_smalln/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_pixel_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-24s mean %.5g (n=%d)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
cat $OUT/counters_list.txt | head -20
tail -3 $OUT/pmc1.log $OUT/pmc2.log
