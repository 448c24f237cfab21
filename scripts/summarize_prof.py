"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield f, r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f, r in rows("trace/**/*kernel_stats.csv"):
    print("%-90s calls %6s total_ns %14s avg_ns %14s pct %6s" % (
        r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

print("== per-dispatch counters, per kernel of the library (mean over its dispatches) ==")
acc = defaultdict(lambda: defaultdict(list))
for f, r in rows("pmc_*/**/*counter_collection.csv"):
    name = r.get("Kernel_Name", "")
    if "rpf::" not in name:
        continue
    short = name.split("(")[0].replace("void ", "").replace("rpf::(anonymous namespace)::", "").replace("rpf::", "")
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
# the instantiation that owns the time first
order = sorted(acc, key=lambda k: -sum(acc[k].get("GRBM_GUI_ACTIVE", acc[k].get("SQ_WAVE_CYCLES", [0]))))
for kern in order:
    print("-- %s" % kern)
    a = acc[kern]
    for k in sorted(a):
        v = a[k]
        print("%-28s mean %.6g  (n=%d)" % (k, sum(v) / len(v), len(v)))
    if "FETCH_SIZE" in a:
        f = sum(a["FETCH_SIZE"]) / len(a["FETCH_SIZE"])
        print("FETCH_SIZE is reported in KiB; gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM): "
              "raw %.4g KiB -> bytes %.4g .. corrected x2 %.4g" % (f, f * 1024, f * 2048))
    if "WRITE_SIZE" in a:
        w = sum(a["WRITE_SIZE"]) / len(a["WRITE_SIZE"])
        print("WRITE_SIZE raw %.4g KiB -> bytes %.4g" % (w, w * 1024))
