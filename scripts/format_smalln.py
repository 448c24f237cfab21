#!/usr/bin/env python3
"""profiles/r03_small_neighbourhood_regime.txt from the JSON lines of scripts/smalln.py (+ optional rocprofv3 kernel-stats CSVs:
label=path pairs after the smalln file)"""
import csv, json, sys
print("# Small-neighbourhood regime, round 3 (scripts/smalln.py; 1920 px wide, one box-7 pass, EPS policy; kernel_ms = hipEvents around the")
print("# pass inside the library).  Generators: sigma_f=1e-5 / 3e-3 jitter only; 94 % flat-quad pixels (zero-variance normal: N = S) next to")
print("# sigma_f=0.05 neighbours; 94 % flat quads + 1e-5 jitter = the stand-in for SURVEY F10's captured buffer (mean N 9.8, p99 26).")
print("# packed 1 = default route (probe -> fused or count-first route, flat plane + prelist, packed kernels), packed 0 = one wave per pixel")
print("# (round 2's route).  Round 1 (profiles/r01h_small_neighbourhood_regime.txt): 8 spp 592, 16 spp 911, 32 spp 908, 64 spp 629 Msamples/s")
print("# on the first generator.")
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    if "packed_vs_one_wave_rel_l2" in d:
        print("      packed vs one-wave route, filtered colours rel-L2: %g" % d["packed_vs_one_wave_rel_l2"])
        continue
    print("spp %2d rows %4d sigma_f %-6g flat %-4g packed %d | mean N %7.2f p50 %4.0f p90 %4.0f p99 %5.0f max %4d  N==S %5.1f%%  N<=64 %5.1f%% | kernel %7.2f ms  %7.1f Msamples/s  %.3f %% of 8 TB/s  launches %d"
          % (d["spp"], d["rows"], d["sigma_f"], d["flat_frac"], d["packed"], d["mean_nbhd"], d["p50"], d["p90"], d["p99"], d["max_nbhd"],
             100 * d["frac_N_eq_S"], 100 * d["frac_N_le_64"], d["kernel_ms"], d["Msamples_per_s"], 100 * d["hbm_frac_of_8TBs"], d["launches"]))
if len(sys.argv) > 2:
    print("\n# per-kernel average durations (rocprofv3 --kernel-trace --stats, 1080p x 8 spp, default route), microseconds")
    for lp in sys.argv[2:]:
        label, path = lp.split("=", 1)
        print("== " + label)
        for r in csv.DictReader(open(path)):
            if "rpf" in r["Name"]:
                print("   %-92s calls %3s avg_us %9.1f" % (r["Name"][:92], r["Calls"], float(r["AverageNs"]) / 1e3))
