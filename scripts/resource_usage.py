#!/usr/bin/env python3
"""Registers / spills / occupancy of every filter_pixel_kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage
on the kernel TU, device side only).  usage: python scripts/resource_usage.py [remarks.txt]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    text = open(sys.argv[1]).read()
else:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "raytracer-rpf_amd", "csrc"),
           "--cuda-device-only", "-c", os.path.join(ROOT, "raytracer-rpf_amd", "csrc", "rpf_kernels.hip"),
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    text = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
for b in re.split(r"remark: Function Name: ", text)[1:]:
    name = b.split()[0]
    m = re.search(r"filter_pixel_kernelILi(\d+)ELb(\d)ELb(\d)ELi(\d)E", name)
    if not m:
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
    print("K=%-2s table_in_lds=%s fast=%s NW=%s  VGPR %3s AGPR %3s spill %3s scratch %4s B/lane  waves/SIMD %s"
          % (m.groups() + (g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"))))
