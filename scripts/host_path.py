#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry rpf_filter() at BASELINE configs[1] (1920x1080x8spp, box 7):
band pipeline vs serial (RPF_FLAG_NO_OVERLAP), pageable vs page-locked (rpf_host_alloc) buffers; sample colours
and pixel means both returned.  Prints one line per variant.  (bench.py's `value` is the HBM-resident rate.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime first)
import rpf_pkg

rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip

W, H, S = 1920, int(os.environ.get("ROWS", "1080")), int(os.environ.get("SPP", "8"))
boxes = tuple(int(b) for b in os.environ.get("BOXES", "7").split(","))
dev = torch.device("cuda", 0)
planes = fb.synth_planes(W, H, S, xp=fb.torch_backend(dev), mode="smooth", sigma_f=0.05, sigma_c=1e-4).cpu().numpy()
ctx = hip.Context(0)
n = W * H * S * len(boxes)
pin = ctx.host_empty(planes.shape)
pin[...] = planes
out_s, out_p = ctx.host_empty((3, H, W, S)), ctx.host_empty((H, W, 3))
ref = None
for name, src, flags, outs in (("pageable serial ", planes, hip.FLAG_NO_OVERLAP, {}),
                               ("pageable pipeline", planes, 0, {}),
                               ("pinned serial    ", pin, hip.FLAG_NO_OVERLAP, dict(out_samples=out_s, out_pixels=out_p)),
                               ("pinned pipeline  ", pin, 0, dict(out_samples=out_s, out_pixels=out_p))):
    d = hip.make_desc(W, H, S, boxes=boxes, flags=flags)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        s_rgb, p_rgb, _ = ctx.filter(src, d, **outs)
        best = min(best, time.perf_counter() - t)
    if ref is None:
        ref = (s_rgb.copy(), p_rgb.copy())
    same = np.array_equal(s_rgb, ref[0]) and np.array_equal(p_rgb, ref[1])
    c = ctx.counters()
    print("%s  %7.1f ms  %6.1f Msamples/s (PCIe-inclusive)  h2d %.1f ms d2h %.1f ms  identical=%s"
          % (name, best * 1e3, n / best / 1e6, c.h2d_ms, c.d2h_ms, same), flush=True)
