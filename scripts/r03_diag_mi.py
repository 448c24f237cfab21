#!/usr/bin/env python3
"""Which pixels' MI values differ from the oracle's, by neighbourhood size, on the shape of test_waves_per_pixel_variants_agree
(11 x 9 pixels, S spp, box 7), with the waves_per_pixel option NW (0 = default)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rpf_pkg
rpf_pkg.load()
import pyoracle
pyoracle.build()
from raytracer_rpf_amd import feature_buffer as fb, hip
S, NW = int(os.environ.get("SPP", "32")), int(os.environ.get("NW", "1"))
W, H = int(os.environ.get("WIDTH", "11")), int(os.environ.get("ROWS", "9"))
planes = fb.synth_planes(W, H, S, seed=31 + S, sigma_f=0.05, sigma_c=1e-4, mode="smooth")
want = pyoracle.filter_pass(planes, pyoracle.make_desc(W, H, S, box=7))
ctx = hip.Context(0)
if NW:
    ctx.set_option("waves_per_pixel", NW)
got = ctx.filter_pass_debug(planes, hip.make_desc(W, H, S), box=7, allow_nonfinite=True)
N = got["nbhd_size"]
dmi = np.abs(got["mi"] - want["mi"])
bad = (dmi > 1e-11) | ~np.isfinite(got["mi"])
print("S %d NW %d: status %d, nonfinite pixels %d; pixels with a wrong MI: %d of %d" % (S, NW, got["status"], got["nonfinite_pixels"], int(bad.any(axis=-1).sum()), W * H))
for y in range(H):
    print(" ".join("%5d%s" % (N[y, x], "*" if bad[y, x].any() else " ") for x in range(W)))
ys, xs = np.nonzero(bad.any(axis=-1))
for y, x in list(zip(ys, xs))[:6]:
    prs = np.nonzero(bad[y, x])[0]
    print("pixel (x=%d,y=%d) N=%d: %d wrong pairs, first %s: got %s want %s" % (x, y, N[y, x], len(prs), prs[:6].tolist(),
          got["mi"][y, x, prs[:4]].tolist(), want["mi"][y, x, prs[:4]].tolist()))
