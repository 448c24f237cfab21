#!/usr/bin/env python3
"""diagnostic (LAYOUT=19|27, S=16|32): one small frame whose pixels span the K = 13 and K = 25 size classes, filtered twice on the
default route and once with the split route off: which stage outputs differ between the runs, at which N, and who agrees with
the oracle.  Found the 27-dim failure of the LDS-head look-ups at K = 13 (rpf_filter_impl.inc, kHead)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch  # noqa
import rpf_pkg
rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip
import pyoracle as O
O.build()
W, H, S = 11, 8, int(os.environ.get("S", "32"))
WIDE = os.environ.get("LAYOUT", "27") == "27"
LK = dict(n_random=4, n_feat=18) if WIDE else {}
planes = fb.synth_planes(W, H, S, seed=91, sigma_f=0.05, sigma_c=1e-3, mode="smooth", dtype="f16" if WIDE else "f32", **LK)
desc = hip.make_desc(W, H, S, policy=hip.DEGEN_EPS, plane_dtype=hip.PLANES_F16 if WIDE else hip.PLANES_F32, **LK)
want = O.filter_pass(planes.astype(np.float32), O.make_desc(W, H, S, box=7, policy=O.DEGEN_EPS, **LK))
runs = {}
ctx = hip.Context(0)
for name, opt in (("on1", None), ("on2", None), ("off", ("split_weights", 0))):
    if opt: ctx.set_option(*opt)
    runs[name] = ctx.filter_pass_debug(planes, desc, box=7)
    if opt: ctx.set_option(opt[0], 0 if opt[0] == "split_chunk" else -1)
ref = runs["off"]
for name, r in runs.items():
    msg = []
    for k in ("nbhd_size", "mean", "stddev", "mi", "alpha", "beta", "wrc", "colour"):
        a, b = r[k], ref[k]
        if not np.array_equal(a, b, equal_nan=True):
            d = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            if k == "colour":
                pix = sorted(set((int(y), int(x)) for _, y, x, _ in d))
            else:
                pix = sorted(set((int(v[0]), int(v[1])) for v in d))
            msg.append("%s differs at %d entries, pixels %s N %s" % (k, len(d), pix[:6], [int(ref["nbhd_size"][y, x]) for y, x in pix[:6]]))
    mi_err = np.abs(r["mi"] - want["mi"]).max()
    col = np.linalg.norm(r["colour"] - want["colour"]) / np.linalg.norm(want["colour"])
    nb = want["nbhd_size"]
    bad = np.abs(r["mi"] - want["mi"]).max(axis=2) > 1e-9
    print("   N of pixels with wrong MI:", sorted(set(nb[bad].tolist()))[:12], "... N range of frame", int(nb.min()), int(nb.max()))
    print(name, "vs split-off:", "; ".join(msg) or "same bits", "| max |mi - oracle| %.2e  colour rel-L2 vs oracle %.2e" % (mi_err, col))
