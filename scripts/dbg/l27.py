import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, rpf_pkg
rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip
import pyoracle as O
L27 = dict(n_random=4, n_feat=18)
W, H, S, box = 9, 7, 32, 7
p16 = fb.synth_planes(W, H, S, dtype="f16", seed=19, sigma_f=0.05, sigma_c=1e-4, mode="smooth", **L27)
p32 = p16.astype(np.float32)
want = O.filter_pass(p32, O.make_desc(W, H, S, box=box, policy=1, **L27))
ctx = hip.Context(0)
for opts in ({}, {"waves_per_pixel": 1}, {"binning": 0}, {"table_in_lds": 0}):
    for k, v in opts.items(): ctx.set_option(k, v)
    got = ctx.filter_pass_debug(p16, hip.make_desc(W, H, S, policy=1, plane_dtype=hip.PLANES_F16, **L27), box=box, allow_nonfinite=True)
    for k, v in opts.items(): ctx.set_option(k, -1 if k != "waves_per_pixel" else 0)
    bad = ~np.isfinite(got["colour"]).all(axis=(0, 3))
    print(opts, "nonfinite", got["nonfinite_pixels"], "N of bad px", sorted(set(want["nbhd_size"][bad].tolist()))[:8],
          "N range", want["nbhd_size"].min(), want["nbhd_size"].max())
    for k in ("nbhd_size", "member_hash", "bin_hash"):
        print("   ", k, "mismatch px", int((got[k] != want[k]).reshape(H * W, -1).any(axis=1).sum()))
    for k in ("mean", "stddev", "mi", "alpha", "beta", "wrc"):
        d = np.abs(got[k] - want[k]); 
        print("   ", k, "max abs diff", float(np.nanmax(d)), "nan in got", int(np.isnan(got[k]).sum()))
got = ctx.filter_pass_debug(p16, hip.make_desc(W, H, S, policy=1, plane_dtype=hip.PLANES_F16, **L27), box=box, allow_nonfinite=True)
d = np.abs(got["mi"] - want["mi"]).reshape(H * W, -1)
badpix = np.where(d.max(axis=1) > 1e-9)[0]
print("bad pixels", badpix[:40], "N", want["nbhd_size"].reshape(-1)[badpix][:40])
goodpix = np.where(d.max(axis=1) <= 1e-9)[0]
print("good N", want["nbhd_size"].reshape(-1)[goodpix][:40])
p0 = badpix[0]
print("bad pair idx", np.where(d[p0] > 1e-9)[0])
print("got", got["mi"].reshape(H * W, -1)[p0][np.where(d[p0] > 1e-9)[0]][:12])
print("want", want["mi"].reshape(H * W, -1)[p0][np.where(d[p0] > 1e-9)[0]][:12])
