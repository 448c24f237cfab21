import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, rpf_pkg
rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip
import pyoracle as O
L27 = dict(n_random=4, n_feat=18)
ctx = hip.Context(0)
for (W, H, S, nw) in ((9, 7, 16, 4), (9, 7, 16, 1), (9, 7, 24, 4), (9, 7, 40, 4), (9, 7, 48, 4), (9, 7, 64, 4)):
    p16 = fb.synth_planes(W, H, S, dtype="f16", seed=19, sigma_f=0.05, sigma_c=1e-4, mode="smooth", **L27)
    want = O.filter_pass(p16.astype(np.float32), O.make_desc(W, H, S, box=7, policy=1, **L27))
    ctx.set_option("waves_per_pixel", nw)
    got = ctx.filter_pass_debug(p16, hip.make_desc(W, H, S, policy=1, plane_dtype=hip.PLANES_F16, **L27), box=7, allow_nonfinite=True)
    d = np.abs(got["mi"] - want["mi"]).reshape(H * W, -1).max(axis=1)
    N = want["nbhd_size"].reshape(-1)
    print(S, nw, "bad px", int((d > 1e-9).sum()), "of", H * W, "N bad", sorted(N[d > 1e-9].tolist())[:6], sorted(N[d > 1e-9].tolist())[-3:], "N good", sorted(N[d <= 1e-9].tolist())[:3], sorted(N[d <= 1e-9].tolist())[-3:])
