#!/usr/bin/env python3
"""Randomised parity sweep: random frame shapes, sample counts, box sizes, generator modes, policies and beta presets
through rpf_filter_pass_debug vs the CPU oracle, with the bars of tests/test_gpu_parity.py (bit-exact membership /
order / bins / statistics, MI 1e-11, alpha/beta/W 1e-9 under both policies, RGB 1e-4 rel-L2).  Round 2: both sample
layouts (19-dim fp32, 27-dim fp16), boxes up to 21 (neighbourhoods beyond 3136 samples run the streaming kernel).  Round 3:
flat-quad pixels (zero-variance normals: the stage-1a shortcut, the prelist, the packed kernels at N = S), and every case also
on the one-wave route (option packed = 0) with bit-identical stage outputs demanded between the two routes; the packed run
takes the probe's route, the fused route and the count-first route in turn (option count_first).
usage: fuzz_parity.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch  # noqa: F401
import rpf_pkg
rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip
import pyoracle as O

O.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
ctx = hip.Context(0)
fails = 0
for i in range(cases):
    box = int(rng.choice([3, 5, 7, 7, 7, 9, 11, 13, 17, 21]))
    wide = bool(rng.integers(0, 4) == 0)   # one case in four: the 27-dim fp16 layout
    L = dict(n_random=4, n_feat=18) if wide else {}
    smax = max(1, 65535 // (box * box))
    S = int(rng.choice([s for s in (1, 2, 3, 4, 5, 8, 8, 12, 16, 24, 32, 48, 64) if s <= smax]))
    W, H = int(rng.integers(3, 26)), int(rng.integers(2, 18))
    while W * H * S > (30000 if box > 11 or wide else 60000):  # keep the oracle in seconds
        W, H = max(3, W - 2), max(2, H - 1)
    mode = str(rng.choice(["smooth", "clustered"]))
    sf = float(rng.choice([1e-5, 1e-3, 0.02, 0.05]))
    policy = int(rng.choice([hip.DEGEN_EPS, hip.DEGEN_EPS, hip.DEGEN_REF_ABORT]))
    beta = int(rng.integers(0, 3))
    seed = int(rng.integers(0, 1 << 30))
    flat = float(rng.choice([0.0, 0.0, 0.5, 0.94]))
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != i:
        continue
    planes = fb.synth_planes(W, H, S, seed=seed, sigma_f=sf, sigma_c=0.01, mode=mode, dtype="f16" if wide else "f32", flat_frac=flat, **L)
    desc = hip.make_desc(W, H, S, policy=policy, beta_map=beta, plane_dtype=hip.PLANES_F16 if wide else hip.PLANES_F32, **L)
    ctx.set_option("count_first", (-1, 0, 1)[i % 3])   # the probe's choice, the fused route, the count-first route (box*box*S <= 512)
    got = ctx.filter_pass_debug(planes, desc, box=box, allow_nonfinite=True)
    ctx.set_option("count_first", -1)
    ctx.set_option("packed", 0)
    old = ctx.filter_pass_debug(planes, desc, box=box, allow_nonfinite=True)
    ctx.set_option("packed", -1)
    want = O.filter_pass(planes.astype(np.float32), O.make_desc(W, H, S, box=box, policy=policy, beta_map=beta, **L))
    ok, why = True, ""
    try:
        for k in ("nbhd_size", "member_hash", "bin_hash", "mean", "stddev", "mi", "alpha", "beta", "wrc"):
            assert np.array_equal(got[k], old[k], equal_nan=True), "packed vs one-wave route: " + k
        assert np.array_equal(np.isnan(got["colour"]), np.isnan(old["colour"])), "packed vs one-wave route: nan pattern"
        for k in ("nbhd_size", "member_hash", "bin_hash"):
            assert (got[k] == want[k]).all(), k
        assert np.array_equal(got["mean"], want["mean"], equal_nan=True), "mean"
        assert np.array_equal(got["stddev"], want["stddev"], equal_nan=True), "stddev"
        np.testing.assert_allclose(got["mi"], want["mi"], rtol=0, atol=1e-11, err_msg="mi")
        rt = 1e-9
        fin = np.isfinite(want["colour"]).all()
        for k in ("alpha", "beta", "wrc"):
            np.testing.assert_allclose(got[k], want[k], rtol=rt, atol=1e-12, err_msg=k, equal_nan=True)
        assert got["nonfinite_pixels"] == want["nonfinite_pixels"], "nonfinite count"
        if fin:
            r = float(np.linalg.norm(got["colour"] - want["colour"]) / max(np.linalg.norm(want["colour"]), 1e-300))
            assert r <= 1e-4, "rgb %g" % r
        else:
            m = np.isfinite(want["colour"])
            assert (np.isfinite(got["colour"]) == m).all(), "nan pattern"
    except AssertionError as e:
        ok, why = False, " | ".join(l.strip() for l in str(e).splitlines()[:12] if l.strip()) or "assert"
        fails += 1
        if os.environ.get("FUZZ_ONLY"):
            print("nonfinite pixels: gpu", got["nonfinite_pixels"], "gpu one-wave route", old["nonfinite_pixels"], "oracle", want["nonfinite_pixels"],
                  "redo", ctx.counters().redo_pixels, "first bad", got["first_bad_pixel"], want["first_bad_pixel"])
            gm, wm = ~np.isfinite(got["colour"]).all(axis=(0, 3)), ~np.isfinite(want["colour"]).all(axis=(0, 3))
            print("pixels NaN on one side only:", np.argwhere(gm != wm).tolist(), "N there:", want["nbhd_size"][gm != wm].tolist())
            bad = np.argwhere(~np.isfinite(want["colour"]).all(axis=(0, 3)))
            print("oracle non-finite pixels (y,x):", bad.tolist(), "gpu:", np.argwhere(~np.isfinite(got["colour"]).all(axis=(0, 3))).tolist())
            for (y, x) in bad.tolist()[:3]:
                for k in ("nbhd_size", "alpha", "beta", "wrc"):
                    print(k, "gpu", got[k][y, x], "oracle", want[k][y, x])
                print("mi diff max", np.nanmax(np.abs(got["mi"][y, x] - want["mi"][y, x])))
    print("%3d %s  %2dx%2dx%2d box %2d %s %-9s sf %-6g flat %-4g policy %d beta %d  maxN %4d small %3d%% bad %d  %s" % (
        i, "ok  " if ok else "FAIL", W, H, S, box, "d27" if wide else "d19", mode, sf, flat, policy, beta, int(want["nbhd_size"].max()),
        int(100 * (want["nbhd_size"] <= 64).mean()), want["nonfinite_pixels"], why), flush=True)
print("fuzz_parity: %d cases, %d failures" % (cases, fails))
sys.exit(1 if fails else 0)
