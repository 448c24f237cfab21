#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Run in the build container (needs
/root/reference to build oracle/_ref/libref_mi.so):   python tests/golden/make_golden.py

  ref_mi.npz     known answers of the REAL reference code: MutualInformation / computeHistogram /
                 computeJointHistogram (mi.cpp) on vectors of length 1..392 incl. constant vectors, ties on bin
                 edges, value == max, heavy duplicates.
  ref_ops.npz    known answers of the REAL ops.h templates: getMean/getStdDev (12 and 19 columns, incl.
                 large-mean/small-variance rows), the 3-sigma test as rpf.cpp:577-580 composes it,
                 SampleData::normalized as sd.h:229-232 composes it.
  e2e_*.npz      small feature buffers (inputs stored verbatim) with the outputs of oracle/rpf_oracle.c:
                 filtered colours, N, member/bin hashes, alpha, beta, W_r_c, MI.  The oracle's MI and statistics
                 are pinned by the two files above; the glue around them is restated from rpf.cpp (which
                 cannot be compiled here: glog/OpenEXR absent) and is "parity unpinned" -- these files pin
                 GPU <-> oracle and guard the oracle against regressions.

  clustered_10x8x8.rpfb (+ _expected.npz)   an on-disk feature buffer in the .rpfb wire format and the oracle's
                 two-pass result on it.

Fixtures are data only: inputs and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import rpf_pkg  # noqa: E402

rpf_pkg.load()
import pyoracle as O  # noqa: E402
from raytracer_rpf_amd import feature_buffer as fb  # noqa: E402


def mi_cases(rng):
    cases = []
    for n in (1, 2, 3, 4, 8, 9, 15, 16, 17, 50, 100, 289, 392):
        x = rng.normal(size=n)
        cases += [(x, rng.normal(size=n)), (x, 0.7 * x + 0.3 * rng.normal(size=n)), (x, np.full(n, 1.25)),
                  (np.full(n, -3.0), np.full(n, 2.0)), (np.round(x * 2) / 2, np.round(rng.normal(size=n))),
                  (np.linspace(0.0, 1.0, n), np.linspace(1.0, 0.0, n) ** 2),                  # exact bin edges, value == max
                  (rng.integers(0, 3, n).astype(float), rng.integers(0, 2, n).astype(float)),  # {0,1,2} x {0,1}: ties
                  (np.float32(rng.random(n)).astype(float), np.float32(rng.random(n) * 1000).astype(float))]
    return cases


def rpfb_fixture():
    """clustered_10x8x8.rpfb: an on-disk feature buffer (feature_buffer.save_rpfb, with a ray-weight plane) and the
    oracle's two-pass {7, 5} result on it (filtered colours + pixel means): closes the loop file -> HIP path."""
    W, H, S = 10, 8, 8
    planes = fb.synth_planes(W, H, S, seed=77, sigma_f=1e-3, sigma_c=0.01, mode="clustered")
    rw = (0.5 + np.random.default_rng(77).random((H, W, S))).astype(np.float32)
    fb.save_rpfb(os.path.join(HERE, "clustered_10x8x8.rpfb"), planes, rw)
    c = None
    for box in (7, 5):
        c = O.filter_pass(planes, O.make_desc(W, H, S, box=box, policy=O.DEGEN_EPS, n_threads=1), colour_in=c,
                          debug=False)["colour"]
    pix = O.pixel_mean(c, O.make_desc(W, H, S), rw)
    np.savez_compressed(os.path.join(HERE, "clustered_10x8x8_expected.npz"), colour=c, pixel_rgb=pix,
                        boxes=np.array([7, 5], np.int32), policy=O.DEGEN_EPS,
                        source="oracle/rpf_oracle.c, EPS policy, beta map REF_GCC11_O3, on clustered_10x8x8.rpfb")
    cin = planes[2:5].astype(np.float64)
    print("clustered_10x8x8.rpfb  %.1f KB, activity %.3e" % (
        os.path.getsize(os.path.join(HERE, "clustered_10x8x8.rpfb")) / 1024, np.linalg.norm(c - cin) / np.linalg.norm(cin)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "rpfb":
        O.build(force=False)
        return rpfb_fixture()
    O.build(force=False)
    if not O.ref_available():
        sys.exit("oracle/_ref/libref_mi.so missing: this script must run where /root/reference exists")
    rng = np.random.default_rng(20250103)

    # ---- ref_mi.npz ---------------------------------------------------------------------------------
    xs, ys, lens, mis = [], [], [], []
    for x, y in mi_cases(rng):
        xs.append(x); ys.append(y); lens.append(len(x)); mis.append(O.ref_mi(x, y))
    np.savez_compressed(os.path.join(HERE, "ref_mi.npz"), x=np.concatenate(xs), y=np.concatenate(ys),
                        n=np.array(lens, np.int32), mi=np.array(mis),
                        source="MutualInformation() of /root/reference/src/custom/mi.cpp, g++ 11.4 -O3 -std=gnu++11")

    # ---- ref_ops.npz --------------------------------------------------------------------------------
    rows12 = [rng.normal(size=(8, 12)), np.float32(rng.normal(size=(16, 12)) * 0.05 + 1000.0).astype(float),
              np.tile(np.float32(rng.normal(size=(1, 12))).astype(float), (8, 1)),          # constant features
              np.float32(rng.random((64, 12))).astype(float)]
    rows19 = [np.float32(rng.normal(size=(n, 19)) * s + o).astype(float)
              for n, s, o in ((8, 1.0, 0.0), (49, 0.01, 300.0), (392, 1.0, 0.0), (200, 1e-3, -1000.0))]
    out = {}
    for i, r in enumerate(rows12):
        m, s = O.ref_mean_std(r)
        out["r12_%d" % i], out["m12_%d" % i], out["s12_%d" % i] = r, m, s
    for i, r in enumerate(rows19):
        m, s = O.ref_mean_std(r)
        out["r19_%d" % i], out["m19_%d" % i], out["s19_%d" % i] = r, m, s
        out["z19_%d" % i] = np.stack([O.ref_normalize(row, m, s) for row in r[:16]])
    # 3-sigma test: samples around the acceptance boundary, std == 0, std NaN
    f = rng.normal(size=(200, 12))
    mean = np.zeros(12)
    sd = np.full(12, 0.6)
    f[:20, 0] = 1.8            # exactly 3*0.6 -> a >= b fails (strict <)
    f[20:40, 0] = np.nextafter(1.8, 0)
    sds = np.tile(sd, (200, 1))
    sds[40:60, 3] = 0.0        # std 0 rejects everything
    sds[60:80, 5] = np.nan     # NaN std never rejects (a >= NaN is false)
    out["w3_f"], out["w3_mean"], out["w3_sd"] = f, mean, sds
    out["w3_pass"] = np.array([O.ref_within_3std(f[i], mean, sds[i]) for i in range(200)])
    np.savez_compressed(os.path.join(HERE, "ref_ops.npz"), **out,
                        source="templates of /root/reference/src/custom/ops.h, g++ 11.4 -O3 -std=gnu++11")

    # ---- e2e_*.npz ----------------------------------------------------------------------------------
    specs = [("e2e_clustered_12x10x8_box7", dict(W=12, H=10, S=8, mode="clustered", sigma_f=1e-3, sigma_c=0.01), 7, 0),
             ("e2e_smooth_10x8x8_box7", dict(W=10, H=8, S=8, mode="smooth", sigma_f=0.05, sigma_c=1e-4), 7, 0),
             ("e2e_clustered_8x6x16_box5", dict(W=8, H=6, S=16, mode="clustered", sigma_f=1e-3, sigma_c=0.01), 5, 1),
             ("e2e_constnormal_8x6x8_box7_eps", dict(W=8, H=6, S=8, mode="smooth", sigma_f=0.05, sigma_c=1e-4), 7, 1)]
    for name, gen, box, policy in specs:
        planes = fb.synth_planes(seed=7, **gen)
        if "constnormal" in name:
            planes[7:10] = np.float32([0.0, 0.0, 1.0])[:, None, None, None]
        r = O.filter_pass(planes, O.make_desc(gen["W"], gen["H"], gen["S"], box=box, policy=policy, n_threads=1))
        cin = planes[2:5].astype(np.float64)
        act = float(np.linalg.norm(r["colour"] - cin) / np.linalg.norm(cin))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), planes=planes, box=box, policy=policy,
                            colour=r["colour"], nbhd_size=r["nbhd_size"], member_hash=r["member_hash"],
                            bin_hash=r["bin_hash"], alpha=r["alpha"], beta=r["beta"], wrc=r["wrc"], mi=r["mi"],
                            mean=r["mean"], stddev=r["stddev"], status=r["status"],
                            nonfinite_pixels=r["nonfinite_pixels"], activity=act,
                            source="oracle/rpf_oracle.c (beta map REF_GCC11_O3); inputs stored verbatim")
        print("%-36s mean N %.1f activity %.3e status %d" % (name, r["sum_nbhd"] / (gen["W"] * gen["H"]), act, r["status"]))
    rpfb_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-40s %7.1f KB" % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


if __name__ == "__main__":
    main()
