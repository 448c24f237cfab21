"""CPU tests (-m "not gpu"): the oracle against (1) golden vectors produced by the REAL reference code
(mi.cpp, ops.h), (2) the compiled reference itself when oracle/_ref/libref_mi.so is present, (3) its own
committed end-to-end fixtures.  These pin the checker; nothing here touches the product path."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


def test_mi_matches_reference_known_answers(oracle):
    g = load("ref_mi.npz")
    off = 0
    for n, want in zip(g["n"], g["mi"]):
        x, y = g["x"][off:off + n], g["y"][off:off + n]
        off += n
        assert same(oracle.mi(x, y), want), (n, oracle.mi(x, y), want)


def test_mean_std_and_normalise_match_reference_known_answers(oracle):
    g = load("ref_ops.npz")
    for tag, count in (("12", 4), ("19", 4)):
        for i in range(count):
            m, s = oracle.mean_std(g["r%s_%d" % (tag, i)])
            assert np.array_equal(m, g["m%s_%d" % (tag, i)])
            assert np.array_equal(s, g["s%s_%d" % (tag, i)], equal_nan=True)
    # SampleData::normalized (sd.h:229-232): (x - mean) / std with std == 0 -> 0
    for i in range(4):
        r, m, s = g["r19_%d" % i], g["m19_%d" % i], g["s19_%d" % i]
        with np.errstate(divide="ignore", invalid="ignore"):
            z = np.where(s == 0, 0.0, (r[:16] - m) / s)
        assert np.array_equal(z, g["z19_%d" % i], equal_nan=True)


def test_three_sigma_rule_matches_reference_known_answers(oracle):
    """a sample passes iff NOT (|f-m| >= 3*sd) for every feature: strict <, sd == 0 rejects, NaN sd accepts"""
    g = load("ref_ops.npz")
    f, mean, sd = g["w3_f"], g["w3_mean"], g["w3_sd"]
    with np.errstate(invalid="ignore"):
        ours = ~np.any(np.abs(f - mean) >= sd * 3, axis=1)
    assert np.array_equal(ours, g["w3_pass"])
    assert not g["w3_pass"][:20].any() and not g["w3_pass"][40:60].any()


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libref_mi.so")),
                    reason="oracle/_ref/libref_mi.so not built (needs /root/reference)")
def test_oracle_against_compiled_reference_randomised(oracle):
    rng = np.random.default_rng(5)
    for n in (1, 2, 5, 8, 49, 64, 200, 392, 784):
        for t in range(12):
            x = rng.normal(size=n)
            y = rng.normal(size=n) + (t % 3) * 0.5 * x
            if t % 4 == 0:
                x = np.round(x * 3) / 3
            if t % 6 == 0:
                y = np.full(n, 0.25)
            assert same(oracle.mi(x, y), oracle.ref_mi(x, y))
    for nc in (12, 19):
        r = np.float32(rng.normal(size=(300, nc)) * 0.02 + 500).astype(float)
        m, s = oracle.mean_std(r)
        m2, s2 = oracle.ref_mean_std(r)
        assert np.array_equal(m, m2) and np.array_equal(s, s2, equal_nan=True)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libref_mi.so")),
                    reason="oracle/_ref/libref_mi.so not built (needs /root/reference)")
def test_oracle_stage3_inputs_and_mi_equal_the_compiled_reference(oracle):
    """Rebuild one pixel's neighbourhood statistics, normalisation and all 96 MI values with the compiled
    reference functions (getMean/getStdDev, divideArrays(subtractArrays), MutualInformation) from the oracle's
    own member list, and require bit-identity with what the oracle's filter pass reports."""
    from raytracer_rpf_amd import feature_buffer as fb
    W, H, S, box = 9, 8, 8, 7
    planes = fb.synth_planes(W, H, S, seed=13, sigma_f=1e-3, sigma_c=0.01, mode="clustered")
    r = oracle.filter_pass(planes, oracle.make_desc(W, H, S, box=box, n_threads=1))
    pa, pb = oracle.pair_table()
    b = (box - 1) // 2
    mean12, sd12 = oracle.pixel_stats(planes, oracle.make_desc(W, H, S))
    for (y, x) in ((4, 4), (0, 0), (7, 3)):
        rows = [planes[:, y, x, s].astype(float) for s in range(S)]
        for xn in range(x - b, x + b + 1):
            for yn in range(y - b, y + b + 1):
                if (xn, yn) == (x, y) or not (0 <= xn < W and 0 <= yn < H):
                    continue
                for s in range(S):
                    v = planes[:, yn, xn, s].astype(float)
                    if oracle.ref_within_3std(v[7:], mean12[y, x], sd12[y, x]):
                        rows.append(v)
        rows = np.array(rows)
        assert len(rows) == r["nbhd_size"][y, x]
        m, s = oracle.ref_mean_std(rows)
        assert np.array_equal(m, r["mean"][y, x]) and np.array_equal(s, r["stddev"][y, x])
        z = np.stack([oracle.ref_normalize(v, m, s) for v in rows])
        mi = np.array([oracle.ref_mi(z[:, a], z[:, bb]) for a, bb in zip(pa, pb)])
        assert np.array_equal(mi, r["mi"][y, x])


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libref_mi.so")),
                    reason="oracle/_ref/libref_mi.so not built (needs /root/reference)")
def test_oracle_stage4a_distances_equal_the_compiled_reference(oracle):
    """stage 4a (rpf.cpp:646-670): the oracle's three weighted squared distances of normalised sample pairs against
    sumArray(multiplyArrays(squareArray(subtractArrays(.,.)), w)) composed from the compiled ops.h, bit for bit, on the
    normalised rows and the alpha / beta of real pixels (incl. large-magnitude rows and NaN / inf weights)"""
    from raytracer_rpf_amd import feature_buffer as fb
    W, H, S, box = 9, 8, 8, 7
    planes = fb.synth_planes(W, H, S, seed=13, sigma_f=1e-3, sigma_c=0.01, mode="clustered")
    planes[10:13] += 700.0
    r = oracle.filter_pass(planes, oracle.make_desc(W, H, S, box=box, n_threads=1))
    rng = np.random.default_rng(9)
    checked = 0
    for (y, x) in ((4, 4), (0, 0), (7, 3), (2, 8)):
        m, s = r["mean"][y, x], r["stddev"][y, x]
        own = np.stack([oracle.ref_normalize(planes[:, y, x, i].astype(float), m, s) for i in range(S)])
        others = np.stack([oracle.ref_normalize(planes[:, (y + dy) % H, (x + dx) % W, i].astype(float), m, s)
                           for dy, dx, i in rng.integers(0, 4, size=(24, 3))])
        for zi in own:
            for zj in np.concatenate([own, others]):
                got = oracle.weighted_sqdist(zi, zj, r["alpha"][y, x], r["beta"][y, x])
                want = oracle.ref_weighted_sqdist(zi, zj, r["alpha"][y, x], r["beta"][y, x])
                assert np.array_equal(got, want, equal_nan=True), (y, x, got, want)
                checked += 1
    # non-finite weights propagate identically
    z = rng.normal(size=(2, 19))
    for a in (np.array([np.nan, 1.0, 2.0]), np.array([np.inf, 0.0, -1.0])):
        b = np.concatenate([a, rng.normal(size=9)])
        assert np.array_equal(oracle.weighted_sqdist(z[0], z[1], a, b), oracle.ref_weighted_sqdist(z[0], z[1], a, b),
                              equal_nan=True)
    assert checked >= 1000


def test_eps_residue_contract(oracle):
    """EPS policy (documented deviation): an MI whose 2^-44 fixed-point integer form lies inside the table's rounding
    band is exactly 0.  Constructed exactly-independent 3x3 tables at N = 12 (not a power of two): the reference's own
    value (REF_ABORT: mi.cpp statement by statement) is rounding residue of either sign or an accidental 0; under EPS
    every such pair is 0 and the weights are the clean 0 / (0 + 0 + eps) limits."""
    # 12 samples, x in 3 bins (4,4,4), y in 2 bins (6,6): J_ij = 2 on every occupied cell -> J*N == hx*hy
    x = np.repeat([0.0, 0.5, 1.0], 4)
    y = np.tile([0.0, 0.0, 1.0, 1.0], 3)
    assert abs(oracle.mi(x, y)) < 1e-15          # the reference: residue or 0, never a real dependence
    z = np.zeros((12, 19))
    z[:, 5], z[:, 6] = x, y                       # random parameters
    z[:, 0], z[:, 1] = y, x                       # pFilm
    for k in range(12):
        z[:, 7 + k] = np.roll(x, 0) if k % 2 == 0 else y   # features exactly independent of r and p where they differ
    for c in range(3):
        z[:, 2 + c] = y if c % 2 == 0 else x
    a_eps, b_eps, w_eps, mi_eps = oracle.cf_weights(z, policy=oracle.DEGEN_EPS)
    a_ref, b_ref, w_ref, mi_ref = oracle.cf_weights(z, policy=oracle.DEGEN_REF_ABORT)
    indep = np.abs(mi_ref) < 1e-12
    assert indep.any() and (mi_eps[indep] == 0).all()
    assert np.array_equal(mi_eps[~indep], mi_ref[~indep])   # real dependences are untouched by the contract
    assert np.isfinite(a_eps).all() and np.isfinite(b_eps).all() and np.isfinite(w_eps)


@pytest.mark.parametrize("name", ["e2e_clustered_12x10x8_box7", "e2e_smooth_10x8x8_box7",
                                  "e2e_clustered_8x6x16_box5", "e2e_constnormal_8x6x8_box7_eps"])
def test_oracle_reproduces_committed_end_to_end_fixtures(oracle, name):
    g = load(name + ".npz")
    planes = g["planes"]
    _, H, W, S = planes.shape
    r = oracle.filter_pass(planes, oracle.make_desc(W, H, S, box=int(g["box"]), policy=int(g["policy"])))
    assert np.array_equal(r["nbhd_size"], g["nbhd_size"])
    assert np.array_equal(r["member_hash"], g["member_hash"]) and np.array_equal(r["bin_hash"], g["bin_hash"])
    assert np.array_equal(r["colour"], g["colour"], equal_nan=True)
    assert np.array_equal(r["alpha"], g["alpha"], equal_nan=True) and np.array_equal(r["beta"], g["beta"], equal_nan=True)
    assert r["status"] == int(g["status"])


def test_oracle_generalised_layout(oracle):
    """the oracle's loops take the column-group sizes from the descriptor: with (2, 12) they are the reference's (every
    pin above runs through them); with (4, 18) the pair order is rpf.cpp:416-442's with the bounds widened"""
    a, b = oracle.pair_table(4, 18)
    assert len(a) == 18 * 6 + 3 * 24 == 180
    assert list(a[:6]) == [9] * 6 and list(b[:6]) == [5, 6, 7, 8, 0, 1]          # f0 x (r0..r3, p0, p1)
    assert list(a[108:114]) == [2] * 6 and list(b[108:114]) == [5, 6, 7, 8, 0, 1] and list(b[114:132]) == list(range(9, 27))
    a19, b19 = oracle.pair_table()
    a2, b2 = oracle.pair_table(2, 12)
    assert np.array_equal(a19, a2) and np.array_equal(b19, b2)
    # a 27-dim buffer whose extra columns are copies: the shared statistics equal the 19-dim run's
    from raytracer_rpf_amd import feature_buffer as fb
    W, H, S = 9, 7, 8
    p19 = fb.synth_planes(W, H, S, seed=4, sigma_f=1e-3, sigma_c=0.01, mode="clustered")
    p27 = fb.synth_planes(W, H, S, seed=4, sigma_f=1e-3, sigma_c=0.01, mode="clustered", n_random=4, n_feat=18)
    assert np.array_equal(p27[:7], p19[:7]) and np.array_equal(p27[9:21], p19[7:19])
    r19 = oracle.filter_pass(p19, oracle.make_desc(W, H, S, policy=1))
    r27 = oracle.filter_pass(p27, oracle.make_desc(W, H, S, policy=1, n_random=4, n_feat=18))
    assert r27["mi"].shape[-1] == 180 and r27["beta"].shape[-1] == 18
    same = r19["nbhd_size"] == r27["nbhd_size"]     # six more features can only reject more neighbours
    assert (r27["nbhd_size"] <= r19["nbhd_size"]).all() and same.any()
    assert np.array_equal(r27["mean"][same][:, :7], r19["mean"][same][:, :7])
    assert np.array_equal(r27["mean"][same][:, 9:21], r19["mean"][same][:, 7:19])


def test_oracle_invariants(oracle):
    """Appendix-A invariants: own samples are the first S members so w_ii = 1; non-colour columns are never
    written; thread count does not change results; pair table is the ComputeCFWeights call order."""
    from raytracer_rpf_amd import feature_buffer as fb
    W, H, S = 10, 9, 8
    planes = fb.synth_planes(W, H, S, seed=21, sigma_f=1e-3, sigma_c=0.01, mode="clustered")
    r1 = oracle.filter_pass(planes, oracle.make_desc(W, H, S, n_threads=1))
    r4 = oracle.filter_pass(planes, oracle.make_desc(W, H, S, n_threads=4))
    assert np.array_equal(r1["colour"], r4["colour"]) and np.array_equal(r1["mi"], r4["mi"])
    assert (r1["nbhd_size"] >= S).all() and (r1["nbhd_size"] <= 49 * S).all()
    assert (r1["mi"] > -1e-12).all()
    a, b = oracle.pair_table()
    assert list(a[:4]) == [7, 7, 7, 7] and list(b[:4]) == [5, 6, 0, 1]
    assert list(a[48:52]) == [2, 2, 2, 2] and list(b[48:52]) == [5, 6, 0, 1] and list(b[52:64]) == list(range(7, 19))
    # rows outside [row_begin,row_end) pass through
    r = oracle.filter_pass(planes, oracle.make_desc(W, H, S, row_begin=3, row_end=6), debug=False)
    cin = planes[2:5].astype(np.float64)
    assert np.array_equal(r["colour"][:, :3], cin[:, :3]) and np.array_equal(r["colour"][:, 6:], cin[:, 6:])
    assert np.array_equal(r["colour"][:, 3:6], r1["colour"][:, 3:6])


def test_reflog_matches_libm_next_to_one(tmp_path):
    """csrc/rpf_reflog.h restates glibc's log() for arguments next to 1 (the REF_ABORT residue path of the streaming kernel
    evaluates mi.cpp:84 with it on the device).  Compiled for the host, it must equal this machine's libm bit for bit on
    q = 1 +- k ulp (the only arguments the residue path sees) and on |q - 1| < 2^-10; further out (to 2^-5) libm's own
    FMA / non-FMA builds differ in the last bit, so only near-total agreement is asked there."""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#define RPF_HD static inline
#include "rpf_reflog.h"
static double fb(uint64_t u){double d; memcpy(&d,&u,8); return d;}
static uint64_t bits(double d){uint64_t u; memcpy(&u,&d,8); return u;}
int main(){
  long bad=0; const uint64_t one = bits(1.0);
  for (long k=-4096;k<=4096;++k){ double q=fb(one + k); if (bits(rpf::reflog_near_one(q)) != bits(std::log(q))) ++bad; }
  printf("ulp %ld\n", bad);
  for (int e=-50;e<=-5;e+=5){ long b2=0,n2=0; srand48(e+100); for(int i=0;i<100000;++i){ double q=1.0+std::ldexp(drand48()*2-1,e);
      if(!rpf::reflog_in_range(q)) continue; ++n2; if(bits(rpf::reflog_near_one(q))!=bits(std::log(q))) ++b2;} printf("e %d %ld %ld\n", e, b2, n2);}
}''')
    exe = tmp_path / "t"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer-rpf_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I" + inc, "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
    assert out[0] == "ulp 0"
    for line in out[1:]:
        if not line:
            continue
        _, e, bad, n = line.split()
        if int(e) <= -10:
            assert int(bad) == 0, line
        else:
            assert int(bad) <= 1e-3 * int(n), line


def test_spill_scanner_rules(tmp_path):
    """scripts/check_spills.py: spill code (scratch stores / reloads, accvgpr copies) between a block label and the first
    EXEC-enabling instruction is refused -- with waits / scalar moves in between too -- while an `if` body that fetches a
    parked operand, or the s_mov-and-s_and head of an inner `if`, passes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = {
        "hazard_plain": ("_Zk:\n.LBB0_1:\n\tscratch_store_dword off, v2, off ; 4-byte Folded Spill\n\ts_or_b64 exec, exec, s[0:1]\n\ts_endpgm\n", 1),
        "hazard_interleaved": ("_Zk:\n.LBB0_1:\n\ts_waitcnt vmcnt(0)\n\tv_accvgpr_write_b32 a1, v2\n\ts_mov_b32 s5, 0\n\ts_mov_b64 exec, s[0:1]\n\ts_endpgm\n", 1),
        "hazard_reload": ("_Zk:\n.LBB0_1:\n\tscratch_load_dword v2, off, off ; 4-byte Folded Reload\n\ts_or_b64 exec, exec, s[0:1]\n\ts_endpgm\n", 1),
        "ok_body": ("_Zk:\n.LBB0_1:\n\tv_accvgpr_read_b32 v4, a1\n\tv_add_f64 v[4:5], v[4:5], v[6:7]\n\ts_or_b64 exec, exec, s[0:1]\n\ts_endpgm\n", 0),
        "ok_narrow": ("_Zk:\n.LBB0_1:\n\tv_accvgpr_read_b32 v4, a1\n\ts_mov_b64 s[0:1], exec\n\ts_and_b64 s[2:3], s[0:1], vcc\n\ts_mov_b64 exec, s[2:3]\n\ts_endpgm\n", 0),
    }
    for name, (txt, want) in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(txt)
        rc = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_spills.py"), str(f)], stdout=subprocess.PIPE).returncode
        assert rc == want, name
