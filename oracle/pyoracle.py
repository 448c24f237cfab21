"""ctypes doorway onto oracle/librpf_oracle.so (fp64 C restatement) and oracle/_ref/libref_mi.so (the
real reference mi.cpp + ops.h built from /root/reference).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package (raytracer-rpf_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NDIM, NFEAT, NPAIR = 19, 12, 96

BETA_REF_GCC11_O3, BETA_REF_GCC11_O2, BETA_PAPER = 0, 1, 2
DEGEN_REF_ABORT, DEGEN_EPS = 0, 1


class Desc(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("S", C.c_int32), ("row_begin", C.c_int32),
                ("row_end", C.c_int32), ("box", C.c_int32), ("beta_map", C.c_int32),
                ("degenerate_policy", C.c_int32), ("eps", C.c_double), ("sigma_seed", C.c_double),
                ("n_threads", C.c_int32), ("reserved", C.c_int32), ("n_random", C.c_int32), ("n_feat", C.c_int32)]


def dims(desc):
    """(ndim, nfeat, npair) of a descriptor's sample layout (0 fields = the reference's 19 / 12 / 96)"""
    nr, nf = desc.n_random or 2, desc.n_feat or 12
    return 5 + nr + nf, nf, nf * (nr + 2) + 3 * (nr + 2 + nf)


class Debug(C.Structure):
    _fields_ = [("nbhd_size", C.c_void_p), ("mean", C.c_void_p), ("stddev", C.c_void_p), ("mi", C.c_void_p),
                ("alpha", C.c_void_p), ("beta", C.c_void_p), ("wrc", C.c_void_p), ("bin_hash", C.c_void_p),
                ("member_hash", C.c_void_p)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("first_bad_pixel", C.c_int32), ("nonfinite_pixels", C.c_int64),
                ("sum_nbhd", C.c_int64), ("max_nbhd", C.c_int32), ("reserved", C.c_int32)]


def build(force=False):
    """(Re)build the oracle libraries with oracle/Makefile (gcc only; _ref needs /root/reference)."""
    lib = os.path.join(_HERE, "librpf_oracle.so")
    src_newer = (not os.path.exists(lib)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(lib) for f in ("rpf_oracle.c", "rpf_oracle.h"))
    if force or src_newer:
        subprocess.check_call(["make", "-C", _HERE, "librpf_oracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(_HERE, "_ref", "libref_mi.so")
    if os.path.isdir("/root/reference/src/custom") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "librpf_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.rpf_oracle_mi.restype = C.c_double
        L.rpf_oracle_mi.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.rpf_oracle_mean_std.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.rpf_oracle_pixel_stats.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpf_oracle_cf_weights.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpf_oracle_filter_pass.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(Debug), C.POINTER(Result)]
        L.rpf_oracle_pixel_mean.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpf_oracle_pair_table.argtypes = [C.c_void_p, C.c_void_p]
        L.rpf_oracle_pair_table_ex.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.rpf_oracle_feature_images.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_void_p]
        L.rpf_oracle_weighted_sqdist.argtypes = [C.c_void_p] * 5
        _lib = L
    return _lib


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_mi.so"))


def ref():
    """The compiled reference (mi.cpp + ops.h). Raises if oracle/_ref/libref_mi.so was never built."""
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(_HERE, "_ref", "libref_mi.so"))
        R.ref_mutual_information.restype = C.c_double
        R.ref_mutual_information.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        R.ref_histogram.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        R.ref_joint_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.c_void_p]
        for n in ("ref_mean_std_12", "ref_mean_std_19"):
            getattr(R, n).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_within_3std_12.restype = C.c_int
        R.ref_within_3std_12.argtypes = [C.c_void_p] * 3
        R.ref_normalize_19.argtypes = [C.c_void_p] * 4
        R.ref_weighted_sqdist_2.restype = C.c_double
        R.ref_weighted_sqdist_2.argtypes = [C.c_void_p] * 2
        R.ref_weighted_sqdist_3.restype = C.c_double
        R.ref_weighted_sqdist_3.argtypes = [C.c_void_p] * 3
        R.ref_weighted_sqdist_12.restype = C.c_double
        R.ref_weighted_sqdist_12.argtypes = [C.c_void_p] * 3
        _ref = R
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- reference (real mi.cpp / ops.h) -----------------------------------------------------------------
def ref_mi(x, y):
    x, y = _f64(x), _f64(y)
    return ref().ref_mutual_information(_p(x), _p(y), len(x))


def ref_mean_std(rows):
    rows = _f64(rows)
    n, nc = rows.shape
    m, s = np.empty(nc), np.empty(nc)
    {12: ref().ref_mean_std_12, 19: ref().ref_mean_std_19}[nc](_p(rows), n, _p(m), _p(s))
    return m, s


def ref_within_3std(f, mean, sd):
    f, mean, sd = _f64(f), _f64(mean), _f64(sd)
    return bool(ref().ref_within_3std_12(_p(f), _p(mean), _p(sd)))


def ref_normalize(x, mean, sd):
    x, mean, sd = _f64(x), _f64(mean), _f64(sd)
    out = np.empty(19)
    ref().ref_normalize_19(_p(x), _p(mean), _p(sd), _p(out))
    return out


# ---- oracle (C restatement) --------------------------------------------------------------------------
def mi(x, y):
    x, y = _f64(x), _f64(y)
    return lib().rpf_oracle_mi(_p(x), _p(y), len(x))


def mean_std(rows):
    rows = _f64(rows)
    n, nc = rows.shape
    m, s = np.empty(nc), np.empty(nc)
    lib().rpf_oracle_mean_std(_p(rows), n, nc, _p(m), _p(s))
    return m, s


def weighted_sqdist(zi, zj, alpha, beta):
    """stage 4a's (position, colour, feature) weighted squared distances of two normalised 19-vectors"""
    zi, zj, alpha, beta = _f64(zi), _f64(zj), _f64(alpha), _f64(beta)
    out = np.empty(3)
    lib().rpf_oracle_weighted_sqdist(_p(zi), _p(zj), _p(alpha), _p(beta), _p(out))
    return out


def ref_weighted_sqdist(zi, zj, alpha, beta):
    """the same three sums composed from the compiled ops.h templates as rpf.cpp:646-660 composes them"""
    zi, zj, alpha, beta = _f64(zi), _f64(zj), _f64(alpha), _f64(beta)
    R = ref()
    p_i, p_j = np.ascontiguousarray(zi[0:2]), np.ascontiguousarray(zj[0:2])
    c_i, c_j = np.ascontiguousarray(zi[2:5]), np.ascontiguousarray(zj[2:5])
    f_i, f_j = np.ascontiguousarray(zi[7:19]), np.ascontiguousarray(zj[7:19])
    return np.array([R.ref_weighted_sqdist_2(_p(p_i), _p(p_j)), R.ref_weighted_sqdist_3(_p(c_i), _p(c_j), _p(alpha)),
                     R.ref_weighted_sqdist_12(_p(f_i), _p(f_j), _p(beta))])


def pair_table(n_random=2, n_feat=12):
    npair = n_feat * (n_random + 2) + 3 * (n_random + 2 + n_feat)
    a, b = np.empty(npair, np.int32), np.empty(npair, np.int32)
    lib().rpf_oracle_pair_table_ex(n_random, n_feat, _p(a), _p(b))
    return a, b


def make_desc(W, H, S, box=7, row_begin=0, row_end=None, beta_map=BETA_REF_GCC11_O3, policy=DEGEN_REF_ABORT,
              eps=1e-10, sigma_seed=0.002, n_threads=0, n_random=0, n_feat=0):
    return Desc(W, H, S, row_begin, H if row_end is None else row_end, box, beta_map, policy, eps, sigma_seed,
                n_threads, 0, n_random, n_feat)


def pixel_stats(planes, desc):
    planes = np.ascontiguousarray(planes, np.float32)
    nf = dims(desc)[1]
    m = np.empty((desc.H, desc.W, nf))
    s = np.empty((desc.H, desc.W, nf))
    lib().rpf_oracle_pixel_stats(C.byref(desc), _p(planes), _p(m), _p(s))
    return m, s


def cf_weights(z, beta_map=BETA_REF_GCC11_O3, policy=DEGEN_REF_ABORT, eps=1e-10):
    z = _f64(z)
    a, b, w, m = np.empty(3), np.empty(12), np.empty(1), np.empty(NPAIR)
    lib().rpf_oracle_cf_weights(_p(z), z.shape[0], beta_map, policy, eps, _p(a), _p(b), _p(w), _p(m))
    return a, b, float(w[0]), m


def filter_pass(planes, desc, colour_in=None, debug=True):
    """planes: float32 [19,H,W,S]. Returns dict(colour=[3,H,W,S] f64, result fields, debug planes)."""
    planes = np.ascontiguousarray(planes, np.float32)   # (an fp16-stored buffer: its exact fp32 image)
    nd, nf, npair = dims(desc)
    assert planes.shape == (nd, desc.H, desc.W, desc.S), planes.shape
    H, W, S = desc.H, desc.W, desc.S
    cin = None if colour_in is None else _f64(colour_in)
    out = np.empty((3, H, W, S))
    res = Result()
    dbg = None
    d = {}
    if debug:
        d = dict(nbhd_size=np.zeros((H, W), np.int32), mean=np.zeros((H, W, nd)), stddev=np.zeros((H, W, nd)),
                 mi=np.zeros((H, W, npair)), alpha=np.zeros((H, W, 3)), beta=np.zeros((H, W, nf)),
                 wrc=np.zeros((H, W)), bin_hash=np.zeros((H, W, nd), np.uint32),
                 member_hash=np.zeros((H, W), np.uint32))
        dbg = Debug(*[_p(d[k]) for k, _ in Debug._fields_])
    lib().rpf_oracle_filter_pass(C.byref(desc), _p(planes), _p(cin), _p(out),
                                 C.byref(dbg) if dbg is not None else None, C.byref(res))
    d.update(colour=out, status=res.status, first_bad_pixel=res.first_bad_pixel,
             nonfinite_pixels=res.nonfinite_pixels, sum_nbhd=res.sum_nbhd, max_nbhd=res.max_nbhd)
    return d


def pixel_mean(colour, desc, ray_weight=None):
    colour = _f64(colour)
    rw = None if ray_weight is None else np.ascontiguousarray(ray_weight, np.float32)
    out = np.zeros((desc.H, desc.W, 3))
    lib().rpf_oracle_pixel_mean(C.byref(desc), _p(colour), _p(rw), _p(out))
    return out


def feature_images(planes, desc):
    """visualizeSF: six max-normalised per-pixel-mean images [6, H, W, 3] (n0, n1, p0, p1, pFilm, pLens)"""
    planes = np.ascontiguousarray(planes, np.float32)
    out = np.zeros((6, desc.H, desc.W, 3))
    lib().rpf_oracle_feature_images(C.byref(desc), _p(planes), _p(out))
    return out
