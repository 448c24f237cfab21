"""ctypes binding of the C ABI declared in include/rpf_hip.h (librpf_hip.so).

This is the ONLY compute path of the package: if the library is missing the import of this module raises
(no CPU fallback, no oracle).  Host-buffer entry points take numpy arrays; device entry points take raw
device pointers (e.g. torch tensors' ``data_ptr()``), torch is used by callers only for HBM allocation,
streams and torch.distributed.
"""
import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RPF_HIP_LIB: load another build of the same ABI (profiling variants, scripts/ablate_mi.sh); default = the in-tree library
LIB_PATH = os.environ.get("RPF_HIP_LIB") or os.path.join(_HERE, "lib", "librpf_hip.so")

NDIM, NFEAT, NPAIR, MAX_BOXES = 19, 12, 96, 8
OK, E_BADARG, E_HIP, E_NONFINITE, E_NOMEM, E_UNSUPPORTED, E_NODEVICE = range(7)
BETA_REF_GCC11_O3, BETA_REF_GCC11_O2, BETA_PAPER = 0, 1, 2
DEGEN_REF_ABORT, DEGEN_EPS = 0, 1
FLAG_TIMING = 1
FLAG_FAST_WEIGHTS = 2
FLAG_NO_OVERLAP = 4
PLANES_F32, PLANES_F16 = 0, 1

EXPORTS = ["rpf_version", "rpf_status_string", "rpf_create", "rpf_destroy", "rpf_last_error", "rpf_filter",
           "rpf_filter_device", "rpf_colour_from_planes_device", "rpf_reduce_device", "rpf_stage_pixel_stats",
           "rpf_filter_pass_debug", "rpf_query_counters", "rpf_lds_bytes_required", "rpf_selftest_udiv", "rpf_feature_images",
           "rpf_host_alloc", "rpf_host_free", "rpf_filter_ex", "rpf_set_option", "rpf_multi_create", "rpf_multi_destroy",
           "rpf_multi_last_error", "rpf_multi_device_count", "rpf_multi_set_option", "rpf_multi_filter",
           "rpf_multi_query_counters", "rpf_query_nbhd", "rpf_query_route"]


class Desc(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("S", C.c_int32), ("row_begin", C.c_int32),
                ("row_end", C.c_int32), ("n_box", C.c_int32), ("box_sizes", C.c_int32 * MAX_BOXES),
                ("beta_map", C.c_int32), ("degenerate_policy", C.c_int32), ("flags", C.c_int32),
                ("eps", C.c_double), ("sigma_seed", C.c_double), ("n_random", C.c_int32), ("n_feat", C.c_int32),
                ("plane_dtype", C.c_int32), ("reserved", C.c_int32)]


def dims(desc):
    """(ndim, nfeat, npair, numpy plane dtype) of a descriptor's sample layout (0 fields = the reference's)"""
    nr, nf = desc.n_random or 2, desc.n_feat or 12
    return 5 + nr + nf, nf, nf * (nr + 2) + 3 * (nr + 2 + nf), (np.float16 if desc.plane_dtype == PLANES_F16 else np.float32)


class Debug(C.Structure):
    _fields_ = [("nbhd_size", C.c_void_p), ("mean", C.c_void_p), ("stddev", C.c_void_p), ("mi", C.c_void_p),
                ("alpha", C.c_void_p), ("beta", C.c_void_p), ("wrc", C.c_void_p), ("bin_hash", C.c_void_p),
                ("member_hash", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [("samples_filtered", C.c_int64), ("sum_nbhd", C.c_int64), ("nonfinite_pixels", C.c_int64),
                ("max_nbhd", C.c_int32), ("first_bad_pixel", C.c_int32), ("filter_kernel_ms", C.c_float),
                ("stats_kernel_ms", C.c_float), ("device_total_ms", C.c_float), ("h2d_ms", C.c_float),
                ("d2h_ms", C.c_float), ("filter_kernel_launches", C.c_int32), ("options_active", C.c_int32),
                ("redo_pixels", C.c_int32)]


class RpfError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (status_string(status), message))
        self.status = status


_lib = None


def load():
    """dlopen librpf_hip.so; raises OSError when it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("librpf_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`. "
                          "There is no CPU fallback." % LIB_PATH)
        try:
            # PyTorch wheels bundle their own HIP runtime; when torch is used in the same process (HBM tensors,
            # streams, torch.distributed) it must be the first to load libamdhip64 so that both sides share ONE
            # runtime -- loading ours first leaves torch unable to see the GPU.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.rpf_version.restype = C.c_char_p
        L.rpf_status_string.restype = C.c_char_p
        L.rpf_status_string.argtypes = [C.c_int32]
        L.rpf_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
        L.rpf_destroy.argtypes = [C.c_void_p]
        L.rpf_destroy.restype = None
        L.rpf_last_error.restype = C.c_char_p
        L.rpf_last_error.argtypes = [C.c_void_p]
        L.rpf_filter.argtypes = [C.c_void_p, C.POINTER(Desc)] + [C.c_void_p] * 4
        L.rpf_filter_ex.argtypes = [C.c_void_p, C.POINTER(Desc)] + [C.c_void_p] * 6
        L.rpf_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.rpf_filter_device.argtypes = [C.c_void_p, C.POINTER(Desc)] + [C.c_void_p] * 3
        L.rpf_colour_from_planes_device.argtypes = [C.c_void_p, C.POINTER(Desc)] + [C.c_void_p] * 3
        L.rpf_reduce_device.argtypes = [C.c_void_p, C.POINTER(Desc)] + [C.c_void_p] * 5
        L.rpf_stage_pixel_stats.argtypes = [C.c_void_p, C.POINTER(Desc)] + [C.c_void_p] * 3
        L.rpf_filter_pass_debug.argtypes = [C.c_void_p, C.POINTER(Desc), C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.POINTER(Debug)]
        L.rpf_query_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.rpf_query_nbhd.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.rpf_query_route.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.rpf_selftest_udiv.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_uint64)]
        L.rpf_feature_images.argtypes = [C.c_void_p, C.POINTER(Desc), C.c_void_p, C.c_void_p]
        L.rpf_lds_bytes_required.restype = C.c_int64
        L.rpf_lds_bytes_required.argtypes = [C.c_int32, C.c_int32]
        L.rpf_multi_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32]
        L.rpf_multi_destroy.argtypes = [C.c_void_p]
        L.rpf_multi_destroy.restype = None
        L.rpf_multi_last_error.restype = C.c_char_p
        L.rpf_multi_last_error.argtypes = [C.c_void_p]
        L.rpf_multi_device_count.argtypes = [C.c_void_p]
        L.rpf_multi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.rpf_multi_filter.argtypes = [C.c_void_p, C.POINTER(Desc)] + [C.c_void_p] * 4
        L.rpf_multi_query_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.rpf_host_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.rpf_host_free.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def status_string(st):
    return load().rpf_status_string(int(st)).decode()


def make_desc(W, H, S, boxes=(7,), row_begin=0, row_end=None, beta_map=BETA_REF_GCC11_O3, policy=DEGEN_REF_ABORT,
              eps=1e-10, sigma_seed=0.002, flags=0, n_random=0, n_feat=0, plane_dtype=PLANES_F32):
    d = Desc()
    d.W, d.H, d.S = W, H, S
    d.row_begin, d.row_end = row_begin, (H if row_end is None else row_end)
    d.n_box = len(boxes)
    for i, b in enumerate(boxes[:MAX_BOXES]):
        d.box_sizes[i] = b
    d.beta_map, d.degenerate_policy, d.flags = beta_map, policy, flags
    d.eps, d.sigma_seed = eps, sigma_seed
    d.n_random, d.n_feat, d.plane_dtype = n_random, n_feat, plane_dtype
    return d


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One rpf_ctx (one HIP device).  Mirrors how RPFIntegrator::Render drives ApplyRPFFilter."""

    def __init__(self, device=0):
        self._L = load()
        h = C.c_void_p()
        st = self._L.rpf_create(C.byref(h), device)
        self._h = h
        if st != OK:
            msg = self._L.rpf_last_error(h).decode() if h else "no HIP device"
            if h:
                self._L.rpf_destroy(h)
                self._h = None
            raise RpfError(st, msg)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rpf_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st, allow=()):
        if st != OK and st not in allow:
            raise RpfError(st, self._L.rpf_last_error(self._h).decode())
        return st

    def set_option(self, name, value):
        """per-context tuning / diagnostic override (rpf_set_option); see include/rpf_hip.h for the names"""
        self._check(self._L.rpf_set_option(self._h, name.encode(), int(value)))

    def counters(self):
        c = Counters()
        self._check(self._L.rpf_query_counters(self._h, C.byref(c)))
        return c

    def nbhd(self, W, H):
        """neighbourhood size of every pixel as the last pass of the most recent call left it (rpf_query_nbhd)"""
        out = np.empty((H, W), np.int32)
        self._check(self._L.rpf_query_nbhd(self._h, _p(out), W * H))
        return out

    def route(self):
        """kernel route of the last pass: 0 fused, 1 count first, 2 size-binned (rpf_query_route)"""
        r = C.c_int32(-1)
        self._check(self._L.rpf_query_route(self._h, C.byref(r)))
        return r.value

    # ---- host-buffer entry points ------------------------------------------------------------------
    def host_empty(self, shape, dtype=np.float32):
        """numpy array in page-locked host memory (rpf_host_alloc); released when the last view of it is collected."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        ptr = C.c_void_p()
        self._check(self._L.rpf_host_alloc(self._h, n, C.byref(ptr)))
        buf = (C.c_char * max(n, 1)).from_address(ptr.value)
        weakref.finalize(buf, self._L.rpf_host_free, None, ptr.value)  # released when the last view dies
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def filter(self, planes, desc, ray_weight=None, want_samples=True, want_pixels=True, allow_nonfinite=False,
               out_samples=None, out_pixels=None, colour64_in=None, want_colour64=False):
        nd, _, _, pdt = dims(desc)
        planes = np.ascontiguousarray(planes, pdt)
        assert planes.shape == (nd, desc.H, desc.W, desc.S), planes.shape
        rw = None if ray_weight is None else np.ascontiguousarray(ray_weight, np.float32)
        srgb = out_samples if out_samples is not None else (
            np.empty((3, desc.H, desc.W, desc.S), np.float32) if want_samples else None)
        prgb = out_pixels if out_pixels is not None else (np.empty((desc.H, desc.W, 3), np.float32) if want_pixels else None)
        if colour64_in is not None or want_colour64:
            cin = None if colour64_in is None else np.ascontiguousarray(colour64_in, np.float64)
            c64 = np.empty((3, desc.H, desc.W, desc.S)) if want_colour64 else None
            st = self._L.rpf_filter_ex(self._h, C.byref(desc), _p(planes), _p(cin), _p(rw), _p(srgb), _p(prgb), _p(c64))
            self._check(st, allow=(E_NONFINITE,) if allow_nonfinite else ())
            return srgb, prgb, st, c64
        st = self._L.rpf_filter(self._h, C.byref(desc), _p(planes), _p(rw), _p(srgb), _p(prgb))
        self._check(st, allow=(E_NONFINITE,) if allow_nonfinite else ())
        return srgb, prgb, st

    def pixel_stats(self, planes, desc):
        _, nf, _, pdt = dims(desc)
        planes = np.ascontiguousarray(planes, pdt)
        m = np.empty((desc.H, desc.W, nf))
        s = np.empty((desc.H, desc.W, nf))
        self._check(self._L.rpf_stage_pixel_stats(self._h, C.byref(desc), _p(planes), _p(m), _p(s)))
        return m, s

    def filter_pass_debug(self, planes, desc, box=7, colour_in=None, debug=True, allow_nonfinite=False):
        nd, nf, npair, pdt = dims(desc)
        planes = np.ascontiguousarray(planes, pdt)
        assert planes.shape == (nd, desc.H, desc.W, desc.S), planes.shape
        H, W, S = desc.H, desc.W, desc.S
        cin = None if colour_in is None else np.ascontiguousarray(colour_in, np.float64)
        out = np.empty((3, H, W, S))
        d, dbg = {}, None
        if debug:
            d = dict(nbhd_size=np.zeros((H, W), np.int32), mean=np.zeros((H, W, nd)), stddev=np.zeros((H, W, nd)),
                     mi=np.zeros((H, W, npair)), alpha=np.zeros((H, W, 3)), beta=np.zeros((H, W, nf)),
                     wrc=np.zeros((H, W)), bin_hash=np.zeros((H, W, nd), np.uint32),
                     member_hash=np.zeros((H, W), np.uint32))
            dbg = Debug(*[_p(d[k]) for k, _ in Debug._fields_])
        st = self._L.rpf_filter_pass_debug(self._h, C.byref(desc), box, _p(planes), _p(cin), _p(out),
                                           C.byref(dbg) if dbg is not None else None)
        self._check(st, allow=(E_NONFINITE,) if allow_nonfinite else ())
        c = self.counters()
        d.update(colour=out, status=st, nonfinite_pixels=c.nonfinite_pixels, first_bad_pixel=c.first_bad_pixel,
                 sum_nbhd=c.sum_nbhd, max_nbhd=c.max_nbhd, filter_kernel_ms=c.filter_kernel_ms)
        return d

    def feature_images(self, planes, desc):
        planes = np.ascontiguousarray(planes, np.float32)
        out = np.empty((6, desc.H, desc.W, 3))
        self._check(self._L.rpf_feature_images(self._h, C.byref(desc), _p(planes), _p(out)))
        return out

    def selftest_udiv(self, n, seed=1, mode=0):
        m = C.c_uint64(0)
        self._check(self._L.rpf_selftest_udiv(self._h, n, seed, mode, C.byref(m)))
        return m.value

    # ---- device-resident entry points (raw device pointers) -----------------------------------------
    def colour_from_planes_device(self, desc, d_planes, d_colour, stream=None):
        self._check(self._L.rpf_colour_from_planes_device(self._h, C.byref(desc), d_planes, d_colour, stream))

    def filter_device(self, desc, d_planes, d_colour, stream=None, allow_nonfinite=False):
        st = self._L.rpf_filter_device(self._h, C.byref(desc), d_planes, d_colour, stream)
        return self._check(st, allow=(E_NONFINITE,) if allow_nonfinite else ())

    def reduce_device(self, desc, d_colour, d_ray_weight, d_sample_rgb, d_pixel_rgb, stream=None):
        self._check(self._L.rpf_reduce_device(self._h, C.byref(desc), d_colour, d_ray_weight, d_sample_rgb,
                                              d_pixel_rgb, stream))


class MultiContext:
    """rpf_multi: one caller, one row slab per entry of ``devices`` (None = every visible GPU; an ordinal may repeat)"""

    def __init__(self, devices=None):
        self._L = load()
        h = C.c_void_p()
        arr = None if devices is None else (C.c_int32 * len(devices))(*devices)
        st = self._L.rpf_multi_create(C.byref(h), arr, 0 if devices is None else len(devices))
        self._h = h
        if st != OK:
            msg = self._L.rpf_multi_last_error(h).decode() if h else "no HIP device"
            if h:
                self._L.rpf_multi_destroy(h)
                self._h = None
            raise RpfError(st, msg)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rpf_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def device_count(self):
        return self._L.rpf_multi_device_count(self._h)

    def counters(self):
        c = Counters()
        self._L.rpf_multi_query_counters(self._h, C.byref(c))
        return c

    def filter(self, planes, desc, ray_weight=None, allow_nonfinite=False):
        nd, _, _, pdt = dims(desc)
        planes = np.ascontiguousarray(planes, pdt)
        assert planes.shape == (nd, desc.H, desc.W, desc.S), planes.shape
        rw = None if ray_weight is None else np.ascontiguousarray(ray_weight, np.float32)
        srgb = np.empty((3, desc.H, desc.W, desc.S), np.float32)
        prgb = np.empty((desc.H, desc.W, 3), np.float32)
        st = self._L.rpf_multi_filter(self._h, C.byref(desc), _p(planes), _p(rw), _p(srgb), _p(prgb))
        if st != OK and not (allow_nonfinite and st == E_NONFINITE):
            raise RpfError(st, self._L.rpf_multi_last_error(self._h).decode())
        return srgb, prgb, st


def lds_bytes_required(S, box):
    return load().rpf_lds_bytes_required(S, box)
