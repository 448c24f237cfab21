"""Per-sample feature buffers: layout helpers and the seeded synthetic generator.

Layout (host and HBM, identical): 19 SoA planes of fp32, plane ``d`` holds ``[y][x][s]``; the dims are
the reference's ``SampleData`` vector (/root/reference/src/custom/sd.h:62-94):

    0,1 pFilm | 2,3,4 L rgb | 5,6 pLens (random parameters) | 7..9 n0 | 10..12 p0 | 13..15 n1 | 16..18 p1

The reference container is ``samples[x][y][s]`` of 19 doubles (sample_film.cpp:32-42); values are
fp32-valued because pbrt's ``Float`` is ``float`` (core/pbrt.h), so fp32 planes hold them exactly.

The generator is counter-based (a 32-bit integer hash of (seed, stream, global sample index)), written
against a tiny array-backend shim so the SAME code runs on numpy (tests, here) and on torch tensors on
the GPU (bench, where 1080p..4K buffers are produced directly in HBM).  Pixel statistics do not depend
on the image size or on which row slab a rank generates (absolute pixel coordinates are used), which is
what makes the multi-GPU weak-scaling workload well defined.
"""
import math

import numpy as np

NDIM = 19
P0, C0, R0, F0 = 0, 2, 5, 7  # column groups


class _NP:
    """numpy backend shim"""
    i64 = np.int64
    f32 = np.float32
    f64 = np.float64
    f16 = np.float16

    @staticmethod
    def empty(shape, dtype):
        return np.empty(shape, dtype)

    @staticmethod
    def arange(n):
        return np.arange(n, dtype=np.int64)

    floor, sqrt, log, cos, sin, where, stack = np.floor, np.sqrt, np.log, np.cos, np.sin, np.where, np.stack

    @staticmethod
    def cast(a, dt):
        return a.astype(dt)


class _TORCH:
    """torch backend shim (device chosen at construction)"""

    def __init__(self, device):
        import torch
        self.t = torch
        self.device = device
        self.i64, self.f32, self.f64, self.f16 = torch.int64, torch.float32, torch.float64, torch.float16
        self.floor, self.sqrt, self.log, self.cos, self.sin, self.where = (
            torch.floor, torch.sqrt, torch.log, torch.cos, torch.sin, torch.where)

    def arange(self, n):
        return self.t.arange(n, dtype=self.t.int64, device=self.device)

    def stack(self, xs, axis=0):
        return self.t.stack(xs, dim=axis)

    def empty(self, shape, dtype):
        return self.t.empty(shape, dtype=dtype, device=self.device)

    @staticmethod
    def cast(a, dt):
        return a.to(dt)


_M32 = 0xFFFFFFFF


def _hash32(xp, x):
    """lowbias32 integer hash evaluated in int64 lanes (identical on numpy and torch)."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _uniform(xp, idx, seed, stream):
    """U[0,1) with 24 random bits (exact in fp32) for global sample index ``idx``."""
    h = _hash32(xp, idx ^ ((seed * 0x9E3779B1 + stream * 0x85EBCA6B) & _M32))
    h = _hash32(xp, h + (idx >> 32) + stream)
    return xp.cast(h >> 8, xp.f64) * (1.0 / 16777216.0)


def _gauss(xp, idx, seed, stream):
    u1 = _uniform(xp, idx, seed, stream)
    u2 = _uniform(xp, idx, seed, stream + 101)
    return xp.sqrt(-2.0 * xp.log(u1 + 1.0 / 33554432.0)) * xp.cos(2.0 * math.pi * u2)


def synth_planes(W, H, S, seed=20250103, sigma_f=0.05, sigma_c=1e-4, row0=0, xp=None, lens_colour=True,
                 mode="smooth", n_random=2, n_feat=12, dtype="f32", flat_frac=0.0):
    """Synthetic feature buffer, planes [5 + n_random + n_feat, H, W, S] (SURVEY.md section 8d recipe); fp32, or fp16
    with dtype="f16" (BASELINE configs[4]: fp16 feature storage).  The defaults give the reference's 19 dims.  Wider
    layouts append: random parameters r3, r4, ... (further sample dimensions, e.g. the light sample) and features
    13.. in groups of six -- three texture / albedo channels smooth in pFilm and three light-direction components that
    are functions of (r3, r4) -- each with the Gaussian jitter sigma_f (clustered mode: per-sample modes instead).

    ``row0`` is the image row of buffer row 0 (slab generation: a rank owning rows [a,b) of a taller image
    passes row0=a and H=b-a and gets exactly the rows the single-buffer call would produce).
    mode="smooth": first-hit features smooth in pFilm + Gaussian jitter sigma_f, second-hit features
    continuous functions of the random parameters.  Large neighbourhoods, near-identity filter output
    (the reference's sigma^2 = 0.002^2/(1-W_r_c)^2 kills every cross weight on generic data):
    the throughput workload.
    mode="clustered": every feature is a per-sample MODE (chosen by which half of the lens square the
    random parameters fall in: a defocus / two-surface edge inside every pixel) plus jitter sigma_f, and
    the colour is a mode value plus noise sigma_c.  Same-mode samples are near-identical in normalised
    space, so the cross-bilateral weights are non-trivial and the filter measurably changes the colours:
    the parity-fixture workload.
    ``flat_frac`` > 0: that fraction of the pixels (chosen by a hash of the pixel index) are "flat quads": their first-hit
    normal is the same three values in all S samples, the smooth field at the pixel centre.  A zero-variance feature is what
    93.9 % of the pixels of a captured pbrt buffer have, and it makes the strict 3-sigma test reject every
    neighbour (|f - m| >= 0 always): N = S exactly for those pixels -- the stand-in for a path-traced buffer (needs the
    EPS policy, as the captured buffers do: the reference aborts on them).
    """
    xp = xp or _NP
    n = H * W * S
    lin = xp.arange(n)
    s = lin % S
    px = (lin // S) % W
    py = (lin // (S * W)) + row0
    gidx = (py * W + px) * S + s  # global sample index: slab independent

    def U(k):
        return _uniform(xp, gidx, seed, k)

    def G(k):
        return _gauss(xp, gidx, seed, k)

    u1, u2, r1, r2 = U(1), U(2), U(3), U(4)
    fx = xp.cast(px, xp.f64) + u1
    fy = xp.cast(py, xp.f64) + u2
    X, Y = fx * 0.01, fy * 0.01

    # first-hit features: smooth in pFilm + Gaussian jitter
    nx = 0.35 * xp.sin(3.0 * X) + sigma_f * G(10)
    ny = 0.35 * xp.cos(2.0 * Y) + sigma_f * G(11)
    nz = math.sqrt(1.0 - 0.35 * 0.35 * 2.0 * 0.5) + 0.05 * xp.sin(X + Y) + sigma_f * G(12)
    step = xp.cast((xp.cast(xp.floor(fx / 256.0), xp.i64) % 2), xp.f64)
    p0x = 0.01 * fx + sigma_f * G(13)
    p0y = 0.01 * fy + sigma_f * G(14)
    p0z = 0.05 * xp.sin(3.0 * X) + 2.0 * step + sigma_f * G(15)
    # second-hit features: functions of the random parameters
    th = 2.0 * math.pi * r1
    cz = 2.0 * r2 - 1.0
    sr = xp.sqrt(1.0 - cz * cz + 1e-12)
    n1x = sr * xp.cos(th) + sigma_f * G(16)
    n1y = sr * xp.sin(th) + sigma_f * G(17)
    n1z = cz + sigma_f * G(18)
    p1x = p0x + 2.0 * n1x + sigma_f * G(19)
    p1y = p0y + 2.0 * n1y + sigma_f * G(20)
    p1z = p0z + 2.0 * n1z + sigma_f * G(21)
    # colour: region albedo x lens-dependent visibility x slow shading + jitter
    region = xp.cast(xp.cast(xp.floor(fx / 64.0), xp.i64) + xp.cast(xp.floor(fy / 64.0), xp.i64), xp.i64) % 3
    shade = 0.75 + 0.25 * xp.sin(X * 2.0) * xp.cos(Y * 2.0)
    if lens_colour:
        vis = xp.where(r1 < 0.5, 1.0 + 0.0 * r1, 0.15 + 0.0 * r1)
    else:
        vis = 1.0 + 0.0 * r1
    alb = [0.8 - 0.25 * xp.cast(region == k, xp.f64) for k in range(3)]
    cr = alb[0] * vis * shade + sigma_c * G(30)
    cg = alb[1] * vis * shade + sigma_c * G(31)
    cb = alb[2] * vis * shade + sigma_c * G(32)

    if mode == "clustered":
        ma = xp.cast(r2 < 0.5, xp.f64)  # first-hit mode (foreground / background through the lens)
        mb = xp.cast(r1 < 0.5, xp.f64)  # second-hit mode
        slow = 0.02 * xp.sin(X * 2.0 + Y)
        nx = 0.6 * ma - 0.3 + slow + sigma_f * G(10)
        ny = 0.5 - 0.7 * ma + slow + sigma_f * G(11)
        nz = 0.8 - 0.2 * ma + slow + sigma_f * G(12)
        p0x = 0.002 * fx + 1.5 * ma + sigma_f * G(13)
        p0y = 0.002 * fy - 0.8 * ma + sigma_f * G(14)
        p0z = 3.0 * ma + slow + sigma_f * G(15)
        n1x = 0.7 * mb - 0.2 + slow + sigma_f * G(16)
        n1y = 0.1 + 0.5 * mb + slow + sigma_f * G(17)
        n1z = 0.9 - 0.6 * mb + slow + sigma_f * G(18)
        p1x = p0x + 2.0 * mb + sigma_f * G(19)
        p1y = p0y - 1.0 * mb + sigma_f * G(20)
        p1z = p0z + 0.5 * mb + sigma_f * G(21)
        vis = 0.15 + 0.85 * mb
        cr = alb[0] * vis * shade + sigma_c * G(30)
        cg = alb[1] * vis * shade + sigma_c * G(31)
        cb = alb[2] * vis * shade + sigma_c * G(32)
    elif mode != "smooth":
        raise ValueError("mode must be 'smooth' or 'clustered'")

    if flat_frac > 0.0:
        pidx = py * W + px  # global pixel index: slab independent
        flat = _uniform(xp, pidx, seed, 77) < flat_frac
        Xc, Yc = (xp.cast(px, xp.f64) + 0.5) * 0.01, (xp.cast(py, xp.f64) + 0.5) * 0.01
        if mode == "clustered":
            fnx, fny, fnz = 0.3 + 0.0 * Xc, -0.2 + 0.0 * Xc, 0.6 + 0.02 * xp.sin(Xc * 2.0 + Yc)
        else:
            fnx, fny = 0.35 * xp.sin(3.0 * Xc), 0.35 * xp.cos(2.0 * Yc)
            fnz = math.sqrt(1.0 - 0.35 * 0.35 * 2.0 * 0.5) + 0.05 * xp.sin(Xc + Yc)
        nx, ny, nz = xp.where(flat, fnx, nx), xp.where(flat, fny, ny), xp.where(flat, fnz, nz)

    rand = [r1, r2] + [U(5 + k) for k in range(n_random - 2)]
    feats = [nx, ny, nz, p0x, p0y, p0z, n1x, n1y, n1z, p1x, p1y, p1z]
    if n_feat > 12:
        r3 = rand[2] if n_random > 2 else U(5)
        r4 = rand[3] if n_random > 3 else U(6)
        mc = xp.cast(r3 < 0.5, xp.f64)
        th2, sr2 = 2.0 * math.pi * r3, xp.sqrt(r4)
        for k in range(n_feat - 12):
            g, j = divmod(k, 6)
            if mode == "clustered":
                base = (0.4 + 0.1 * j + 0.3 * ma + slow) if j < 3 else (0.2 * (j - 2) + 0.5 * mc + slow)
            elif j < 3:
                base = 0.5 + 0.3 * xp.sin((j + 2.0 + g) * X + (j + 0.5) * Y)
            else:
                base = (sr2 * xp.cos(th2), sr2 * xp.sin(th2), xp.sqrt(1.0 - r4 + 1e-12))[j - 3] * (1.0 + 0.25 * g)
            feats.append(base + sigma_f * G(40 + k))
    cols = [fx, fy, cr, cg, cb] + rand[:n_random] + feats[:n_feat]
    out_dt = {"f32": xp.f32, "f16": xp.f16}[dtype]
    # (fp16: through fp32 on every backend, so numpy and torch round identically)
    planes = xp.stack([xp.cast(xp.cast(c, xp.f32), out_dt) for c in cols], 0)
    return planes.reshape(5 + n_random + n_feat, H, W, S)


def synth_planes_chunked(W, H, S, rows_per_chunk=64, row0=0, xp=None, **kw):
    """synth_planes() for buffers too large to generate in one piece (the generator's fp64 temporaries are ~60x the
    output): row chunks written into one preallocated buffer -- same values as the single call (global sample
    indices).  This is how the 8192-wide x 64 spp slabs of BASELINE configs[4] are produced on the device."""
    xp = xp or _NP
    first = synth_planes(W, min(rows_per_chunk, H), S, row0=row0, xp=xp, **kw)
    if H <= rows_per_chunk:
        return first
    out = xp.empty((first.shape[0], H, W, S), first.dtype)
    out[:, :first.shape[1]] = first
    del first
    r = rows_per_chunk
    while r < H:
        n = min(rows_per_chunk, H - r)
        out[:, r:r + n] = synth_planes(W, n, S, row0=row0 + r, xp=xp, **kw)
        r += n
    return out


def torch_backend(device):
    return _TORCH(device)


# ---- AoS <-> SoA (the reference container order is samples[x][y][s][19] doubles) -------------------
def aos_to_planes(aos):
    """aos: float64 [W][H][S][19] (SamplingFilm order, sample_film.cpp:32-42) -> planes f32 [19,H,W,S]."""
    aos = np.asarray(aos)
    return np.ascontiguousarray(np.transpose(aos, (3, 1, 0, 2)).astype(np.float32))


def planes_to_aos(planes):
    planes = np.asarray(planes)
    return np.ascontiguousarray(np.transpose(planes, (2, 1, 3, 0)).astype(np.float64))


def halo_rows(box):
    """rows of neighbouring slabs a row-tiled rank needs (rpf.cpp:561: b = (box-1)/2)."""
    return (box - 1) // 2


# ---- on-disk feature buffers (".rpfb") -------------------------------------------------------------------
# The reference has no file format for its per-sample buffers (they live on Render()'s stack, rpf.cpp:745).
# This one makes the bench/test workloads and captured pbrt buffers replayable:
#   64-byte little-endian header: magic "RPFB", u32 version (1), u32 W, u32 H, u32 S, u32 ndim (19),
#   u32 dtype (0 = fp32), u32 flags (bit 0: a ray-weight plane follows), 8 reserved u32
#   then planes [ndim][H][W][S] fp32, then (flag bit 0) ray_weight [H][W][S] fp32.
import struct

RPFB_MAGIC = b"RPFB"


def save_rpfb(path, planes, ray_weight=None):
    planes = np.ascontiguousarray(planes, np.float32)
    nd, H, W, S = planes.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<4s7I8I", RPFB_MAGIC, 1, W, H, S, nd, 0, 1 if ray_weight is not None else 0, *([0] * 8)))
        f.write(planes.tobytes())
        if ray_weight is not None:
            rw = np.ascontiguousarray(ray_weight, np.float32)
            assert rw.shape == (H, W, S)
            f.write(rw.tobytes())


def load_rpfb(path, mmap=False):
    """returns (planes [ndim,H,W,S] f32, ray_weight [H,W,S] f32 or None); mmap=True maps large files lazily"""
    with open(path, "rb") as f:
        hdr = f.read(64)
    magic, ver, W, H, S, nd, dtype, flags = struct.unpack("<4s7I", hdr[:32])
    if magic != RPFB_MAGIC or ver != 1 or dtype != 0:
        raise ValueError("%s: not an RPFB v1 fp32 file" % path)
    n = nd * H * W * S
    if mmap:
        planes = np.memmap(path, np.float32, "r", 64, (nd, H, W, S))
        rw = np.memmap(path, np.float32, "r", 64 + 4 * n, (H, W, S)) if flags & 1 else None
    else:
        data = np.fromfile(path, np.float32, offset=64)
        planes = data[:n].reshape(nd, H, W, S)
        rw = data[n:n + H * W * S].reshape(H, W, S) if flags & 1 else None
    return planes, rw
