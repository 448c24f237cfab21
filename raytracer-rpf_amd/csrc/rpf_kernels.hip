// rpf_kernels.hip -- hand-written gfx950 kernels of the RPF pass.  Compiled with -ffp-contract=off: every
// fp64 operation whose rounding decides a DISCRETE outcome (3-sigma membership, histogram bin) is issued
// in the reference's operation order, so those outcomes are bit-identical to the CPU path; fused
// multiply-adds appear only where written as fma().
//
// Kernel map (reference lines -> kernel):
//   rpf.cpp:302-353 FillMeanAndStddev                       -> pixel_stats_kernel      (thread / pixel)
//   rpf.cpp:556-717 gather, normalise, ComputeCFWeights,
//                   weights, blend   + mi.cpp:5-90          -> filter_pixel_kernel     (one or four wave64 / pixel)
//   (box*box*S > 512 only) N per pixel, size classes         -> nbhd_count_kernel, classify_kernel
//   rpf.cpp:779-794 per-pixel reduction (box r=0.5)         -> reduce_kernel
//   rpf.cpp:37-101  visualizeSF + vis.cpp:34-51             -> feature_mean_kernel, feature_normalise_kernel
//
// filter_pixel_kernel<K, ., ., NW>: NW = 1 -- one 64-lane workgroup (= one wavefront) per pixel, 12 resident per CU at
// K = 7 (160 VGPRs, 12 KiB LDS); NW = 4 (K = 25, 49) -- four waves share a large neighbourhood's LDS and split columns /
// histogram groups / own samples.  No inter-workgroup communication; DESIGN.md section 4 walks through the stages:
//   1b  candidates of the box window are tested 64 at a time in the reference's visiting order with their gathers
//       three steps ahead; ballot + prefix popcount append accepted offsets to an LDS list (list order ==
//       reference neighbourhood order)
//   2   in-order (reference-order) sum / sum-of-squares chains per column through an LDS staging buffer; the
//       column min/max of x ride along
//   3a  bins_stage: z = (x-M)/SD and t = (z-lo)/(hi-lo)*B by exact division with a hoisted reciprocal; bin ids
//       packed 5 bits per sample in LDS
//   3b  mi_stage / mi_stage_deep / mi_stage_tiny: 19 marginal + 96 joint histograms by LDS atomics; MI from a
//       2^-44 fixed-point k ln k table (exact integer sums, no log on the device)
//   3c  alpha, beta, W_r_c lane-parallel through an LDS scratch area
//   4   pair weights in z-space (A_i + B_j + u_i . z_j), fp64 (or fp32 with RPF_FLAG_FAST_WEIGHTS), LDS-free
//       transposed-butterfly reductions (rpf_xlane.h)
// Tuning / profiling knobs are per-context options (rpf_set_option -> struct Tuning): stage_mask (skip stages; results
// wrong), lds_pad (lower occupancy), table_in_lds, waves_per_pixel, binning.  Nothing is read from the environment.
#include "rpf_device_common.h"

namespace rpf {

namespace {

struct ClassCaps { int cap[kNumClasses]; };

// pixels in slab_pixel order -> one list per class (class = first capacity >= N); lists[c][*], counts[c]
__global__ __launch_bounds__(256) void classify_kernel(PassParams p, ClassCaps caps, uint32_t *lists, uint32_t *counts,
                                                       uint64_t list_stride, int max_class, int rest_class) {
    const int lane = threadIdx.x & (kWave - 1);
    const int rows_band = (p.row_end - p.row_begin + 7) / 8;
    const int64_t per_band = (int64_t)rows_band * p.W;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int x = 0, y = 0, cls = -1;
    uint32_t pix = 0;
    if (t < 8 * per_band && slab_pixel(p, (int)(t / per_band), t % per_band, x, y)) {
        pix = (uint32_t)y * (uint32_t)p.W + (uint32_t)x;
        const int n = p.nbhd[pix];
        cls = kNumClasses - 1;
#pragma unroll
        for (int c = kNumClasses - 2; c >= 0; --c)
            if (n <= caps.cap[c]) cls = c;
        if (cls >= max_class) cls = rest_class; // (-1: not listed)
    }
    // one atomic per class and WORKGROUP (a per-wave append serialises on the class counter: 1.1 ms per 1080p frame when two
    // classes share the frame); positions inside the workgroup keep the slab order
    __shared__ uint32_t sCnt[4][kNumClasses], sBase[kNumClasses];
    const int wv = threadIdx.x >> 6;
    unsigned long long mine = 0ull;
#pragma unroll
    for (int c = 0; c < kNumClasses; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (lane == 0) sCnt[wv][c] = (uint32_t)__popcll(m);
        if (cls == c) mine = m;
    }
    __syncthreads();
    if (threadIdx.x < kNumClasses) {
        const int c = threadIdx.x;
        const uint32_t tot = sCnt[0][c] + sCnt[1][c] + sCnt[2][c] + sCnt[3][c];
        sBase[c] = tot ? atomicAdd(&counts[c], tot) : 0u;
    }
    __syncthreads();
    if (cls >= 0) {
        uint32_t at = sBase[cls] + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; ++w) at += sCnt[w][cls];
        lists[(uint64_t)cls * list_stride + at] = pix;
    }
}

// pixels in slab_pixel order -> the list of those whose neighbourhood is not proven to be the own samples (PassParams::flat)
__global__ __launch_bounds__(256) void prelist_kernel(PassParams p, uint32_t *list, uint32_t *count) {
    __shared__ uint32_t sBase, sWaveCnt[4];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const int rows_band = (p.row_end - p.row_begin + 7) / 8;
    const int64_t per_band = (int64_t)rows_band * p.W;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int x = 0, y = 0;
    bool general = false;
    uint32_t pix = 0;
    if (t < 8 * per_band && slab_pixel(p, (int)(t / per_band), t % per_band, x, y)) {
        pix = (uint32_t)y * (uint32_t)p.W + (uint32_t)x;
        general = !(p.flat[pix] != 0 && *p.nan_flag == 0);
        if (!general) p.nbhd[pix] = p.S; // rpf.cpp:556-586 with every candidate rejected: the own samples
    }
    const unsigned long long m = __ballot(general);
    if (lane == 0) sWaveCnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = sWaveCnt[0] + sWaveCnt[1] + sWaveCnt[2] + sWaveCnt[3];
        sBase = tot ? atomicAdd(count, tot) : 0u; // one atomic per workgroup: a per-wave append on one counter serialises
    }
    __syncthreads();
    if (general) {
        uint32_t at = sBase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; ++w) at += sWaveCnt[w];
        list[at] = pix;
    }
}

// self-test of udiv(): bitwise comparison with the compiler's IEEE division on pseudo-random operands
// drawn from the magnitudes stage 3a sees (and a band of extreme ones that must take the fallback)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void udiv_selftest_kernel(uint64_t n, uint64_t seed, int mode,
                                                             unsigned long long *mismatch) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0;
    for (; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t h1 = mix64(seed + 2 * i), h2 = mix64(seed + 2 * i + 1);
        // mantissas: random 52 bits; exponents: mode 0 -> [-40, 40], mode 1 -> [-600, 600] (hits the fallback)
        const int span = mode == 0 ? 81 : 1201;
        const int ea = (int)((h1 >> 52) % span) - span / 2, eb = (int)((h2 >> 52) % span) - span / 2;
        double a = ldexp(1.0 + (double)(h1 & 0xFFFFFFFFFFFFFull) * 0x1p-52, ea);
        double b = ldexp(1.0 + (double)(h2 & 0xFFFFFFFFFFFFFull) * 0x1p-52, eb);
        if (h1 & (1ull << 63)) a = -a;
        if ((h2 >> 60) == 0) a = (double)(float)a;          // fp32-valued numerators, as the feature planes are
        if ((h2 >> 60) == 1) a = b * (double)(int)(h1 % 41); // exact quotients (bin edges)
        if ((h2 >> 60) == 2) a = 0.0;
        const UDiv d = udiv_prepare(b);
        const double q = udiv(a, d), want = a / b;
        if (__double_as_longlong(q) != __double_as_longlong(want)) ++bad;
    }
    if (bad) atomicAdd(mismatch, bad);
}

// elements [e0, e0+cnt) of each of the three colour planes
template <class TP>
__global__ __launch_bounds__(256) void colour_from_planes_kernel(const TP *planes, double *colour, uint64_t ps,
                                                                  uint64_t e0, uint64_t cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * cnt) return;
    const uint64_t c = i / cnt, e = e0 + (i - c * cnt);
    colour[c * ps + e] = (double)(float)planes[(2 + c) * ps + e];
}

__global__ __launch_bounds__(256) void copy_f64_kernel(const double *src, double *dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void copy_colour_span_kernel(const double *src, double *dst, uint64_t ps, uint64_t e0,
                                                                uint64_t cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * cnt) return;
    const uint64_t c = i / cnt, e = e0 + (i - c * cnt);
    dst[c * ps + e] = src[c * ps + e];
}

// rpf.cpp:783-794 with the default box reconstruction filter: pixel = sum_s(L*rayWeight) / S
__global__ __launch_bounds__(256) void reduce_kernel(const double *colour, const float *ray_weight, float *sample_rgb,
                                                      float *pixel_rgb, int W, int H, int S, uint64_t pix0,
                                                      uint64_t pix1) {
    const uint64_t HW = (uint64_t)H * W;
    const uint64_t pix = pix0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= pix1) return;
    const uint64_t ps = HW * S;
    for (int c = 0; c < 3; ++c) {
        double acc = 0.0;
        for (int s = 0; s < S; ++s) {
            const uint64_t o = pix * S + s;
            const double v = colour[c * ps + o];
            if (sample_rgb) sample_rgb[c * ps + o] = (float)v;
            acc += v * (ray_weight ? (double)ray_weight[o] : 1.0);
        }
        if (pixel_rgb) pixel_rgb[pix * 3 + c] = (float)(acc / (double)S);
    }
}

// visualizeSF (rpf.cpp:37-101): per-pixel mean of a feature triple over the S samples (in-order fp64 sum) and the
// per-channel image maximum (vis.cpp:38-45; the maximum starts at 0, so only positive means can raise it and the
// bit pattern of a positive double orders like an unsigned integer)
__global__ __launch_bounds__(256) void feature_mean_kernel(const float *planes, uint64_t ps, uint64_t HW, int S,
                                                            double *out, unsigned long long *maxbits) {
    const int first_col[6] = {7, 13, 10, 16, 0, 5}, ncol[6] = {3, 3, 3, 3, 2, 2};
    const uint64_t pix = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int im = blockIdx.y;
    if (pix >= HW) return;
    for (int c = 0; c < 3; ++c) {
        double acc = 0.0;
        if (c < ncol[im]) {
            const float *src = planes + (uint64_t)(first_col[im] + c) * ps + pix * S;
            for (int s = 0; s < S; ++s) acc = acc + (double)src[s];
        }
        acc = acc / (double)S;
        out[((uint64_t)im * HW + pix) * 3 + c] = acc;
        if (acc > 0.0) atomicMax(&maxbits[im * 3 + c], (unsigned long long)__double_as_longlong(acc));
    }
}
__global__ __launch_bounds__(256) void feature_normalise_kernel(double *out, uint64_t HW, const unsigned long long *maxbits) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 6 * HW * 3) return;
    const int im = (int)(i / (HW * 3)), c = (int)(i % 3);
    const double mx = __longlong_as_double((long long)maxbits[im * 3 + c]);
    out[i] = (mx == 0.0) ? 0.0 : out[i] / mx; // vis.h:32-37
}

__global__ __launch_bounds__(256) void nbhd_reduce_kernel(const int32_t *nbhd, uint64_t begin, uint64_t end,
                                                           unsigned long long *out2) {
    uint64_t i = begin + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long s = 0;
    unsigned int mx = 0;
    for (; i < end; i += (uint64_t)gridDim.x * 256) {
        const unsigned int v = (unsigned int)nbhd[i];
        s += v;
        mx = max(mx, v);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        s += __shfl_xor(s, m, 64);
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out2[0], s);
        atomicMax(&out2[1], (unsigned long long)mx);
    }
}

// ---- the layout-dependent kernels live in their own translation units (rpf_impl_*.hip: rpf_filter_impl.inc once per
// sample-vector layout and size-class part, compiled in parallel); their entry points:
} // namespace
#define RPF_DECLARE_IMPL(NS)                                                                                              \
    namespace NS {                                                                                                        \
    hipError_t impl_filter_small(const PassParams &p, const LdsLayout &L, bool t_in_lds, unsigned grid, hipStream_t s);   \
    hipError_t impl_filter_mid(const PassParams &p, const LdsLayout &L, const LdsLayout &L2, const LdsLayout &L3, bool t_in_lds, unsigned grid, hipStream_t s); \
    hipError_t impl_filter_large(const PassParams &p, const LdsLayout &L, const LdsLayout &L2, const LdsLayout &L3, bool t_in_lds, unsigned grid, hipStream_t s); \
    hipError_t impl_pixel_stats(const PassParams &p, uint64_t pix0, uint64_t pix1, hipStream_t s);                        \
    hipError_t impl_nbhd_count(const PassParams &p, int step, uint32_t *probe, const uint32_t *list, const uint32_t *list_count, uint32_t list_max, hipStream_t s); \
    hipError_t impl_filter_big(const PassParams &p, void *list, void *bins, uint32_t slots, const uint32_t *count_dev, hipStream_t s); \
    hipError_t impl_filter_packed(const PassParams &p, int lanes_per_pixel, const uint32_t *count_dev, hipStream_t s);    \
    }
RPF_DECLARE_IMPL(d19) // the reference's 19 dims (2 random parameters, 12 features), fp32 planes
RPF_DECLARE_IMPL(d27) // BASELINE configs[4]: 27 dims (4 random parameters, 18 features), fp16 feature storage
#undef RPF_DECLARE_IMPL

int max_lds_per_block() { return 160 * 1024; }

static uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

int samples_per_lane(int nmax) {
    const int per_lane = (nmax + kWave - 1) / kWave;
    const int ks[] = {1, 2, 4, 7, 13, 25, 49};
    for (int k : ks)
        if (per_lane <= k) return k;
    return 0;
}

// waves per pixel: 4 once a neighbourhood is too large for more than a few one-wave workgroups to share a CU's
// LDS (K >= 25: 32 spp and up at box 7); option "waves_per_pixel" = 1 / 4 overrides for experiments (4 needs K >= 13)
int waves_per_pixel(int nmax, const Tuning &tun) {
    const int K = samples_per_lane(nmax);
    int nw = K >= 25 ? 4 : 1;
    const int v = tun.waves_per_pixel;
    if (v == 1 || (v == 4 && K >= 13)) nw = v;
    return nw;
}

bool table_in_lds(int S, int nmax, int bmax, const Tuning &tun, const SampleLayout &lay) {
    // One-wave kernels read the D table through L1: 3 .. 25 KiB of LDS per workgroup buy resident waves, which is what
    // those latency-bound kernels need.  The four-wave kernels run 1-2 workgroups per CU, every table look-up is a
    // round trip none of their few waves can cover, and one copy serves four waves: they keep the table in LDS
    // whenever that costs no resident workgroup (32 spp: 161 -> 206 ms without it).  Option "table_in_lds" overrides.
    if ((uint32_t)nmax * 8u > 65536u) return false;
    if (tun.table_in_lds >= 0) return tun.table_in_lds != 0;
    if (waves_per_pixel(nmax, tun) == 1) return false;
    const uint32_t without = lds_layout(S, nmax, bmax, false, tun, lay).total, with = lds_layout(S, nmax, bmax, true, tun, lay).total;
    return with <= (uint32_t)max_lds_per_block() && (uint32_t)max_lds_per_block() / with == (uint32_t)max_lds_per_block() / without;
}

LdsLayout lds_layout(int S, int nmax, int bmax, bool t_in_lds, const Tuning &tun, const SampleLayout &lay) {
    LdsLayout L{};
    const int K = samples_per_lane(nmax);
    const uint32_t kNDim = (uint32_t)lay.ndim(), kNPair = (uint32_t)lay.npair(), kNWt = (uint32_t)lay.nwt();
    const uint32_t KW = (uint32_t)pack_words(K);
    uint32_t o = 0;
    L.off_T = o;
    if (t_in_lds) o += align_up((uint32_t)nmax * 8u, 16);
    else if (K <= 8) o += (uint32_t)kDHead * 8u; // the one-wave K <= 8 kernels keep the head of the D table in LDS
    L.off_stat = o; o += align_up(4 * kNDim * 8, 16);
    L.off_hx = o; o += align_up(kNDim * 8, 16);
    L.off_pair = o; o += align_up(kNPair * 8, 16);
    L.off_mi = L.off_pair; // the MI values overwrite the pair sums in place
    const int nw = waves_per_pixel(nmax, tun);
    L.nw = (uint32_t)nw;
    uint32_t cells = (uint32_t)bmax * (uint32_t)bmax;
    if (K <= 8 && cells < 512u) cells = 512u; // mi_stage clears with unconditional 1-KiB stores
    if (cells < 256u) cells = 256u;           // the buffer doubles as scratch (stage 1b masks, stage 3c, flags)
    if (K > 8 || !lay.is_ref19()) cells += 64u;      // mi_group V2: one always-zero cell per lane behind the live histogram (its +0 atomics)
    L.hist_stride = align_up(cells * 4u, 1024);   // zero_cells clears whole 1-KiB rows
    // The one-wave K = 13 kernel keeps the raw own samples INSIDE its histogram buffer, behind the first KiB (stage 3c / 4's
    // scratch): nothing reads them between stage 2 and stage 4, and the kernel gathers them again when the histograms are done
    // (off_own >= off_hist tells it).  2.4 KiB at 16 spp: the eighth resident workgroup of a CU.
    const uint32_t own_bytes = align_up((uint32_t)S * kNDim * 8u, 16);
    const bool own_in_hist = K == 13 && nw == 1 && 1024u + own_bytes <= L.hist_stride;
    L.off_own = o; if (!own_in_hist) o += own_bytes;
    L.off_off = o; o += align_up((uint32_t)nmax * 4u, 16);
    L.off_union = o;
    const uint32_t stage = nw > 1 ? align_up((uint32_t)(nw - 2) * kNDim * (kStageChunk + 1) * 8u, 16) // one chunk per producer wave
                                  : align_up(kNDim * (kStageHalf + 1) * 8, 16);
    const uint32_t bins = align_up((uint32_t)kNDim * kWave * (uint32_t)pack_bytes(K), 16); // see BinIds: 5 / 9 / 4 KW bytes per (column, lane)
    uint32_t uni = bins > stage ? bins : stage;
    uint32_t fastz = align_up((uint32_t)S * ((kNWt + 2u) & ~1u) * 8u, 16);               // own rows of the weight stage
    if (nw > 1) fastz += align_up(((uint32_t)S + 1u) / 2u * (2u * ((kNWt + 2u) & ~1u)) * 4u, 16); // + the far-pair screen's fp32 rows, two own samples interleaved
    if (fastz > uni) uni = fastz;
    o += uni;
    L.off_hist = o;
    if (own_in_hist) L.off_own = L.off_hist + 1024u;
    o += L.hist_stride * (uint32_t)nw;        // one histogram buffer per wave of the pixel
    if (tun.lds_pad > 0) o += (uint32_t)tun.lds_pad; // occupancy experiment knob
    L.total = o;
    return L;
}


// LDS of the weight kernel of the split route of the 32- / 64-spp classes (filter_pixel_kernel<.., PHASE 2>): member list, own samples and
// their rows, the statistics block, 1 KiB of scratch per wave -- no table, no bin ids, no histograms
LdsLayout lds_layout_weights(int S, int nmax, const SampleLayout &lay, int nw) {
    LdsLayout L{};
    const uint32_t kNDim = (uint32_t)lay.ndim(), kNPair = (uint32_t)lay.npair(), kNWt = (uint32_t)lay.nwt();
    uint32_t o = 0;
    L.off_T = o;
    L.off_stat = o; o += align_up(4 * kNDim * 8, 16);
    L.off_hx = o; o += align_up(kNDim * 8, 16);
    L.off_pair = o; o += align_up(kNPair * 8, 16);
    L.off_mi = L.off_pair;
    L.off_own = o; o += align_up((uint32_t)S * kNDim * 8u, 16);
    L.off_off = o; o += align_up((uint32_t)nmax * 4u, 16);
    L.off_union = o;
    o += align_up((uint32_t)S * ((kNWt + 2u) & ~1u) * 8u, 16) + align_up(((uint32_t)S + 1u) / 2u * (2u * ((kNWt + 2u) & ~1u)) * 4u, 16);
    L.off_hist = o;
    L.nw = (uint32_t)nw;
    L.hist_stride = 1024;
    o += L.hist_stride * L.nw;
    // the next sample slot's values, staged by global_load_lds: per wave [2 + nF plane values][64 lanes] dwords + 3 fp64 colours as 2 x [64] dwords
    L.off_T = o;
    o += L.nw * (uint32_t)((2 + lay.nF) * kWave * 4 + 6 * kWave * 4); // (fp16 planes too: the aligned dword around each half)
    L.total = o;
    return L;
}

// ... and of its chain kernel (PHASE 3): member list, own samples, the producers' staging chunks, 2 KiB of scratch per wave
LdsLayout lds_layout_chains(int S, int nmax, const SampleLayout &lay, int nw) {
    LdsLayout L{};
    const uint32_t kNDim = (uint32_t)lay.ndim(), kNPair = (uint32_t)lay.npair();
    uint32_t o = 0;
    L.off_T = o;
    L.off_stat = o; o += align_up(4 * kNDim * 8, 16);
    L.off_hx = o; o += align_up(kNDim * 8, 16);
    L.off_pair = o; o += align_up(kNPair * 8, 16);
    L.off_mi = L.off_pair;
    L.off_own = o; o += align_up((uint32_t)S * kNDim * 8u, 16);
    L.off_off = o; o += align_up((uint32_t)nmax * 4u, 16);
    L.off_union = o;
    L.nw = (uint32_t)nw;
    o += align_up((L.nw - 2) * kNDim * (kStageChunk + 1) * 8u, 16); // one staged chunk per producer wave
    L.off_hist = o;
    L.hist_stride = 0;   // ONE 2-KiB scratch block for the workgroup (block masks of stage 1b, the producers' min / max rows):
    o += 2048;           // the chain kernel has no per-wave histograms (every wave's sHist aliases wave 0's)
    L.total = o;
    return L;
}

hipError_t launch_udiv_selftest(uint64_t n, uint64_t seed, int mode, unsigned long long *d_mismatch, hipStream_t s) {
    hipLaunchKernelGGL(udiv_selftest_kernel, dim3(2048), dim3(256), 0, s, n, seed, mode, d_mismatch);
    return hipGetLastError();
}

hipError_t launch_pixel_stats_rows(const PassParams &p, int r0, int r1, hipStream_t s) {
    if (r1 <= r0) return hipSuccess;
    const uint64_t pix0 = (uint64_t)r0 * p.W, pix1 = (uint64_t)r1 * p.W;
    return p.lay.is_ref19() ? d19::impl_pixel_stats(p, pix0, pix1, s) : d27::impl_pixel_stats(p, pix0, pix1, s);
}

hipError_t launch_pixel_stats(const PassParams &p, hipStream_t s) { return launch_pixel_stats_rows(p, 0, p.H, s); }

hipError_t launch_filter_pass(const PassParams &p, const Tuning &tun, hipStream_t s, uint32_t *lds_bytes_out) {
    if (!p.lay.supported()) return hipErrorNotSupported;
    const bool t_in_lds = table_in_lds(p.S, p.nmax, p.bmax, tun, p.lay);
    const LdsLayout L = lds_layout(p.S, p.nmax, p.bmax, t_in_lds, tun, p.lay);
    if (lds_bytes_out) *lds_bytes_out = L.total;
    if ((int)L.total > max_lds_per_block()) return hipErrorInvalidValue;
    if (L.off_T != 0) return hipErrorInvalidValue; // the kernels address the D table (head) through the LDS base itself (sD0)
    const int rows_own = p.row_end - p.row_begin;
    if (rows_own <= 0) return hipSuccess;
    if (p.pix_list != nullptr && p.list_count == 0) return hipSuccess;
    const int64_t band = p.pix_list ? (int64_t)((p.list_count + 7u) / 8u)
                                    : (int64_t)((rows_own + 7) / 8) * p.W; // pixels per XCD band (see slab_pixel)
    const unsigned grid = (unsigned)(band * 8);
    LdsLayout L2{}; // weight kernel of the split route (total == 0: not split)
    LdsLayout L3{}; // ... and of its chain kernel
    if (p.carry != nullptr && tun.split_weights != 0 && L.nw == 4 && samples_per_lane(p.nmax) >= 25) {
        const int sweeps = (p.S + (p.lay.is_ref19() ? 15 : 7)) / (p.lay.is_ref19() ? 16 : 8); // own samples per sweep of the weight kernel
        L2 = lds_layout_weights(p.S, p.nmax, p.lay, sweeps <= 2 ? 2 : 4);
        L3 = lds_layout_chains(p.S, p.nmax, p.lay, 4);
    }
    const int K = samples_per_lane(p.nmax);
    const bool r19 = p.lay.is_ref19();
    if (K == 0) return hipErrorInvalidValue;
    if (L2.total != 0 && tun.split_chunk > 0 && p.pix_list != nullptr && p.list_count > (uint32_t)tun.split_chunk) {
        // experiment (option "split_chunk"): the three launches of the split route over one chunk of the pixel list after the
        // other, so that a chunk's samples, masks and hand-over records are still in the 256 MiB Infinity Cache when the next
        // phase gathers them again (the list is in slab order: a chunk is a band of strips)
        for (uint32_t off = 0; off < p.list_count; off += (uint32_t)tun.split_chunk) {
            PassParams q = p;
            q.pix_list = p.pix_list + off;
            q.list_count = (p.list_count - off < (uint32_t)tun.split_chunk) ? p.list_count - off : (uint32_t)tun.split_chunk;
            const unsigned g = (unsigned)(((q.list_count + 7u) / 8u) * 8u);
            const hipError_t e = K <= 25 ? (r19 ? d19::impl_filter_mid(q, L, L2, L3, t_in_lds, g, s) : d27::impl_filter_mid(q, L, L2, L3, t_in_lds, g, s))
                                         : (r19 ? d19::impl_filter_large(q, L, L2, L3, t_in_lds, g, s) : d27::impl_filter_large(q, L, L2, L3, t_in_lds, g, s));
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (K <= 8) return r19 ? d19::impl_filter_small(p, L, t_in_lds, grid, s) : d27::impl_filter_small(p, L, t_in_lds, grid, s);
    if (K <= 25) return r19 ? d19::impl_filter_mid(p, L, L2, L3, t_in_lds, grid, s) : d27::impl_filter_mid(p, L, L2, L3, t_in_lds, grid, s);
    return r19 ? d19::impl_filter_large(p, L, L2, L3, t_in_lds, grid, s) : d27::impl_filter_large(p, L, L2, L3, t_in_lds, grid, s);
}

int class_capacity(int c) {
    static const int caps[kNumClasses] = {8, 16, 32, 64, 128, 256, 448, 832, 1600, 3136, kMaxNbhd}; // packed lane classes; 64 * {2, 4, 7, 13, 25, 49}; streaming
    return caps[c];
}

hipError_t launch_nbhd_count(const PassParams &p, int step, uint32_t *probe, const uint32_t *list, const uint32_t *list_count, uint32_t list_max, hipStream_t s) {
    if (!p.lay.supported() || step < 1) return hipErrorInvalidValue;
    return p.lay.is_ref19() ? d19::impl_nbhd_count(p, step, probe, list, list_count, list_max, s)
                            : d27::impl_nbhd_count(p, step, probe, list, list_count, list_max, s);
}

hipError_t launch_filter_packed(const PassParams &p, int lanes_per_pixel, const uint32_t *count_dev, hipStream_t s) {
    if (!p.lay.supported()) return hipErrorNotSupported;
    return p.lay.is_ref19() ? d19::impl_filter_packed(p, lanes_per_pixel, count_dev, s) : d27::impl_filter_packed(p, lanes_per_pixel, count_dev, s);
}

hipError_t launch_filter_big(const PassParams &p, void *list, void *bins, uint32_t slots, const uint32_t *count_dev, hipStream_t s) {
    if (!p.lay.supported() || p.pix_list == nullptr) return hipErrorInvalidValue;
    return p.lay.is_ref19() ? d19::impl_filter_big(p, list, bins, slots, count_dev, s) : d27::impl_filter_big(p, list, bins, slots, count_dev, s);
}

hipError_t launch_classify(const PassParams &p, uint32_t *lists, uint32_t *counts, int max_class, int rest_class, hipStream_t s) {
    const int64_t total = (int64_t)((p.row_end - p.row_begin + 7) / 8) * p.W * 8;
    if (total <= 0) return hipSuccess;
    ClassCaps caps;
    for (int c = 0; c < kNumClasses; ++c) caps.cap[c] = class_capacity(c);
    hipLaunchKernelGGL(classify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, caps, lists, counts,
                       (uint64_t)p.H * p.W, max_class, rest_class);
    return hipGetLastError();
}

hipError_t launch_prelist(const PassParams &p, uint32_t *list, uint32_t *count, hipStream_t s) {
    const int64_t total = (int64_t)((p.row_end - p.row_begin + 7) / 8) * p.W * 8;
    if (total <= 0 || p.flat == nullptr || p.nan_flag == nullptr) return hipSuccess;
    hipLaunchKernelGGL(prelist_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, list, count);
    return hipGetLastError();
}

hipError_t launch_colour_from_planes_span(const void *planes, bool f16, double *colour, uint64_t ps, uint64_t e0,
                                          uint64_t cnt, hipStream_t s) {
    if (cnt == 0) return hipSuccess;
    const dim3 grid((unsigned)((3 * cnt + 255) / 256));
    if (f16) hipLaunchKernelGGL(colour_from_planes_kernel<__half>, grid, dim3(256), 0, s, (const __half *)planes, colour, ps, e0, cnt);
    else hipLaunchKernelGGL(colour_from_planes_kernel<float>, grid, dim3(256), 0, s, (const float *)planes, colour, ps, e0, cnt);
    return hipGetLastError();
}

hipError_t launch_colour_from_planes(const void *planes, bool f16, double *colour, uint64_t ps, hipStream_t s) {
    return launch_colour_from_planes_span(planes, f16, colour, ps, 0, ps, s);
}

hipError_t launch_copy_colour_span(const double *src, double *dst, uint64_t ps, uint64_t e0, uint64_t cnt, hipStream_t s) {
    if (cnt == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_colour_span_kernel, dim3((unsigned)((3 * cnt + 255) / 256)), dim3(256), 0, s, src, dst, ps,
                       e0, cnt);
    return hipGetLastError();
}

hipError_t launch_copy_f64(const double *src, double *dst, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(copy_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_reduce_rows(const double *colour, const float *ray_weight, float *sample_rgb, float *pixel_rgb, int W,
                              int H, int S, int r0, int r1, hipStream_t s) {
    if (r1 <= r0) return hipSuccess;
    const uint64_t pix0 = (uint64_t)r0 * W, pix1 = (uint64_t)r1 * W;
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((pix1 - pix0 + 255) / 256)), dim3(256), 0, s, colour, ray_weight,
                       sample_rgb, pixel_rgb, W, H, S, pix0, pix1);
    return hipGetLastError();
}

hipError_t launch_reduce(const double *colour, const float *ray_weight, float *sample_rgb, float *pixel_rgb, int W,
                         int H, int S, hipStream_t s) {
    return launch_reduce_rows(colour, ray_weight, sample_rgb, pixel_rgb, W, H, S, 0, H, s);
}

hipError_t launch_feature_images(const float *planes, int W, int H, int S, double *out, unsigned long long *maxbits, hipStream_t s) {
    const uint64_t HW = (uint64_t)H * W;
    hipError_t e = hipMemsetAsync(maxbits, 0, 18 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(feature_mean_kernel, dim3((unsigned)((HW + 255) / 256), 6), dim3(256), 0, s, planes, HW * S, HW, S, out, maxbits);
    hipLaunchKernelGGL(feature_normalise_kernel, dim3((unsigned)((18 * HW + 255) / 256)), dim3(256), 0, s, out, HW, maxbits);
    return hipGetLastError();
}

hipError_t launch_nbhd_reduce(const int32_t *nbhd, int W, int row_begin, int row_end, unsigned long long *out2,
                              hipStream_t s) {
    const uint64_t begin = (uint64_t)row_begin * W, end = (uint64_t)row_end * W;
    if (end <= begin) return hipSuccess;
    unsigned grid = (unsigned)((end - begin + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(nbhd_reduce_kernel, dim3(grid), dim3(256), 0, s, nbhd, begin, end, out2);
    return hipGetLastError();
}

} // namespace rpf
