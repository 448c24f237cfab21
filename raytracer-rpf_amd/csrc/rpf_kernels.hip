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
#include <hip/hip_runtime.h>
#include <math.h>
#include <cstdlib>
#include <type_traits>

#include "rpf_internal.h"
#include "rpf_xlane.h"

namespace rpf {

namespace {

constexpr int kDHead = 128;   // entries of the D table the one-wave kernels keep in LDS
constexpr int kTFixBits = 44; // T[k] = k ln k is tabulated as round(T * 2^44): exact integer sums, |T| < 2^15 * 2^44

// MI pair table in ComputeCFWeights call order (rpf.cpp:416-442)
struct PairTable {
    unsigned char a[kNPair], b[kNPair];
};
constexpr PairTable make_pairs() {
    PairTable t{};
    int p = 0;
    for (int i = 0; i < 12; ++i) {
        for (int l = 0; l < 2; ++l) { t.a[p] = kColF + i; t.b[p] = kColR + l; ++p; }
        for (int l = 0; l < 2; ++l) { t.a[p] = kColF + i; t.b[p] = kColP + l; ++p; }
    }
    for (int c = 0; c < 3; ++c) {
        for (int l = 0; l < 2; ++l) { t.a[p] = kColC + c; t.b[p] = kColR + l; ++p; }
        for (int l = 0; l < 2; ++l) { t.a[p] = kColC + c; t.b[p] = kColP + l; ++p; }
        for (int j = 0; j < 12; ++j) { t.a[p] = kColC + c; t.b[p] = kColF + j; ++p; }
    }
    return t;
}
__constant__ PairTable c_pairs = make_pairs();

// The workgroup is exactly one wavefront, and one wave's LDS operations execute in issue order, so a
// producer/consumer hand-off between lanes through LDS needs no s_barrier and no counter drain (a
// __syncthreads() here would also wait for every gather in flight: vmcnt(0)).  What it needs is that the
// compiler keeps the program order of the memory operations: a wavefront-scope fence.
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Workgroup barrier that publishes LDS traffic only: lgkmcnt(0) + s_barrier.  __syncthreads() also drains vmcnt, i.e.
// it would wait for every gather a producer wave has in flight; here those gathers are meant to stay in flight across
// the barrier (the compiler still waits on them, by register dependence, where their values are used).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint32_t fnv1a_u32(uint32_t h, uint32_t v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 16777619u; }
    return h;
}
__device__ __forceinline__ uint32_t fnv1a_u16(uint32_t h, uint32_t v) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 16777619u; }
    return h;
}

// KW consecutive 32-bit words per lane, as one wide LDS access where the width allows
template <int KW>
__device__ __forceinline__ void load_words(const uint32_t *src, uint32_t (&w)[KW]) {
    if constexpr (KW % 4 == 0) {
#pragma unroll
        for (int i = 0; i < KW / 4; ++i) {
            const uint4 v = reinterpret_cast<const uint4 *>(src)[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    } else if constexpr (KW % 2 == 0) {
#pragma unroll
        for (int i = 0; i < KW / 2; ++i) {
            const uint2 v = reinterpret_cast<const uint2 *>(src)[i];
            w[2 * i] = v.x; w[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < KW; ++i) w[i] = src[i];
    }
}
template <int KW>
__device__ __forceinline__ void store_words(uint32_t *dst, const uint32_t (&w)[KW]) {
    if constexpr (KW % 4 == 0) {
#pragma unroll
        for (int i = 0; i < KW / 4; ++i) reinterpret_cast<uint4 *>(dst)[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    } else if constexpr (KW % 2 == 0) {
#pragma unroll
        for (int i = 0; i < KW / 2; ++i) reinterpret_cast<uint2 *>(dst)[i] = make_uint2(w[2 * i], w[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < KW; ++i) dst[i] = w[i];
    }
}

// clear `cells` 32-bit histogram cells (buffer is 16-byte aligned and padded to a multiple of 4 cells)
__device__ __forceinline__ void zero_words(uint32_t *h, int cells, int lane) {
    for (int t = lane * 4; t < cells; t += kWave * 4) *reinterpret_cast<uint4 *>(h + t) = make_uint4(0u, 0u, 0u, 0u);
}

// value of column c of the sample at plane offset `off`: colours come from the fp64 colour planes
__device__ __forceinline__ double load_col(const PassParams &p, int c, uint32_t off) {
    if (c >= kColC && c < kColC + 3) return p.col_in[(uint64_t)(c - kColC) * p.plane_stride + off];
    return (double)p.planes[(uint64_t)c * p.plane_stride + off];
}

// ------------------------------------------------------------------------------------------------
// stage 1a: per-pixel mean / std of the 12 features over the pixel's own S samples, sequential sums
// (rpf.cpp:338-347, ops.h:111-144).  Output planes [12][H*W] so the filter kernel reads them with
// wave-uniform (scalar) loads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pixel_stats_kernel(PassParams p, uint64_t pix0, uint64_t pix1) {
    const uint64_t HW = (uint64_t)p.H * p.W;
    const uint64_t pix = pix0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= pix1) return;
    const double dn = (double)p.S;
    for (int k = 0; k < kNFeat; ++k) {
        const float *src = p.planes + (uint64_t)(kColF + k) * p.plane_stride + pix * p.S;
        double sum = 0.0, sq = 0.0;
        for (int s = 0; s < p.S; ++s) {
            double v = (double)src[s];
            sum = sum + v;     // ops.h:121
            sq = sq + v * v;   // ops.h:138 (v*v is exact for fp32-valued v)
        }
        double mean = sum / dn;                     // ops.h:123
        double sd = sqrt(sq / dn - mean * mean);    // ops.h:141
        if (p.policy == RPF_DEGEN_EPS && isnan(sd)) sd = 0.0;
        ((double *)p.pmean)[(uint64_t)k * HW + pix] = mean;
        ((double *)p.pstd)[(uint64_t)k * HW + pix] = sd;
    }
}

// ---- exact fp64 division by a wave-uniform divisor ---------------------------------------------------
// hipcc lowers a/b (f64) to v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal, q0 = a*r,
// rem = fma(-b,q0,a), q = fma(rem,r,q0) (v_div_fmas), v_div_fixup.  For operands in the normal range the
// scale steps are the identity and the fixup passes q through, so the quotient is exactly
// fma(fma(-b, a*r, a), r, a*r) with r depending on b only.  Every division of stage 3a has a divisor that
// is the same for all samples of a column, so r is refined ONCE per column and each sample pays three
// instructions.  The numerators there are differences of fp32-valued samples and their means: 0 or of
// magnitude within [2^-215, 2^130], so with the divisor inside [2^-100, 2^100] every intermediate stays in
// the normal range; a column whose divisor is outside that window takes the plain operator instead
// (wave-uniform branch).  Bit-identity with a/b: tests/test_gpu_parity.py::test_uniform_divisor_division_is_exact.
struct UDiv {
    double b, r;
    bool fast;
};
__device__ __forceinline__ UDiv udiv_prepare(double b) {
    UDiv d;
    d.b = b;
    const double ab = fabs(b);
    d.fast = (ab > 0x1p-100) && (ab < 0x1p100);
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    d.r = r;
    return d;
}
// valid when d.fast and the numerator is 0 or has magnitude in [2^-400, 2^400]
__device__ __forceinline__ double udiv_fast(double a, const UDiv &d) {
    const double q0 = a * d.r;
    const double rem = fma(-d.b, q0, a);
    return fma(rem, d.r, q0);
}
__device__ __forceinline__ double udiv(double a, const UDiv &d) {
    const double aa = fabs(a);
    const bool ok = d.fast && ((aa == 0.0) || ((aa > 0x1p-400) && (aa < 0x1p400)));
    return ok ? udiv_fast(a, d) : a / d.b;
}

// ---- stage 3b: histograms -> mutual information (mi.cpp:45-90) ------------------------------------------
// mi.cpp:79-86 over integer counts:  N*MI = T[N] + sum_ij T[J_ij] - sum_i T[hx_i] - sum_j T[hy_j],  T[k] = k ln k.
// T is tabulated in 2^-44 fixed point, so every sum is an exact integer sum: the result does not depend on the
// order in which lanes hit a cell, a single-bin column gives exactly MI == 0 like the reference (pX == 1 =>
// every log term is log(1)), and no log is evaluated on the device.
// Each increment is a returning LDS atomic; the old count c contributes D[c] = T[c+1]-T[c], which telescopes
// to T[J] per cell.  The histogram is cleared by wide stores queued right behind the atomics (one wave's LDS
// operations execute in order).  Histograms are processed four at a time: the atomics of histogram u+1 are
// queued before the D look-ups of histogram u are consumed, and the four per-lane sums are reduced together by
// one transposed butterfly (no LDS).  KD = number of occupied sample slots of this pixel (compile time).
// Clearing a histogram with unconditional full-wave stores (no exec masking, no branch; cells past the live histogram
// are scratch; the buffer holds >= 512 cells whenever ZN > 0).  ZN selects the store set by histogram size:
//   1: one 16-byte store per lane (256 cells)        3: + one 4-byte store  (320 cells: B <= 17)
//   4: + one 8-byte store (384 cells: B <= 19)        2: + one 16-byte store (512 cells)
//   0: generic loop (large neighbourhoods)
template <int ZN>
__device__ __forceinline__ void zero_cells(uint32_t *h, int cells, int lane) {
    if constexpr (ZN > 0) {
        *reinterpret_cast<uint4 *>(h + lane * 4) = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (ZN == 2) *reinterpret_cast<uint4 *>(h + 256 + lane * 4) = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (ZN == 3) h[256 + lane] = 0u;
        if constexpr (ZN == 4) *reinterpret_cast<uint2 *>(h + 256 + lane * 2) = make_uint2(0u, 0u);
    } else {
        for (int t = lane * 4; t < cells; t += kWave * 4) *reinterpret_cast<uint4 *>(h + t) = make_uint4(0u, 0u, 0u, 0u);
    }
}

// ---- bin-id storage of one pixel ---------------------------------------------------------------------------
// PACK5 names the packing scheme (an int, historically a bool):
//   1  kernels with K <= 8 (B = floor(sqrt(N)) <= 22): sample slots 0..5 of a lane are 5-bit fields of ONE 32-bit
//      word per (column, lane) -- [19][64] words -- and slot 6 is a byte in a side array [19][64] behind them:
//      6 KiB per pixel, which is what lets 12 single-wave workgroups share a CU's LDS
//   5  K <= 17 (B <= 32): 5-bit fields, six per word, KW = ceil(K/6) words per (column, lane)
//   6  larger K (B <= 64): 6-bit fields, five per word, KW = ceil(K/5) words per (column, lane)
__host__ __device__ constexpr int pack_scheme(int K) { return K <= 8 ? 1 : (K <= 17 ? 5 : 6); }
__host__ __device__ constexpr int pack_words(int K) { return K <= 8 ? 1 : (K <= 17 ? (K + 5) / 6 : (K + 4) / 5); }

template <int KW, int PACK5>
struct BinIds {
    static constexpr int BITS = PACK5 == 6 ? 6 : 5;
    static constexpr int SPW = 32 / BITS; // slots per word
    uint32_t w[KW];
    uint32_t b6;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < KW; ++i) w[i] = 0u;
        b6 = 0u;
    }
    __device__ __forceinline__ void set(int kk, uint32_t bin) { // kk is a compile-time constant at every call site
        if constexpr (PACK5 == 1) {
            if (kk < 6) w[0] |= bin << (5 * kk); else b6 = bin;
        } else {
            w[kk / SPW] |= bin << (BITS * (kk % SPW));
        }
    }
    __device__ __forceinline__ uint32_t get(int kk) const {
        if constexpr (PACK5 == 1) return kk < 6 ? ((w[0] >> (5 * kk)) & 31u) : b6;
        else return (w[kk / SPW] >> (BITS * (kk % SPW))) & ((1u << BITS) - 1u);
    }
};
template <int KD, int KW, int PACK5>
__device__ __forceinline__ void store_bins(uint32_t *sBinW, int c, int lane, const BinIds<KW, PACK5> &b) {
    if constexpr (PACK5 == 1) {
        sBinW[c * kWave + lane] = b.w[0];
        if (KD > 6) reinterpret_cast<uint8_t *>(sBinW + kNDim * kWave)[c * kWave + lane] = (uint8_t)b.b6;
    } else {
        store_words<KW>(sBinW + ((size_t)c * kWave + lane) * KW, b.w);
    }
}
template <int KD, int KW, int PACK5>
__device__ __forceinline__ void load_bins(const uint32_t *sBinW, int c, int lane, BinIds<KW, PACK5> &b) {
    if constexpr (PACK5 == 1) {
        b.w[0] = sBinW[c * kWave + lane];
        b.b6 = (KD > 6) ? (uint32_t) reinterpret_cast<const uint8_t *>(sBinW + kNDim * kWave)[c * kWave + lane] : 0u;
    } else {
        load_words<KW>(sBinW + ((size_t)c * kWave + lane) * KW, b.w);
        b.b6 = 0u;
    }
}
// bin id of sample j = lane + 64*slot (debug hash only)
template <int KW, int PACK5>
__device__ __forceinline__ uint32_t bin_of_sample(const uint32_t *sBinW, int c, int j) {
    const int ln = j & 63, slot = j >> 6;
    if constexpr (PACK5 == 1) {
        if (slot < 6) return (sBinW[c * kWave + ln] >> (5 * slot)) & 31u;
        return reinterpret_cast<const uint8_t *>(sBinW + kNDim * kWave)[c * kWave + ln];
    } else {
        constexpr int BITS = PACK5 == 6 ? 6 : 5, SPW = 32 / BITS;
        return (sBinW[((size_t)c * kWave + ln) * KW + slot / SPW] >> (BITS * (slot % SPW))) & ((1u << BITS) - 1u);
    }
}

// ---- stage 3a: normalise, bin ids (sd.h:229-232, mi.cpp:14-16) -----------------------------------------
// bin ids are bytes packed per lane: sample kk of the lane is byte kk of KW words per column, written to LDS
// [column][lane][KW] over the (now dead) staging buffer of stage 2.  KD = sample slots handled (the occupied
// ones when the kernel is specialised on them, else K); holes compute on a dummy value and are masked.
// NW > 1 (several waves per pixel): the waves split the COLUMNS -- wave wv takes fp32 columns wv, wv+NW, ... of the
// 16 and colour column wv -- and each still covers all n samples, so no bin word is shared between waves.
// Per-column constants of stage 3a, formed ONCE per pixel by lane c for column c (they are wave-uniform: every lane
// forming all 19 sets was ~1000 redundant VALU instructions per pixel) and read back by broadcast:
//   ck[0] refined 1/SD   ck[1] lo = min z   ck[2] range = max z - min z   ck[3] refined 1/range
//   ck[4] flags: 1 SD == 0 | 2 empty z range | 4 both divisors inside the fast window
constexpr int kColConst = 5;
__device__ __forceinline__ void column_constants(const double *sStat, double *sCK, int c) {
    const double Mc = sStat[c], SDc = sStat[kNDim + c];
    const double xlo = sStat[2 * kNDim + c], xhi = sStat[3 * kNDim + c];
    const bool sd0 = (SDc == 0.0);
    const UDiv dsd = udiv_prepare(SDc);
    const double lo = sd0 ? 0.0 : udiv(xlo - Mc, dsd); // min_element over z (mi.cpp:47,49)
    const double hi = sd0 ? 0.0 : udiv(xhi - Mc, dsd); // max_element over z (mi.cpp:48,50)
    const double range = hi - lo;
    const bool flat = !(hi != lo);                       // mi.cpp:7 / 28 / 34
    const UDiv drg = udiv_prepare(range);
    const bool fast = dsd.fast && (flat || drg.fast);
    double *ck = sCK + c * kColConst;
    ck[0] = dsd.r; ck[1] = lo; ck[2] = range; ck[3] = drg.r;
    ck[4] = (double)((sd0 ? 1 : 0) | (flat ? 2 : 0) | (fast ? 4 : 0));
}

template <int KD, int KW, int PACK5, int NW = 1>
__device__ __forceinline__ void bins_stage(const PassParams &p, const double *sStat, const uint32_t *sOff, uint32_t *sBinW,
                                           int lane, int n, int B, const double *sCK, int wv = 0) {
    const double dB = (double)B;
    {
        uint32_t offk[KD];
#pragma unroll
        for (int kk = 0; kk < KD; ++kk) offk[kk] = (lane + kWave * kk < n) ? sOff[lane + kWave * kk] : 0u;
        // one column: z, t, bin for the lane's K samples, packed into KW words
        auto do_column = [&](int c, const double (&xv)[KD]) {
            const double Mc = sStat[c], SDc = sStat[kNDim + c];
            const double *ck = sCK + c * kColConst;             // column_constants(): wave-uniform broadcast reads
            const double lo = ck[1], range = ck[2];
            const int flags = (int)ck[4];
            const bool sd0 = flags & 1, flat = flags & 2, fast = flags & 4;
            UDiv dsd, drg;
            dsd.b = SDc; dsd.r = ck[0]; dsd.fast = true;
            drg.b = range; drg.r = ck[3]; drg.fast = true;
            BinIds<KW, PACK5> w;
            w.clear();
            if (sd0 || flat) {
                // SD == 0 normalises every sample to z == 0 (ops.h:48), and a column whose z range is empty bins every
                // sample to 0 (mi.cpp:7): the cleared words are the answer -- no per-sample work (constant normals /
                // positions inside a cluster make this the common case on path-traced buffers)
            } else if (fast) {
                // the common case, straight-line for all K samples of the lane (holes compute on a dummy value
                // and are masked at the pack), so the K dependent chains interleave
#pragma unroll
                for (int kk = 0; kk < KD; ++kk) {
                    const double a = xv[kk] - Mc;                          // subtractArrays
                    const double z = udiv_fast(a, dsd);                    // divideArrays, ops.h:48
                    const double t = udiv_fast(z - lo, drg) * dB;          // mi.cpp:14
                    int bin = (int)t;
                    bin = min(bin, B - 1);
                    bin = max(bin, 0);
                    bin = (lane + kWave * kk < n) ? bin : 0;
                    w.set(kk, (uint32_t)bin);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KD; ++kk) {
                    if (lane + kWave * kk < n) {
                        const double a = xv[kk] - Mc;
                        const double z = sd0 ? 0.0 : a / SDc;
                        int bin = 0;
                        if (!flat) {
                            const double t = (z - lo) / range * dB;
                            bin = (int)t;
                            bin = min(bin, B - 1);
                            bin = max(bin, 0);
                        }
                        w.set(kk, (uint32_t)bin);
                    }
                }
            }
            store_bins<KD, KW, PACK5>(sBinW, c, lane, w);
        };
        if constexpr (NW > 1) {
            if (p.stage_mask & 2) {
                auto colidx = [](int i) { return i < 2 ? i : i + 3; };
                float xb[KD];
                auto issue3 = [&](int i) {
                    const float *fplane = p.planes + (uint64_t)colidx(i) * p.plane_stride;
#pragma unroll
                    for (int kk = 0; kk < KD; ++kk) xb[kk] = fplane[offk[kk]];
                };
                issue3(wv); // NW <= 16
#pragma unroll 1
                for (int i = wv; i < 16; i += NW) {
                    double xv[KD];
#pragma unroll
                    for (int kk = 0; kk < KD; ++kk) xv[kk] = (double)xb[kk];
                    if (i + NW < 16) issue3(i + NW);
                    do_column(colidx(i), xv);
                }
#pragma unroll 1
                for (int c = wv; c < 3; c += NW) {
                    double xc[KD];
                    const double *dplane = p.col_in + (uint64_t)c * p.plane_stride;
#pragma unroll
                    for (int kk = 0; kk < KD; ++kk) xc[kk] = dplane[offk[kk]];
                    do_column(kColC + c, xc);
                }
            }
        } else if (p.stage_mask & 2) {
            // the 16 fp32 columns (0,1,5..18) through kPF3 rotating register buffers, gathers kPF3 columns ahead
            constexpr int kPF3 = KD <= 8 ? 4 : (KD <= 13 ? 2 : 1); // register budget: KD floats per buffer
            float xb[kPF3][KD];
            auto colidx = [](int i) { return i < 2 ? i : i + 3; };
            auto issue3 = [&](int i, float (&dst)[KD]) {
                const float *fplane = p.planes + (uint64_t)colidx(i) * p.plane_stride;
#pragma unroll
                for (int kk = 0; kk < KD; ++kk) dst[kk] = fplane[offk[kk]];
            };
#pragma unroll
            for (int u = 0; u < kPF3; ++u) issue3(u, xb[u]);
#pragma unroll 1
            for (int i0 = 0; i0 < 16; i0 += kPF3) {
#pragma unroll
                for (int u = 0; u < kPF3; ++u) {
                    double xv[KD];
#pragma unroll
                    for (int kk = 0; kk < KD; ++kk) xv[kk] = (double)xb[u][kk];
                    if (i0 + u + kPF3 < 16) issue3(i0 + u + kPF3, xb[u]);
                    do_column(colidx(i0 + u), xv);
                }
            }
            // the 3 fp64 colour columns
            double xc[3][KD];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double *dplane = p.col_in + (uint64_t)c * p.plane_stride;
#pragma unroll
                for (int kk = 0; kk < KD; ++kk) xc[c][kk] = dplane[offk[kk]];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) do_column(kColC + c, xc[c]);
        }
    }
}

// ---- marginal counts off a resident joint histogram ------------------------------------------------------------
// sum_i T[hx_i] of a column is needed once per column (19 of them).  Building 19 more histograms for that is the most
// expensive way to get it (a marginal has only B cells: 64 lanes pile onto a handful of addresses).  Instead the
// counts are read off a joint histogram that is resident anyway, between its atomics and its clearing store (one
// wave's LDS operations execute in order): hy_j = sum_i J[i][j] (column sums: the partner's marginal), hx_i = sum_j
// J[i][j] (row sums: the anchor's).  Counts are integers: same totals, bit for bit.  B <= 32: lane = (h, j), h = lane
// >> 5 takes every other row / column, the two halves meet in one v_permlane32_swap; NIT = compile-time trip count
// (>= ceil(B/2)).
template <int NIT, bool ROWS>
__device__ __forceinline__ uint32_t marginal_counts(const uint32_t *sHist, int lane, int B) {
    uint32_t tot = 0u;
    const int j = lane & 31, h = lane >> 5;
    uint32_t v[NIT];
    if constexpr (!ROWS) {
        // column sums: rows i = h, h + 2, ... of column j.  No masks: a row index past B lands in the cleared cells
        // behind the live histogram (the clearing stores of every ZN class cover 2 * NIT * B cells, and nothing
        // increments them), lanes j >= B read a neighbouring row and are dropped below.
        const uint32_t *col = sHist + h * B + j;
#pragma unroll
        for (int t = 0; t < NIT; ++t) v[t] = col[2 * t * B];
#pragma unroll
        for (int t = 0; t < NIT; ++t) tot += v[t];
    } else {
        const int jj = min(j, B - 1);
#pragma unroll
        for (int t = 0; t < NIT; ++t) v[t] = sHist[jj * B + min(2 * t + h, B - 1)];
#pragma unroll
        for (int t = 0; t < NIT; ++t) tot += (2 * t + h < B) ? v[t] : 0u;
    }
    tot = xl::exch32<xl::OpSum>(tot, tot);          // both halves now hold the full count of bin j
    return (j < B && h == 0) ? tot : 0u;            // one lane per bin contributes T[count] (T[0] == 0)
}

// D[c] through a 32-bit byte offset from the (wave-uniform) table base: scalar base + vector offset addressing
__device__ __forceinline__ uint64_t dlook(const uint64_t *dtab, uint32_t c) {
    return *reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(dtab) + (c << 3));
}

// ---- one-wave kernels (K <= 8, 3 <= KD): PARTNER-major histogram groups ---------------------------------------
// All joint histograms of ONE partner column, one per anchor: the partner's bin ids are unpacked once for the
// whole group and every anchor's (bin * B * 4 + histogram base) sits in registers (akey4[anchor][slot], formed once per
// pixel), so the address of an increment is a single v_lshl_add.  NA = anchors of the group (7 for a feature partner:
// r0 r1 p0 p1 c0 c1 c2; 4 for a colour partner: r0 r1 p0 p1).  Slots kk < KD-1 are full by definition of KD
// (= ceil(n/64)), so only the last slot carries the hole mask: `last_ok` lanes exist, the others aim a +0 atomic at `hole`.
// Histogram u+1's atomics are queued before the D look-ups of histogram u are consumed; the clearing store sits right
// behind the atomics.  MARG bit 0: the partner's marginal from the column sums of histogram 0 -> macc; bit 1: the
// anchors' marginals from the row sums of every histogram of the group -> racc[u].
// D look-ups: the first kDHead entries of the table sit in LDS (1 KiB: costs no resident workgroup), which is where
// nearly every look-up lands (a cell that holds >= kDHead of a neighbourhood's <= 512 samples is a degenerate pixel); the
// 480 look-ups per pixel were the kernel's main load on the texture-address path.  A wave in which some count reaches
// kDHead repeats the pixel's MI stage with the full table in global memory (one wave-uniform branch per pixel, outside the
// straight-line histogram code: a branch per histogram cost 146 spilled registers).
template <int KD, int KW, int ZN, int NA, int NAMAX, int PACK5, int MARG, bool DHEAD>
__device__ __forceinline__ void mi_pgroup(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *dtab, const uint64_t *sDh, const uint64_t *ttab,
                                          int lane, int B, int pcol, const uint32_t (&akey4)[NAMAX][KD], bool last_ok,
                                          uint32_t hole4, int cells, uint64_t (&acc)[8], uint64_t &macc, uint64_t (&racc)[4],
                                          uint32_t &mx) {
    constexpr int NIT = ZN == 1 ? 8 : (ZN == 3 ? 9 : (ZN == 4 ? 10 : 11)); // B <= 16 / 17 / 19 / 22
    static_assert(NA <= 8 && (!(MARG & 2) || NA <= 4), "slot budget of the reductions");
    uint32_t bin4[KD];
    {
        BinIds<KW, PACK5> w;
        load_bins<KD, KW, PACK5>(sBinW, pcol, lane, w);
#pragma unroll
        for (int kk = 0; kk < KD; ++kk) bin4[kk] = w.get(kk) << 2;
    }
    char *hbase = reinterpret_cast<char *>(sHist);
    uint32_t old[2][KD];
    uint32_t mcnt = 0u, rcnt[2] = {0u, 0u};
    const uint32_t last_inc = last_ok ? 1u : 0u;
#pragma unroll
    for (int u = 0; u <= NA; ++u) {
        if (u < NA) {
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) {
                uint32_t off = akey4[u][kk] + bin4[kk];                                  // mi.cpp:39 (x 4 bytes)
                if (kk == KD - 1) off = last_ok ? off : hole4;
                old[u & 1][kk] = atomicAdd(reinterpret_cast<uint32_t *>(hbase + off), kk == KD - 1 ? last_inc : 1u);
            }
            if constexpr (MARG != 0) {
                if ((MARG & 2) || u == 0) {
                    wsync(); // the counts other lanes' atomics left are read below (in-order LDS: program order only)
                    if constexpr (MARG & 1) if (u == 0) mcnt = marginal_counts<NIT, false>(sHist, lane, B);
                    if constexpr ((MARG & 2) != 0) rcnt[u & 1] = marginal_counts<NIT, true>(sHist, lane, B);
                    wsync();
                }
            }
            zero_cells<ZN>(sHist, cells, lane);
        }
        if (u >= 1) {
            uint64_t d[KD];
            if constexpr (DHEAD) {
#pragma unroll
                for (int kk = 0; kk < KD; ++kk) {
                    const uint32_t o = old[(u - 1) & 1][kk];
                    mx = max(mx, o);                // a count past the head: the caller repeats the stage (see there)
                    d[kk] = dlook(sDh, o);          // ds_read_b64
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KD; ++kk) d[kk] = dlook(dtab, old[(u - 1) & 1][kk]);
            }
            uint64_t a = last_ok ? d[KD - 1] : 0ull;
#pragma unroll
            for (int kk = 0; kk < KD - 1; ++kk) a += d[kk];
            acc[u - 1] = a;
            if constexpr ((MARG & 2) != 0) racc[u - 1] = dlook(ttab, rcnt[(u - 1) & 1]);
        }
    }
    if constexpr (MARG & 1) macc = dlook(ttab, mcnt);
#pragma unroll
    for (int u = NA; u < 8; ++u) acc[u] = 0ull;
}

// The partner-major MI stage.  Anchor a: 0..1 = r0, r1; 2..3 = p0, p1; 4..6 = c0..c2.  Pair index = position in
// ComputeCFWeights' call order (rpf.cpp:416-442).  Marginal sums sHXf: every partner's from its own group (f0..f11,
// c0..c2 are all partners), r0 r1 p0 p1 from the row sums of the c0 group's four histograms.
template <int KD, int KW, int ZN, int PACK5, bool DHEAD>
__device__ __forceinline__ uint32_t mi_stage_pm(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *dtab, const uint64_t *sDh, const uint64_t *ttab,
                                            uint64_t *sHXf, uint64_t *sPairF, int lane, int n, int B) {
    constexpr int NA = 7;
    const int ncell2 = B * B;
    const bool last_ok = (lane + kWave * (KD - 1)) < n;                 // does this lane's last sample slot exist?
    const uint32_t hole4 = (uint32_t)min(lane, ncell2 - 1) << 2;        // harmless, spread-out targets of the +0 atomics
    zero_cells<ZN>(sHist, ncell2, lane);
    uint32_t mx = 0u; // largest count any look-up of this lane saw (DHEAD)
    uint32_t akey4[NA][KD];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int acol = a < 2 ? kColR + a : (a < 4 ? kColP + (a - 2) : kColC + (a - 4));
        BinIds<KW, PACK5> w;
        load_bins<KD, KW, PACK5>(sBinW, acol, lane, w);
#pragma unroll
        for (int kk = 0; kk < KD; ++kk) akey4[a][kk] = w.get(kk) * (uint32_t)(B * 4);
    }
    auto pair_index = [&](int a, int i) { // anchor a, partner i (0..11 features, 12..14 colours)
        return a < 4 ? (i < 12 ? i * 4 + a : 48 + (i - 12) * 16 + a) : 48 + (a - 4) * 16 + 4 + i;
    };
#pragma unroll 1
    for (int i = 0; i < kNFeat; ++i) { // feature partners: 7 histograms each
        uint64_t acc[8], macc = 0ull, racc[4];
        mi_pgroup<KD, KW, ZN, 7, NA, PACK5, 1, DHEAD>(sBinW, sHist, dtab, sDh, ttab, lane, B, kColF + i, akey4, last_ok, hole4, ncell2, acc, macc, racc, mx);
        acc[7] = macc; // the free eighth slot carries the partner's marginal
        const uint64_t tot = xl::reduce8<xl::OpSum>(acc, lane);
        if ((lane & 7) == 0) {
            const int sl = xl::slot8(lane);
            if (sl < 7) sPairF[pair_index(sl, i)] = tot;
            else sHXf[kColF + i] = tot;
        }
    }
#pragma unroll 1
    for (int c = 0; c < 3; ++c) { // colour partners: 4 histograms each (anchors r0 r1 p0 p1)
        uint64_t acc[8], macc = 0ull, racc[4] = {0ull, 0ull, 0ull, 0ull};
        if (c == 0) mi_pgroup<KD, KW, ZN, 4, NA, PACK5, 3, DHEAD>(sBinW, sHist, dtab, sDh, ttab, lane, B, kColC, akey4, last_ok, hole4, ncell2, acc, macc, racc, mx);
        else mi_pgroup<KD, KW, ZN, 4, NA, PACK5, 1, DHEAD>(sBinW, sHist, dtab, sDh, ttab, lane, B, kColC + c, akey4, last_ok, hole4, ncell2, acc, macc, racc, mx);
        acc[7] = macc;
        if (c == 0) { acc[4] = racc[0]; acc[5] = racc[1]; acc[6] = racc[2]; }
        const uint64_t tot = xl::reduce8<xl::OpSum>(acc, lane);
        const uint64_t r3 = (c == 0) ? xl::allreduce<xl::OpSum>(racc[3]) : 0ull; // wave-uniform branch
        if ((lane & 7) == 0) {
            const int sl = xl::slot8(lane);
            if (sl < 4) sPairF[pair_index(sl, 12 + c)] = tot;
            else if (sl == 7) sHXf[kColC + c] = tot;
            else if (c == 0) sHXf[sl == 4 ? kColR : (sl == 5 ? kColR + 1 : kColP)] = tot;
        }
        if (c == 0 && lane == 0) sHXf[kColP + 1] = r3;
    }
    return mx;
}

template <int KD, int KW, int ZN, int PACK5, bool DHEAD>
__device__ __forceinline__ void mi_pm(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *dtab, const uint64_t *sDh,
                                      const uint64_t *ttab, uint64_t *sHXf, uint64_t *sPairF, int lane, int n, int B) {
    const uint32_t mx = mi_stage_pm<KD, KW, ZN, PACK5, DHEAD>(sBinW, sHist, dtab, sDh, ttab, sHXf, sPairF, lane, n, B);
    if constexpr (DHEAD) {
        if (__any(mx >= (uint32_t)kDHead)) { // some cell count ran past the LDS head of the table: once more, full table
            wsync();
            mi_stage_pm<KD, KW, ZN, PACK5, false>(sBinW, sHist, dtab, sDh, ttab, sHXf, sPairF, lane, n, B);
        }
    }
}

// ---- large-neighbourhood kernels (K >= 13): ANCHOR-major groups of G (3 or 4) histograms ----------------------------
// For marginals (JOINT = false) histogram u bins column cols[u]; for joints it bins (anchor, partner cols[u]).  Every
// slot carries its own mask (lane + 64*kk < n): masked lanes aim a +0 atomic at `hole`.
template <int KD, int KW, int G, int PACK5, bool JOINT>
__device__ __forceinline__ void mi_group(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *dtab, int lane, int n,
                                         const int (&cols)[4], const uint32_t (&akey)[KD], uint32_t hole, int cells,
                                         uint64_t (&acc4)[4]) {
    uint32_t old[2][KD];
    auto slot_ok = [&](int kk) -> bool { return lane + kWave * kk < n; };
#pragma unroll
    for (int u = 0; u <= G; ++u) {
        if (u < G) {
            BinIds<KW, PACK5> w;
            load_bins<KD, KW, PACK5>(sBinW, cols[u], lane, w);
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) {
                uint32_t key = w.get(kk);
                if (JOINT) key += akey[kk];                                  // mi.cpp:39
                const bool ok = slot_ok(kk);
                old[u & 1][kk] = atomicAdd(&sHist[ok ? key : hole], ok ? 1u : 0u);
            }
            zero_cells<0>(sHist, cells, lane);
        }
        if (u >= 1) {
            uint64_t d[KD];
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) d[kk] = dtab[old[(u - 1) & 1][kk]];
            uint64_t a = slot_ok(KD - 1) ? d[KD - 1] : 0ull;
#pragma unroll
            for (int kk = 0; kk < KD - 1; ++kk) a += slot_ok(kk) ? d[kk] : 0ull;
            acc4[u - 1] = a;
        }
    }
#pragma unroll
    for (int u = G; u < 4; ++u) acc4[u] = 0ull;
}

// Small neighbourhoods (KD <= 2 sample slots, i.e. N <= 128: the regime of real path-traced buffers, where most
// pixels keep only their own S samples): a histogram costs a handful of LDS cycles, so the stage is latency bound.
// This variant queues the atomics of up to 16 histograms back to back (one wave's LDS operations execute in
// order, the clearing store sits between two histograms), then issues all D look-ups, then reduces the 16 sums
// with one 16-slot butterfly: two round trips per 16 histograms instead of per 4.
template <int KD, int KW, int ZN, int G, bool JOINT, int PACK5>
__device__ __forceinline__ void mi_group_deep(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *dtab, int lane,
                                              const int (&cols)[16], const uint32_t (&akey)[KD], bool last_ok,
                                              uint32_t hole, int cells, uint64_t (&acc)[16]) {
    uint32_t old[G][KD];
    const uint32_t one = 1u, last_inc = last_ok ? 1u : 0u;
#pragma unroll
    for (int u = 0; u < G; ++u) {
        BinIds<KW, PACK5> w;
        load_bins<KD, KW, PACK5>(sBinW, cols[u], lane, w);
#pragma unroll
        for (int kk = 0; kk < KD; ++kk) {
            uint32_t key = w.get(kk);
            if (JOINT) key += akey[kk];                                      // mi.cpp:39
            if (kk == KD - 1) key = last_ok ? key : hole;
            old[u][kk] = atomicAdd(&sHist[key], kk == KD - 1 ? last_inc : one);
        }
        zero_cells<(JOINT ? ZN : (ZN > 0 ? 1 : 0))>(sHist, cells, lane);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        uint64_t a = 0ull;
        if (u < G) {
            uint64_t d[KD];
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) d[kk] = dtab[old[u][kk]];
            a = last_ok ? d[KD - 1] : 0ull;
#pragma unroll
            for (int kk = 0; kk < KD - 1; ++kk) a += d[kk];
        }
        acc[u] = a;
    }
}

template <int KD, int KW, int ZN, int PACK5>
__device__ __forceinline__ void mi_stage_deep(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *dtab, uint64_t *sHXf,
                                              uint64_t *sPairF, int lane, int n, int B) {
    const int ncell2 = B * B;
    const bool last_ok = (lane + kWave * (KD - 1)) < n;
    const uint32_t hole1 = (uint32_t)min(lane, B - 1);
    const uint32_t hole2 = (uint32_t)min(lane, ncell2 - 1);
    zero_cells<ZN>(sHist, ncell2, lane);
    uint32_t akey[KD];
#pragma unroll
    for (int kk = 0; kk < KD; ++kk) akey[kk] = 0u;
    uint64_t acc[16];
    { // marginals: columns 0..15, then 16..18
        const int cols[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
        mi_group_deep<KD, KW, ZN, 16, false, PACK5>(sBinW, sHist, dtab, lane, cols, akey, last_ok, hole1, B, acc);
        const uint64_t tot = xl::reduce16<xl::OpSum>(acc, lane);
        if ((lane & 3) == 0) sHXf[xl::slot16(lane)] = tot;
    }
    {
        const int cols[16] = {16, 17, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18};
        mi_group_deep<KD, KW, ZN, 3, false, PACK5>(sBinW, sHist, dtab, lane, cols, akey, last_ok, hole1, B, acc);
        const uint64_t tot = xl::reduce16<xl::OpSum>(acc, lane);
        if ((lane & 3) == 0 && xl::slot16(lane) < 3) sHXf[16 + xl::slot16(lane)] = tot;
    }
#pragma unroll 1
    for (int g = 0; g < 7; ++g) {
        const int acol = g < 2 ? kColR + g : (g < 4 ? kColP + (g - 2) : kColC + (g - 4));
        const int l = g < 2 ? g : 2 + (g - 2);
        {
            BinIds<KW, PACK5> w;
            load_bins<KD, KW, PACK5>(sBinW, acol, lane, w);
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) akey[kk] = w.get(kk) * (uint32_t)B;
        }
        auto pair_index = [&](int i) { return g < 4 ? (i < 12 ? i * 4 + l : 48 + (i - 12) * 16 + l) : 48 + (g - 4) * 16 + 4 + i; };
        // partners f0..f11 then (anchors r,p only) c0..c2; the c-anchors repeat a column for the unused slots
        const int cols[16] = {kColF, kColF + 1, kColF + 2, kColF + 3, kColF + 4, kColF + 5, kColF + 6, kColF + 7,
                              kColF + 8, kColF + 9, kColF + 10, kColF + 11, kColC, kColC + 1, kColC + 2, kColC + 2};
        const int np = g < 4 ? 15 : 12;
        if (g < 4) mi_group_deep<KD, KW, ZN, 15, true, PACK5>(sBinW, sHist, dtab, lane, cols, akey, last_ok, hole2, ncell2, acc);
        else mi_group_deep<KD, KW, ZN, 12, true, PACK5>(sBinW, sHist, dtab, lane, cols, akey, last_ok, hole2, ncell2, acc);
        const uint64_t tot = xl::reduce16<xl::OpSum>(acc, lane);
        const int i = xl::slot16(lane);
        if ((lane & 3) == 0 && i < np) sPairF[pair_index(i)] = tot;
    }
}

// Tiny neighbourhoods (one sample slot: N <= 64, B <= 8, at most 64 cells) -- what real path-traced buffers mostly
// are (N = S for >90 % of pixels).  Returning atomics would serialise here: a dozen lanes hit the same
// few cells.  Instead each 16-lane row increments its own replica of the histogram with a plain (non-returning)
// atomic, lane c then reads the four replica counts of cell c with one 16-byte read, looks up T[count] and clears
// the cell with one 16-byte store; two 256-word regions ping-pong so the atomics of histogram u+1 are queued before
// histogram u is read.  Sixteen histograms share one 16-slot butterfly.  Layout of a region: [cell][replica].
template <int G, bool JOINT, int KW, int PACK5>
__device__ __forceinline__ void mi_group_tiny(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *ttab, int lane,
                                              const int (&cols)[16], uint32_t akey, bool ok, uint64_t (&acc)[16]) {
    const uint32_t rep = (uint32_t)lane >> 4;
#pragma unroll
    for (int u = 0; u <= G; ++u) {
        if (u < G) {
            BinIds<KW, PACK5> w;
            load_bins<1, KW, PACK5>(sBinW, cols[u], lane, w);
            uint32_t key = w.get(0);
            if (JOINT) key += akey;                                          // mi.cpp:39
            if (ok) atomicAdd(&sHist[(u & 1) * 256 + key * 4 + rep], 1u);
        }
        if (u >= 1) {
            uint32_t *cell = sHist + ((u - 1) & 1) * 256 + lane * 4;         // lane = cell id (64 cells per region)
            const uint4 c4 = *reinterpret_cast<const uint4 *>(cell);
            *reinterpret_cast<uint4 *>(cell) = make_uint4(0u, 0u, 0u, 0u);
            acc[u - 1] = ttab[c4.x + c4.y + c4.z + c4.w];                    // T[J] of this lane's cell (T[0] = 0)
        }
    }
#pragma unroll
    for (int u = G; u < 16; ++u) acc[u] = 0ull;
}

template <int KW, int PACK5>
__device__ __forceinline__ void mi_stage_tiny(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *ttab, uint64_t *sHXf,
                                              uint64_t *sPairF, int lane, int n, int B) {
    const bool ok = lane < n;
    *reinterpret_cast<uint4 *>(sHist + lane * 4) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4 *>(sHist + 256 + lane * 4) = make_uint4(0u, 0u, 0u, 0u);
    uint64_t acc[16];
    {
        const int cols[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
        mi_group_tiny<16, false, KW, PACK5>(sBinW, sHist, ttab, lane, cols, 0u, ok, acc);
        const uint64_t tot = xl::reduce16<xl::OpSum>(acc, lane);
        if ((lane & 3) == 0) sHXf[xl::slot16(lane)] = tot;
    }
    {
        const int cols[16] = {16, 17, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18};
        mi_group_tiny<3, false, KW, PACK5>(sBinW, sHist, ttab, lane, cols, 0u, ok, acc);
        const uint64_t tot = xl::reduce16<xl::OpSum>(acc, lane);
        if ((lane & 3) == 0 && xl::slot16(lane) < 3) sHXf[16 + xl::slot16(lane)] = tot;
    }
#pragma unroll 1
    for (int g = 0; g < 7; ++g) {
        const int acol = g < 2 ? kColR + g : (g < 4 ? kColP + (g - 2) : kColC + (g - 4));
        const int l = g < 2 ? g : 2 + (g - 2);
        uint32_t akey;
        {
            BinIds<KW, PACK5> w;
            load_bins<1, KW, PACK5>(sBinW, acol, lane, w);
            akey = w.get(0) * (uint32_t)B;
        }
        auto pair_index = [&](int i) { return g < 4 ? (i < 12 ? i * 4 + l : 48 + (i - 12) * 16 + l) : 48 + (g - 4) * 16 + 4 + i; };
        const int cols[16] = {kColF, kColF + 1, kColF + 2, kColF + 3, kColF + 4, kColF + 5, kColF + 6, kColF + 7,
                              kColF + 8, kColF + 9, kColF + 10, kColF + 11, kColC, kColC + 1, kColC + 2, kColC + 2};
        const int np = g < 4 ? 15 : 12;
        if (g < 4) mi_group_tiny<15, true, KW, PACK5>(sBinW, sHist, ttab, lane, cols, akey, ok, acc);
        else mi_group_tiny<12, true, KW, PACK5>(sBinW, sHist, ttab, lane, cols, akey, ok, acc);
        const uint64_t tot = xl::reduce16<xl::OpSum>(acc, lane);
        const int i = xl::slot16(lane);
        if ((lane & 3) == 0 && i < np) sPairF[pair_index(i)] = tot;
    }
}

// Large neighbourhoods (K >= 13).  NW > 1: the 30 histogram groups (5 marginal + 25 joint) are dealt round-robin to
// the NW waves of the pixel; every wave has its own histogram buffer and covers all n samples of its groups, so the
// stage needs no barrier.  (These kernels sit at the register limit of 2 or 1 waves per SIMD: the marginal read-back
// and the partner-major key registers of the one-wave kernels cost them spills, measured 305 -> 466 ms on a 4K x 32 spp slab.)
template <int KD, int KW, int PACK5, int NW = 1>
__device__ __forceinline__ void mi_stage(const uint32_t *sBinW, uint32_t *sHist, const uint64_t *dtab, uint64_t *sHXf,
                                         uint64_t *sPairF, int lane, int n, int B, int wv = 0) {
    int gi = 0; // running group number (wave-uniform)
    auto mine = [&]() { const bool m = (NW == 1) || (gi % NW) == wv; ++gi; return m; };
    const int ncell2 = B * B;
    const uint32_t hole1 = (uint32_t)min(lane, B - 1);      // harmless, spread-out targets of the +0 atomics
    const uint32_t hole2 = (uint32_t)min(lane, ncell2 - 1);
    zero_cells<0>(sHist, ncell2, lane);
    uint32_t akey[KD];
#pragma unroll
    for (int kk = 0; kk < KD; ++kk) akey[kk] = 0u;
    // ---- marginals: sum_i T[hx_i] per column; 19 columns = 4 groups of 4 + one group of 3
#pragma unroll 1
    for (int c0 = 0; c0 < 16; c0 += 4) {
        if (!mine()) continue;
        uint64_t acc4[4];
        const int cols[4] = {c0, c0 + 1, c0 + 2, c0 + 3};
        mi_group<KD, KW, 4, PACK5, false>(sBinW, sHist, dtab, lane, n, cols, akey, hole1, B, acc4);
        const uint64_t tot = xl::reduce4<xl::OpSum>(acc4);
        if ((lane & 15) == 0) sHXf[c0 + xl::slot4(lane)] = tot;
    }
    if (mine()) {
        uint64_t acc4[4];
        const int cols[4] = {16, 17, 18, 18};
        mi_group<KD, KW, 3, PACK5, false>(sBinW, sHist, dtab, lane, n, cols, akey, hole1, B, acc4);
        const uint64_t tot = xl::reduce4<xl::OpSum>(acc4);
        if ((lane & 15) == 0 && xl::slot4(lane) < 3) sHXf[16 + xl::slot4(lane)] = tot;
    }
    // ---- joint histograms, grouped by an anchor column whose (bin * B) stays in registers
    //   anchors 0..3 = r0, r1, p0, p1 with partners f0..f11, c0..c2 ; anchors 4..6 = c0..c2 with f0..f11
#pragma unroll 1
    for (int g = 0; g < 7; ++g) {
        const int acol = g < 2 ? kColR + g : (g < 4 ? kColP + (g - 2) : kColC + (g - 4));
        const int l = g < 2 ? g : 2 + (g - 2); // r0,r1 -> 0,1 ; p0,p1 -> 2,3
        {
            BinIds<KW, PACK5> w;
            load_bins<KD, KW, PACK5>(sBinW, acol, lane, w);
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) akey[kk] = w.get(kk) * (uint32_t)B;
        }
        // pair index in ComputeCFWeights call order (rpf.cpp:416-442)
        auto pair_index = [&](int i) { return g < 4 ? (i < 12 ? i * 4 + l : 48 + (i - 12) * 16 + l) : 48 + (g - 4) * 16 + 4 + i; };
#pragma unroll 1
        for (int i0 = 0; i0 < 12; i0 += 4) { // partners f0..f11
            if (!mine()) continue;
            uint64_t acc4[4];
            const int cols[4] = {kColF + i0, kColF + i0 + 1, kColF + i0 + 2, kColF + i0 + 3};
            mi_group<KD, KW, 4, PACK5, true>(sBinW, sHist, dtab, lane, n, cols, akey, hole2, ncell2, acc4);
            const uint64_t tot = xl::reduce4<xl::OpSum>(acc4);
            if ((lane & 15) == 0) sPairF[pair_index(i0 + xl::slot4(lane))] = tot;
        }
        if (g < 4 && mine()) { // partners c0..c2 (wave-uniform branch)
            uint64_t acc4[4];
            const int cols[4] = {kColC, kColC + 1, kColC + 2, kColC + 2};
            mi_group<KD, KW, 3, PACK5, true>(sBinW, sHist, dtab, lane, n, cols, akey, hole2, ncell2, acc4);
            const uint64_t tot = xl::reduce4<xl::OpSum>(acc4);
            if ((lane & 15) == 0 && xl::slot4(lane) < 3) sPairF[pair_index(12 + xl::slot4(lane))] = tot;
        }
    }
}

// floor(n / d) for n * d < 2^32 with the precomputed M = floor((2^32 - 1) / d) + 1: one v_mul_hi_u32 instead of the
// ~25-instruction sequence hipcc emits for an integer division by a run-time divisor (exact: n (M d - 2^32) < 2^32)
__device__ __forceinline__ uint32_t div_magic(uint32_t d) { return 0xFFFFFFFFu / d + 1u; }
__device__ __forceinline__ uint32_t div_small(uint32_t n, uint32_t M) { return M ? __umulhi(n, M) : n; } // M == 0: d == 1

// XCD- and L2-aware pixel order of a slab.  Blocks with equal (blockIdx % 8) share an XCD and its 4 MiB L2: XCD r
// filters one contiguous band of rows, and walks it in vertical strips of kStripW pixels (row by row inside a strip),
// so the 7-row window data of the ~256 pixels in flight on the XCD (and the 6 rows shared with the next strip row)
// stay L2-resident instead of being re-fetched once per image row.  (r, ql) -> pixel; false = no such pixel.
__device__ __forceinline__ bool slab_pixel(const PassParams &p, int r, int64_t ql, int &x, int &y) {
    constexpr int kStripW = 128;
    const int W = p.W;
    const int rows_own = p.row_end - p.row_begin;
    const int rows_band = (rows_own + 7) / 8;
    const int band_row0 = r * rows_band;
    const int band_rows = min(rows_band, rows_own - band_row0);
    if (band_rows <= 0) return false;
    if (ql >= (int64_t)band_rows * W) return false;
    const uint32_t q = (uint32_t)ql;                       // band_rows * W < 2^32 (host-checked: W*H*S < 2^32)
    const uint32_t full_strips = (uint32_t)W / kStripW;
    const uint32_t strip_px = (uint32_t)kStripW * (uint32_t)band_rows;
    int yl;
    if (q < full_strips * strip_px) {
        const uint32_t sidx = q / strip_px;
        const uint32_t rr = q - sidx * strip_px;
        yl = (int)(rr / kStripW);
        x = (int)(sidx * kStripW + (rr - (uint32_t)yl * kStripW));
    } else {
        const uint32_t tw = (uint32_t)W - full_strips * kStripW;
        const uint32_t rr = q - full_strips * strip_px;
        yl = (int)(rr / tw);
        x = (int)(full_strips * kStripW + (rr - (uint32_t)yl * tw));
    }
    y = p.row_begin + band_row0 + yl;
    return true;
}

// ------------------------------------------------------------------------------------------------
// the fused per-pixel kernel
//   K         compile-time bound on samples per lane: K*64 >= nmax ; lane owns samples j = lane + 64*kk
//   T_IN_LDS  keep the D table in LDS (small neighbourhoods) instead of reading it through L1
// ------------------------------------------------------------------------------------------------
//   NW        waves per pixel.  1: the workgroup is one wavefront (8 spp: twelve pixels in flight per CU).  4 (large
//             neighbourhoods, whose 40-85 KiB of LDS would otherwise leave 1-3 waves on a CU): the waves of a pixel
//             share the member list, the bin ids and the own rows, and split the work by COLUMN (stages 2, 3a), by
//             HISTOGRAM GROUP (3b, one private histogram buffer per wave) and by OWN SAMPLE (4), so that every wave
//             still walks all n samples in lane + 64*kk order and the stages need a barrier only where they meet.
template <int K, bool T_IN_LDS, bool FAST, int NW>
__global__ __launch_bounds__(64 * NW, (NW > 1 ? (K <= 25 ? 2 : 1) : (K <= 8 ? 3 : (K <= 13 ? 2 : 1)))) void filter_pixel_kernel(PassParams p, LdsLayout L) {
    constexpr int KW = pack_words(K);        // 32-bit words of packed bin ids per lane and column
    constexpr int PACK5 = pack_scheme(K);    // packing scheme, see BinIds
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *sD = reinterpret_cast<uint64_t *>(smem + L.off_T); // D[c] = T[c+1]-T[c], 2^-44 fixed point
    double *sStat = reinterpret_cast<double *>(smem + L.off_stat); // M[19], SD[19], xmin[19], xmax[19]
    uint64_t *sHXf = reinterpret_cast<uint64_t *>(smem + L.off_hx);     // sum_i T[hx_i] per column (fixed point)
    uint64_t *sPairF = reinterpret_cast<uint64_t *>(smem + L.off_pair); // sum_ij T[J_ij] per pair (fixed point)
    double *sMI = reinterpret_cast<double *>(smem + L.off_pair);   // 96 MI values, in place over the pair sums
    double *sOwn = reinterpret_cast<double *>(smem + L.off_own);   // raw own samples [S][19]
    uint32_t *sOff = reinterpret_cast<uint32_t *>(smem + L.off_off);
    double *sStage = reinterpret_cast<double *>(smem + L.off_union);    // [19][kStageHalf+1] (aliases bins)
    uint32_t *sBinW = reinterpret_cast<uint32_t *>(smem + L.off_union); // bin ids [19][64][KW] words (K > 8)
    uint32_t *sHist0 = reinterpret_cast<uint32_t *>(smem + L.off_hist); // wave 0's histogram buffer (also scratch)

    const int tid = threadIdx.x;
    const int lane = NW > 1 ? (tid & (kWave - 1)) : tid;
    const int wv = NW > 1 ? (tid >> 6) : 0; // wave-uniform
    uint32_t *sHist = reinterpret_cast<uint32_t *>(smem + L.off_hist + (NW > 1 ? (uint32_t)wv * L.hist_stride : 0u));
    // hand-off between the waves of the pixel (s_barrier) -- or between the lanes of the only wave
    auto bsync = [&]() { if constexpr (NW > 1) __syncthreads(); else wsync(); };
    constexpr int kThreads = kWave * NW;
    const int W = p.W, H = p.H, S = p.S, b = p.b;

    // pixel of this workgroup: see slab_pixel(); a size-binned launch walks its pixel list instead, dealt to the
    // XCDs in eight contiguous chunks (the list is in slab_pixel order, so a chunk is again a band of strips)
    int x, y;
    if (p.pix_list != nullptr) {
        const uint32_t chunk = (p.list_count + 7u) / 8u;
        const uint32_t ql = blockIdx.x >> 3, e = (blockIdx.x & 7u) * chunk + ql;
        if (ql >= chunk || e >= p.list_count) return;
        const uint32_t pp = p.pix_list[e];
        y = (int)(pp / (uint32_t)W);
        x = (int)(pp - (uint32_t)y * (uint32_t)W);
    } else if (!slab_pixel(p, (int)(blockIdx.x & 7), (int64_t)(blockIdx.x >> 3), x, y)) {
        return;
    }
    const uint64_t HW = (uint64_t)H * W;
    const uint64_t pix = (uint64_t)y * W + x;


    if (T_IN_LDS) {
        for (int k = tid; k < p.nmax; k += kThreads) sD[k] = p.dfix[k];
    } else if (K <= 8) {
        for (int k = tid; k < min(p.nmax, kDHead); k += kThreads) sD[k] = p.dfix[k]; // the head of the table (mi_pgroup)
    }

    // ---------------- stage 1b: neighbourhood membership (rpf.cpp:556-586) ----------------------
    const int x0 = max(x - b, 0), x1 = min(x + b, W - 1);
    const int y0 = max(y - b, 0), y1 = min(y + b, H - 1);
    const int nyv = y1 - y0 + 1;
    const int ncells = (x1 - x0 + 1) * nyv;
    const int centre_rank = (x - x0) * nyv + (y - y0);
    const int ncand = (ncells - 1) * S;
    const uint32_t magic_S = div_magic((uint32_t)S), magic_ny = div_magic((uint32_t)nyv); // qq < 4096, S, nyv <= 64

    for (int s = tid; s < S; s += kThreads) sOff[s] = (uint32_t)(pix * S + s); // own samples first

    int n = S;
    if constexpr (NW > 1) {
        // Several waves: the 64-candidate blocks are dealt round-robin to the waves.  Pass A tests a wave's blocks and
        // leaves one acceptance mask per block in LDS; pass B turns the masks into list positions (an exclusive scan
        // over <= 64 block counts, done by every wave for itself) and appends, so the list order is the reference's.
        uint64_t *sMask = reinterpret_cast<uint64_t *>(sHist0);
        const int nblk = (ncand + kWave - 1) / kWave; // <= 64 (host-checked: nmax <= 4096)
        const bool have_masks = p.masks != nullptr;   // size-binned launch: nbhd_count_kernel already ran the test
        if (have_masks && tid < nblk) sMask[tid] = p.masks[pix * p.mask_stride + tid];
        double m12[kNFeat], lim12[kNFeat];
#pragma unroll
        for (int k = 0; k < kNFeat; ++k) {
            m12[k] = p.pmean[(uint64_t)k * HW + pix];
            lim12[k] = p.pstd[(uint64_t)k * HW + pix] * 3.0; // multiplyArray(std, 3), rpf.cpp:579
        }
        auto cand_off = [&](int qq) -> uint32_t {
            int cell = (int)div_small((uint32_t)qq, magic_S);
            const int s = qq - cell * S;
            if (cell >= centre_rank) ++cell;          // rpf.cpp:565: skip the centre pixel
            const int ix = (int)div_small((uint32_t)cell, magic_ny); // xn outer ascending (rpf.cpp:562)
            const int iy = cell - ix * nyv;            // yn inner ascending (rpf.cpp:563)
            return (uint32_t)(((uint64_t)(y0 + iy) * W + (x0 + ix)) * S + s);
        };
        constexpr int kPF1 = 3;
        float fb[kPF1][kNFeat];
        auto issue1 = [&](int blk, float (&f)[kNFeat]) {
            const int qq = blk * kWave + lane;
            if (blk < nblk && qq < ncand) {
                const uint32_t off = cand_off(qq);
#pragma unroll
                for (int k = 0; k < kNFeat; ++k) f[k] = p.planes[(uint64_t)(kColF + k) * p.plane_stride + off];
            }
        };
#pragma unroll
        for (int u = 0; u < kPF1; ++u)
            if (!have_masks) issue1(wv + NW * u, fb[u]);
#pragma unroll 1
        for (int t0 = 0; !have_masks && wv + NW * t0 < nblk; t0 += kPF1) {
#pragma unroll
            for (int u = 0; u < kPF1; ++u) {
                const int blk = wv + NW * (t0 + u);
                if (blk < nblk) { // wave-uniform
                    bool pass = (blk * kWave + lane) < ncand;
#pragma unroll
                    for (int k = 0; k < kNFeat; ++k) {
                        const double a = fabs((double)fb[u][k] - m12[k]);
                        if (a >= lim12[k]) pass = false;       // allLessThan: fails iff a >= b (ops.h:101-104)
                    }
                    const unsigned long long mask = __ballot(pass);
                    if (lane == 0) sMask[blk] = mask;
                    issue1(blk + NW * kPF1, fb[u]);
                }
            }
        }
        __syncthreads();
        const int cnt = lane < nblk ? __popcll(sMask[lane]) : 0;
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int t = __shfl_up(incl, d, kWave);
            if (lane >= d) incl += t;
        }
        const int excl = incl - cnt;
#pragma unroll 1
        for (int blk = wv; blk < nblk; blk += NW) {
            const int base = S + __shfl(excl, blk, kWave);
            const unsigned long long mask = sMask[blk];
            const int at = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (((mask >> lane) & 1ull) && at < p.nmax) sOff[at] = cand_off(blk * kWave + lane);
        }
        n = S + __shfl(incl, kWave - 1, kWave);
    } else if (p.masks != nullptr) {
        // size-binned launch: nbhd_count_kernel already ran the 3-sigma test and left one acceptance mask per 64
        // candidates; only the list is rebuilt here (wave-uniform mask loads, no feature gathers)
        const uint64_t *pm = p.masks + pix * p.mask_stride;
#pragma unroll 1
        for (int qb = 0; qb < ncand; qb += kWave) {
            const unsigned long long mask = pm[qb >> 6];
            const int at = n + __popcll(mask & ((1ull << lane) - 1ull));
            if (((mask >> lane) & 1ull) && at < p.nmax) {
                const int qq = qb + lane;
                int cell = (int)div_small((uint32_t)qq, magic_S);
                const int s = qq - cell * S;
                if (cell >= centre_rank) ++cell;          // rpf.cpp:565: skip the centre pixel
                const int ix = (int)div_small((uint32_t)cell, magic_ny); // xn outer ascending (rpf.cpp:562)
                const int iy = cell - ix * nyv;            // yn inner ascending (rpf.cpp:563)
                sOff[at] = (uint32_t)(((uint64_t)(y0 + iy) * W + (x0 + ix)) * S + s);
            }
            n += __popcll(mask);
        }
    } else {
        double m12[kNFeat], lim12[kNFeat];
#pragma unroll
        for (int k = 0; k < kNFeat; ++k) {
            m12[k] = p.pmean[(uint64_t)k * HW + pix];
            lim12[k] = p.pstd[(uint64_t)k * HW + pix] * 3.0; // multiplyArray(std, 3), rpf.cpp:579
        }
        // candidate -> plane offset of its sample, in the reference's visiting order
        auto cand_off = [&](int qq) -> uint32_t {
            int cell = (int)div_small((uint32_t)qq, magic_S);
            const int s = qq - cell * S;
            if (cell >= centre_rank) ++cell;          // rpf.cpp:565: skip the centre pixel
            const int ix = (int)div_small((uint32_t)cell, magic_ny); // xn outer ascending (rpf.cpp:562)
            const int iy = cell - ix * nyv;            // yn inner ascending (rpf.cpp:563)
            return (uint32_t)(((uint64_t)(y0 + iy) * W + (x0 + ix)) * S + s);
        };
        // rotating register buffers: the 12 feature gathers of the next kPF1 64-candidate steps are in flight while
        // a step is tested and appended (an L2/MALL round trip is ~1-2k cycles under load, a step ~0.5k)
        constexpr int kPF1 = 3;
        float fb[kPF1][kNFeat];
        uint32_t ob[kPF1];
        auto issue1 = [&](int qq, float (&f)[kNFeat], uint32_t &off) {
            if (qq < ncand) {
                off = cand_off(qq);
#pragma unroll
                for (int k = 0; k < kNFeat; ++k) f[k] = p.planes[(uint64_t)(kColF + k) * p.plane_stride + off];
            }
        };
#pragma unroll
        for (int u = 0; u < kPF1; ++u) issue1(u * kWave + lane, fb[u], ob[u]);
#pragma unroll 1
        for (int q0 = 0; q0 < ncand; q0 += kWave * kPF1) {
#pragma unroll
            for (int u = 0; u < kPF1; ++u) {
                const int qb = q0 + u * kWave;
                if (qb < ncand) { // wave-uniform
                    bool pass = (qb + lane) < ncand;
#pragma unroll
                    for (int k = 0; k < kNFeat; ++k) {
                        const double a = fabs((double)fb[u][k] - m12[k]);
                        if (a >= lim12[k]) pass = false;       // allLessThan: fails iff a >= b (ops.h:101-104)
                    }
                    const unsigned long long mask = __ballot(pass);
                    const int at = n + __popcll(mask & ((1ull << lane) - 1ull));
                    if (pass && at < p.nmax) sOff[at] = ob[u]; // (at < nmax always: the bound only guards LDS)
                    n += __popcll(mask);
                    issue1(qb + kWave * kPF1 + lane, fb[u], ob[u]);
                }
            }
        }
    }
    bsync();
    if (tid == 0) p.nbhd[pix] = n;

    if (p.dbg.member_hash != nullptr && tid == 0) {
        uint32_t h = 2166136261u;
        for (int j = 0; j < n; ++j) {
            const uint32_t o = sOff[j];
            const uint32_t s = o % (uint32_t)S;
            const uint32_t pp = o / (uint32_t)S;
            const int yn = (int)(pp / (uint32_t)W), xn = (int)(pp % (uint32_t)W);
            h = fnv1a_u32(h, (uint32_t)(((xn - x + b) * p.box + (yn - y + b)) * S) + s);
        }
        p.dbg.member_hash[pix] = h;
    }

    // ---------------- stage 2: mean / std over the neighbourhood, reference order ---------------
    // Chunks of 64 samples: lane t gathers all 19 values of sample j0+t (the next chunk is already in
    // flight in registers), stages them (two halves of 32) as doubles [column][t] in LDS, then lanes 0..18 run the in-order
    // sum(x) chain of column `lane` and lanes 32..50 the sum(x*x) chain of column `lane-32`.
    // The per-column min / max of x ride along (order independent): z = (x-M)/SD is monotone in x, so
    // min z = z(min x) and max z = z(max x) exactly, which is all mi.cpp:47-50 needs.
    if constexpr (NW > 1) {
        // Several waves: the in-order chains cannot be split (fp64 addition is not associative), so two waves do nothing
        // but run them while the others are producers: in round r producer w gathers the 19 values of the 64 samples of
        // chunk r*NP+w (issued one round ahead), tracks the column min / max and stages the chunk as doubles
        // [column][65] in its own LDS buffer.  Barrier A: the buffers of the round are complete; the chain waves walk
        // them in order while the producers' next gathers are in flight; barrier B: the buffers may be overwritten.
        // Two chain waves: wave 0 sums x, wave 1 sums x*x (lanes 0..18 each) -- a wave's own instruction stream is what
        // bounds a chain, and one wave doing both needed a per-element select; waves 2.. are the producers.
        constexpr int NCH = 2;
        constexpr int NP = NW - NCH;
        constexpr int kCS = kStageChunk + 1; // doubles per staged column
        const int nrun = (p.stage_mask & 1) ? n : 0;
        const int nround = (nrun + NP * kStageChunk - 1) / (NP * kStageChunk);
        double *sMM = reinterpret_cast<double *>(sHist0); // [NP][38]: column min | max seen by each producer
        double *sSq = sMM + NP * 2 * kNDim;               // [19]: wave 1's sums of squares
        double chain_acc = 0.0;                           // chain waves: the running sum of column `lane`
        if (wv < NCH) {
            auto chain_loop = [&](auto sq_tag) {
                constexpr bool SQ = decltype(sq_tag)::value;
                double acc = 0.0;
                const bool chain = lane < kNDim;
                for (int r = 0; r < nround; ++r) {
                    lds_barrier(); // A
                    if (nrun - r * NP * kStageChunk >= NP * kStageChunk) {
                        // full round: NP*4 batches of 16 staged values, the reads of batch t+1 issued before batch t
                        // is summed (this wave is alone on its SIMD: nothing else hides the LDS latency)
                        constexpr int NB = NP * (kStageChunk / 16);
                        const double *col = sStage + (chain ? lane : 0) * kCS;
                        double vb[2][16];
                        auto load16 = [&](int t, double (&v)[16]) {
                            const double *src = col + (t / 4) * (kNDim * kCS) + (t % 4) * 16;
#pragma unroll
                            for (int q = 0; q < 16; ++q) v[q] = src[q];
                        };
                        load16(0, vb[0]);
#pragma unroll
                        for (int t = 0; t < NB; ++t) {
                            if (t + 1 < NB) load16(t + 1, vb[(t + 1) & 1]);
                            double (&v)[16] = vb[t & 1];
                            if constexpr (SQ) {
#pragma unroll
                                for (int q = 0; q < 16; ++q) v[q] = v[q] * v[q];       // ops.h:138 multiplyArrays
                            }
#pragma unroll
                            for (int q = 0; q < 16; ++q) acc = acc + v[q];             // ops.h:121 / 138 sumArrays
                        }
                    } else {
#pragma unroll 1
                        for (int w = 0; w < NP; ++w) {
                            const int cnt = min(kStageChunk, nrun - (r * NP + w) * kStageChunk); // wave-uniform
                            if (cnt <= 0) break;
                            if (chain) {
                                const double *src = sStage + w * (kNDim * kCS) + lane * kCS;
                                int q0 = 0;
                                for (; q0 + 16 <= cnt; q0 += 16) {
                                    double v[16];
#pragma unroll
                                    for (int q = 0; q < 16; ++q) v[q] = src[q0 + q];
                                    if constexpr (SQ) {
#pragma unroll
                                        for (int q = 0; q < 16; ++q) v[q] = v[q] * v[q];
                                    }
#pragma unroll
                                    for (int q = 0; q < 16; ++q) acc = acc + v[q];
                                }
                                for (int q = q0; q < cnt; ++q) { const double v = src[q]; acc = acc + (SQ ? v * v : v); }
                            }
                        }
                    }
                    lds_barrier(); // B
                }
                return acc;
            };
            if (wv == 0) {
                chain_acc = chain_loop(std::false_type{});
            } else {
                const double sq = chain_loop(std::true_type{});
                if (lane < kNDim) sSq[lane] = sq;
            }
        } else {
            float fmn[16], fmx[16];  // non-colour columns: 0,1 then 5..18
            double cmn[3], cmx[3];
#pragma unroll
            for (int i = 0; i < 16; ++i) { fmn[i] = INFINITY; fmx[i] = -INFINITY; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { cmn[i] = INFINITY; cmx[i] = -INFINITY; }
            double *sStageP = sStage + (wv - NCH) * (kNDim * kCS);
            float vf[16];
            double vd[3];
            auto fetch = [&](int j) {
                if (j < nrun) {
                    const uint32_t off = sOff[j];
#pragma unroll
                    for (int i = 0; i < 16; ++i) vf[i] = p.planes[(uint64_t)(i < 2 ? i : i + 3) * p.plane_stride + off];
#pragma unroll
                    for (int i = 0; i < 3; ++i) vd[i] = p.col_in[(uint64_t)i * p.plane_stride + off];
                }
            };
            fetch((wv - NCH) * kStageChunk + lane);
            for (int r = 0; r < nround; ++r) {
                const int j0 = (r * NP + wv - NCH) * kStageChunk;
                if (j0 + lane < nrun) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        fmn[i] = fminf(fmn[i], vf[i]);
                        fmx[i] = fmaxf(fmx[i], vf[i]);
                        sStageP[(i < 2 ? i : i + 3) * kCS + lane] = (double)vf[i];
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        cmn[i] = fmin(cmn[i], vd[i]);
                        cmx[i] = fmax(cmx[i], vd[i]);
                        sStageP[(kColC + i) * kCS + lane] = vd[i];
                    }
                    if (j0 + lane < S) { // own samples are entries 0..S-1 of the neighbourhood
#pragma unroll
                        for (int i = 0; i < 16; ++i) sOwn[(j0 + lane) * kNDim + (i < 2 ? i : i + 3)] = (double)vf[i];
#pragma unroll
                        for (int i = 0; i < 3; ++i) sOwn[(j0 + lane) * kNDim + kColC + i] = vd[i];
                    }
                }
                fetch(j0 + NP * kStageChunk + lane); // in flight across both barriers, while wave 0 chains
                lds_barrier(); // A
                lds_barrier(); // B
            }
            // this wave's min / max of x per column -> sMM[wv-1][c], sMM[wv-1][19 + c]
            double *mm = sMM + (wv - NCH) * (2 * kNDim);
            float a32[32];
#pragma unroll
            for (int i = 0; i < 16; ++i) { a32[i] = fmn[i]; a32[16 + i] = -fmx[i]; }
            const float rr = xl::reduce32<xl::OpMin>(a32, lane);
            const int slot = xl::slot32(lane);
            if ((lane & 1) == 0) {
                const int i = slot & 15;
                const int col = i < 2 ? i : i + 3;
                if (slot < 16) mm[col] = (double)rr;
                else mm[kNDim + col] = (double)(-rr);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double lo = xl::allreduce<xl::OpMin>(cmn[i]), hi = xl::allreduce<xl::OpMax>(cmx[i]);
                if (lane == 0) { mm[kColC + i] = lo; mm[kNDim + kColC + i] = hi; }
            }
        }
        __syncthreads();
        if (tid < kNDim) { // wave 0 holds the sums, wave 1 left the sums of squares in LDS
            const double dn = (double)n;
            const double mean = chain_acc / dn;                       // ops.h:123
            double sd = sqrt(sSq[tid] / dn - mean * mean);            // ops.h:141
            if (p.policy == RPF_DEGEN_EPS && isnan(sd)) sd = 0.0;
            sStat[tid] = mean;
            sStat[kNDim + tid] = sd;
            if (p.dbg.mean) p.dbg.mean[pix * kNDim + tid] = mean;
            if (p.dbg.stddev) p.dbg.stddev[pix * kNDim + tid] = sd;
        }
        if (tid < 2 * kNDim) { // min over the producers (slots 0..18), max (slots 19..37)
            double v = sMM[tid];
#pragma unroll
            for (int w = 1; w < NP; ++w) v = tid < kNDim ? fmin(v, sMM[w * 2 * kNDim + tid]) : fmax(v, sMM[w * 2 * kNDim + tid]);
            sStat[2 * kNDim + tid] = v;
        }
        __syncthreads();
    } else {
        float fmn[16], fmx[16];  // non-colour columns: 0,1 then 5..18
        double cmn[3], cmx[3];
#pragma unroll
        for (int i = 0; i < 16; ++i) { fmn[i] = INFINITY; fmx[i] = -INFINITY; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { cmn[i] = INFINITY; cmx[i] = -INFINITY; }
        {
            double acc = 0.0;
            const int myc = lane & 31;
            const bool chain = myc < kNDim;
            const bool is_sq = lane >= 32;
            const int nrun = (p.stage_mask & 1) ? n : 0;
            float vf[16];
            double vd[3];
            auto fetch = [&](int j) {
                if (j < nrun) {
                    const uint32_t off = sOff[j];
#pragma unroll
                    for (int i = 0; i < 16; ++i) vf[i] = p.planes[(uint64_t)(i < 2 ? i : i + 3) * p.plane_stride + off];
#pragma unroll
                    for (int i = 0; i < 3; ++i) vd[i] = p.col_in[(uint64_t)i * p.plane_stride + off];
                }
            };
            fetch(lane);
            for (int j0 = 0; j0 < nrun; j0 += kStageChunk) {
                const int cnt = min(kStageChunk, nrun - j0);
                if (lane < cnt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        fmn[i] = fminf(fmn[i], vf[i]);
                        fmx[i] = fmaxf(fmx[i], vf[i]);
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        cmn[i] = fmin(cmn[i], vd[i]);
                        cmx[i] = fmax(cmx[i], vd[i]);
                    }
                    if (j0 + lane < S) { // own samples are entries 0..S-1 of the neighbourhood
#pragma unroll
                        for (int i = 0; i < 16; ++i) sOwn[(j0 + lane) * kNDim + (i < 2 ? i : i + 3)] = (double)vf[i];
#pragma unroll
                        for (int i = 0; i < 3; ++i) sOwn[(j0 + lane) * kNDim + kColC + i] = vd[i];
                    }
                }
                // the 64 gathered samples go through the LDS staging buffer 32 at a time ([19][33] doubles: 5 KiB)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int cnth = min(kStageHalf, cnt - hf * kStageHalf); // wave-uniform, may be <= 0
                    const int t = lane - hf * kStageHalf;
                    if (t >= 0 && t < cnth) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) sStage[(i < 2 ? i : i + 3) * (kStageHalf + 1) + t] = (double)vf[i];
#pragma unroll
                        for (int i = 0; i < 3; ++i) sStage[(kColC + i) * (kStageHalf + 1) + t] = vd[i];
                    }
                    wsync();
                    if (hf == 1) fetch(j0 + kStageChunk + lane); // next chunk's gathers overlap the serial chains below
                    if (chain && cnth > 0) {
                        const double *src = sStage + myc * (kStageHalf + 1);
                        if (cnth == kStageHalf) { // full half: LDS reads issue 16 at a time, only the adds are serial
#pragma unroll
                            for (int h = 0; h < kStageHalf; h += 16) {
                                double v[16];
#pragma unroll
                                for (int q = 0; q < 16; ++q) v[q] = src[h + q];
#pragma unroll
                                for (int q = 0; q < 16; ++q) v[q] = is_sq ? v[q] * v[q] : v[q]; // ops.h:138 multiplyArrays (branch-free)
#pragma unroll
                                for (int q = 0; q < 16; ++q) acc = acc + v[q];          // ops.h:121 / 138 sumArrays
                            }
                        } else if (!is_sq) {
                            for (int q = 0; q < cnth; ++q) acc = acc + src[q];
                        } else {
                            for (int q = 0; q < cnth; ++q) { const double v = src[q]; acc = acc + v * v; }
                        }
                    }
                    wsync();
                }
            }
            const double sq = __shfl(acc, (lane & 31) + 32, 64);
            const double dn = (double)n;
            const double mean = acc / dn;                      // ops.h:123
            double sd = sqrt(sq / dn - mean * mean);           // ops.h:141
            if (p.policy == RPF_DEGEN_EPS && isnan(sd)) sd = 0.0;
            if (lane < kNDim) {
                sStat[lane] = mean;
                sStat[kNDim + lane] = sd;
                if (p.dbg.mean) p.dbg.mean[pix * kNDim + lane] = mean;
                if (p.dbg.stddev) p.dbg.stddev[pix * kNDim + lane] = sd;
            }
            // wave min / max of x per column -> sStat[38 + c], sStat[57 + c]
            {
                float a32[32];
#pragma unroll
                for (int i = 0; i < 16; ++i) { a32[i] = fmn[i]; a32[16 + i] = -fmx[i]; }
                const float r = xl::reduce32<xl::OpMin>(a32, lane);
                const int slot = xl::slot32(lane);
                if ((lane & 1) == 0) {
                    const int i = slot & 15;
                    const int col = i < 2 ? i : i + 3;
                    if (slot < 16) sStat[2 * kNDim + col] = (double)r;
                    else sStat[3 * kNDim + col] = (double)(-r);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double lo = xl::allreduce<xl::OpMin>(cmn[i]), hi = xl::allreduce<xl::OpMax>(cmx[i]);
                    if (lane == 0) { sStat[2 * kNDim + kColC + i] = lo; sStat[3 * kNDim + kColC + i] = hi; }
                }
            }
            wsync();
        }

    }

    // ---------------- stage 3a: normalise, bin ids: bins_stage() above --------------------------
    const int B = max(1, (int)sqrt((double)n)); // mi.cpp:54
    const int kdyn = (n + kWave - 1) / kWave;   // wave-uniform: sample slots kk < kdyn exist
    double *sCK = reinterpret_cast<double *>(sPairF); // [19][5] column constants (the pair sums are not live before 3b)
    if (tid < kNDim) column_constants(sStat, sCK, tid);
    bsync();
    if constexpr (K <= 8) {
        switch (kdyn) {
        case 1: bins_stage<1, KW, PACK5>(p, sStat, sOff, sBinW, lane, n, B, sCK); break;
        case 2: if constexpr (K >= 2) bins_stage<2, KW, PACK5>(p, sStat, sOff, sBinW, lane, n, B, sCK); break;
        case 3: if constexpr (K >= 3) bins_stage<3, KW, PACK5>(p, sStat, sOff, sBinW, lane, n, B, sCK); break;
        case 4: if constexpr (K >= 4) bins_stage<4, KW, PACK5>(p, sStat, sOff, sBinW, lane, n, B, sCK); break;
        case 5: if constexpr (K >= 5) bins_stage<5, KW, PACK5>(p, sStat, sOff, sBinW, lane, n, B, sCK); break;
        case 6: if constexpr (K >= 6) bins_stage<6, KW, PACK5>(p, sStat, sOff, sBinW, lane, n, B, sCK); break;
        case 7: if constexpr (K >= 7) bins_stage<7, KW, PACK5>(p, sStat, sOff, sBinW, lane, n, B, sCK); break;
        default: if constexpr (K >= 8) bins_stage<8, KW, PACK5>(p, sStat, sOff, sBinW, lane, n, B, sCK); break;
        }
    } else {
        bins_stage<K, KW, PACK5, NW>(p, sStat, sOff, sBinW, lane, n, B, sCK, wv);
    }
    bsync();
    if (p.dbg.bin_hash != nullptr && tid < kNDim) { // debug only: hash in sample order j = lane + 64*slot
        uint32_t h = 2166136261u;
        for (int j = 0; j < n; ++j) h = fnv1a_u16(h, bin_of_sample<KW, PACK5>(sBinW, tid, j));
        p.dbg.bin_hash[pix * kNDim + tid] = h;
    }
    // ---------------- stage 3b: histograms -> mutual information: mi_stage() above ------------
    if (p.stage_mask & 4) {
        const uint64_t *dtab = T_IN_LDS ? sD : p.dfix;
        if constexpr (K <= 8) {
            // one straight-line instantiation per number of occupied sample slots (and per number of 1-KiB clearing
            // stores): no branch sits between the LDS operations of a histogram group, so they pipeline under
            // counted lgkmcnt waits
#define RPF_MI_CASE(KD_)                                                                                     \
    if constexpr (K >= KD_) {                                                                                \
        if constexpr (KD_ == 1) mi_stage_tiny<KW, PACK5>(sBinW, sHist, p.tfix, sHXf, sPairF, lane, n, B); /* B*B <= 64 */ \
        else if constexpr (KD_ == 2) mi_stage_deep<KD_, KW, 1, PACK5>(sBinW, sHist, dtab, sHXf, sPairF, lane, n, B); /* B*B <= 121 */ \
        else if (B * B <= 256) mi_pm<KD_, KW, 1, PACK5, !T_IN_LDS>(sBinW, sHist, dtab, sD, p.tfix, sHXf, sPairF, lane, n, B); \
        else if (B * B <= 320) mi_pm<KD_, KW, 3, PACK5, !T_IN_LDS>(sBinW, sHist, dtab, sD, p.tfix, sHXf, sPairF, lane, n, B); \
        else if (B * B <= 384) mi_pm<KD_, KW, 4, PACK5, !T_IN_LDS>(sBinW, sHist, dtab, sD, p.tfix, sHXf, sPairF, lane, n, B); \
        else mi_pm<KD_, KW, 2, PACK5, !T_IN_LDS>(sBinW, sHist, dtab, sD, p.tfix, sHXf, sPairF, lane, n, B);               \
    }
            switch (kdyn) {
            case 1: RPF_MI_CASE(1) break;
            case 2: RPF_MI_CASE(2) break;
            case 3: RPF_MI_CASE(3) break;
            case 4: RPF_MI_CASE(4) break;
            case 5: RPF_MI_CASE(5) break;
            case 6: RPF_MI_CASE(6) break;
            case 7: RPF_MI_CASE(7) break;
            default: RPF_MI_CASE(8) break;
            }
#undef RPF_MI_CASE
        } else {
            mi_stage<K, KW, PACK5, NW>(sBinW, sHist, dtab, sHXf, sPairF, lane, n, B, wv);
        }
    }
    bsync();
    {
        const int64_t TNf = (int64_t)p.tfix[n];
        const double dn = (double)n;
        for (int pr = tid; pr < kNPair; pr += kThreads) {
            const int ca = c_pairs.a[pr], cb = c_pairs.b[pr];
            int64_t f = TNf + (int64_t)sPairF[pr] - (int64_t)sHXf[ca] - (int64_t)sHXf[cb];
            // Exactly independent histograms (J_ij N == hx_i hy_j on every occupied cell): the reference's terms are
            // p log(1.0) == 0 exactly whenever its quotients are exact (always for N a power of two, e.g. N == S), and
            // that exact zero decides 0/0 at rpf.cpp:465/470.  The table entries are rounded to 2^-44, so such a sum
            // lands within (#terms / 2) units of zero instead of on it: snap it.  (A non-zero N*MI is >= 1/(2 N^2 E),
            // orders of magnitude above the bound for all but contrived N > 1700 tables.)
            const int64_t zero_band = ((int64_t)B * B + 2 * B + 1) / 2 + 1;
            if (f <= zero_band && f >= -zero_band) f = 0;
            const double mi = ldexp((double)f, -kTFixBits) / dn;
            sMI[pr] = mi;
            if (p.dbg.mi) p.dbg.mi[pix * kNPair + pr] = mi;
        }
    }
    bsync();

    // ---------------- stage 3c: alpha, beta, W_r_c (rpf.cpp:444-487), lane-parallel -----------------
    // lane k < 12 owns feature k, lane c < 3 also owns colour channel c; values meet through a scratch area in the
    // (now dead) histogram buffer: Drf[12] | Drc,Dpc,Dfc [9] | alpha[3] | beta[12] | wrc | coef[17]
    double *sT = reinterpret_cast<double *>(sHist0);
    double *sDrf = sT, *sD9 = sT + 12, *sAlpha = sT + 24, *sBeta = sT + 28, *sWrc = sT + 40, *sCoef = sT + 44;
    int *sBadFlag = reinterpret_cast<int *>(sT + 64); // NW > 1: did any wave of the pixel see a NaN colour
    if (NW == 1 || wv == 0) {
        const int k = min(lane, kNFeat - 1), c = min(lane, 2);
        const double Drf = 0.0 + sMI[k * 4 + 0] + sMI[k * 4 + 1]; // rpf.cpp:421
        const double Dpf = 0.0 + sMI[k * 4 + 2] + sMI[k * 4 + 3]; // rpf.cpp:425
        const double Dcf = 0.0 + sMI[52 + k] + sMI[68 + k] + sMI[84 + k]; // PAPER numerator: sum_c MI(c_c, f_k)
        const int base = 48 + c * 16;
        const double Drc = 0.0 + sMI[base + 0] + sMI[base + 1];   // rpf.cpp:432
        const double Dpc = 0.0 + sMI[base + 2] + sMI[base + 3];   // rpf.cpp:436
        double Dfc = 0.0;
#pragma unroll
        for (int j = 0; j < kNFeat; ++j) Dfc += sMI[base + 4 + j]; // rpf.cpp:440
        wsync(); // every lane has read its MI values; the scratch area may now be written
        if (lane < kNFeat) sDrf[lane] = Drf;
        if (lane < 3) { sD9[lane] = Drc; sD9[3 + lane] = Dpc; sD9[6 + lane] = Dfc; }
        wsync();
        double D_f_c = 0.0, D_r_c = 0.0, D_p_c = 0.0;             // rpf.cpp:449-456
#pragma unroll
        for (int i = 0; i < 3; ++i) { D_f_c += sD9[6 + i]; D_r_c += sD9[i]; D_p_c += sD9[3 + i]; }
        const double e = (p.policy == RPF_DEGEN_EPS) ? p.eps : 0.0;
        const double den = D_f_c + D_r_c + D_p_c + e;
        double wsum = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) wsum += sD9[i] / (sD9[i] + sD9[3 + i] + e); // rpf.cpp:470, 485
        const double wrc = wsum / 3;                                           // rpf.cpp:487
        const double alpha_c = 1 - Drc / (Drc + Dpc + e);                      // rpf.cpp:470, 475
        double num; // what rpf.cpp:464 reads as D_f_ck[k] (3-element array indexed to 11: SURVEY F3)
        if (p.beta_map == RPF_BETA_PAPER) num = Dcf;
        else if (p.beta_map == RPF_BETA_REF_GCC11_O2) num = k < 3 ? sD9[6 + c] : (k < 8 ? 0.0 : sDrf[max(k - 8, 0)]);
        else num = k < 3 ? sD9[6 + c] : (k < 4 ? 0.0 : sDrf[max(k - 4, 0)]);
        const double Wc = num / den;                       // rpf.cpp:464
        const double Wr = Drf / (Drf + Dpf + e);           // rpf.cpp:465
        const double beta_k = (1 - Wr) * Wc;               // rpf.cpp:479
        if (lane < kNFeat) {
            sBeta[lane] = beta_k;
            if (p.dbg.beta) p.dbg.beta[pix * kNFeat + lane] = beta_k;
        }
        if (lane < 3) {
            sAlpha[lane] = alpha_c;
            if (p.dbg.alpha) p.dbg.alpha[pix * 3 + lane] = alpha_c;
        }
        if (lane == 0) {
            sWrc[0] = wrc;
            if (NW > 1) sBadFlag[0] = 0;
            if (p.dbg.wrc) p.dbg.wrc[pix] = wrc;
        }
        wsync();
    }
    if constexpr (NW > 1) __syncthreads();
    const double wrc = sWrc[0];

    // ---------------- stage 4: weights and blend (rpf.cpp:627-717) ------------------------------
    // z-space set-up shared by both weight modes (dead LDS regions: x-min/x-max slots -> M and 1/SD of the 17 weighted
    // columns; bin ids -> the own samples' rows).  With z = (x-M)/SD the exponent of w_ij is
    //     sum_k cz_k (z_ik - z_jk)^2 = A_i + B_j + sum_k u_ik z_jk,   cz_k = weight_k / (2 sigma^2),
    //     A_i = sum_k cz_k z_ik^2,  B_j = sum_k cz_k z_jk^2,  u_ik = -2 cz_k z_ik :
    // 17 FMAs per pair instead of 17 x (sub, mul, fma).  z is O(1) by construction, so the cancellation in
    // A + B - 2 dot costs ~1e-15 * cz in E (cz <= ~1e5): far below what exp() resolves.
    double *sFastM = sStat + 2 * kNDim;  // the x-min / x-max slots are dead after stage 3a
    double *sFastI = sFastM + 17;
    float *sFastZ = reinterpret_cast<float *>(sBinW);   // FAST: own z rows, fp32 [S][20]
    double *sOwnU = reinterpret_cast<double *>(sBinW);  // !FAST: own rows, fp64 [S][18] = u_i0..u_i16, A_i
    float coefz[17];
    {
        const double sigma_c2 = p.seed * p.seed / (1 - wrc) / (1 - wrc);     // rpf.cpp:662
        const double inv2sc = 1.0 / (2 * sigma_c2), inv2sp = 1.0 / (2 * (p.sigma_p * p.sigma_p)); // rpf.cpp:664,668
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const double wkk = k < 2 ? 1.0 : (k < 5 ? sAlpha[k >= 2 && k < 5 ? k - 2 : 0] : sBeta[k >= 5 ? k - 5 : 0]);
            coefz[k] = (float)(wkk * (k < 2 ? inv2sp : inv2sc));
        }
        bsync(); // every wave has read alpha / beta: the coefficient slots may be overwritten
        if (tid < 17) {
            const int col = lane < 5 ? lane : lane + 2;
            const double sd = sStat[kNDim + col];
            const double wkk = lane < 2 ? 1.0 : (lane < 5 ? sAlpha[max(lane - 2, 0)] : sBeta[max(lane - 5, 0)]);
            sFastM[lane] = sStat[col];
            sFastI[lane] = (sd == 0.0) ? 0.0 : 1.0 / sd; // SD == 0 normalises to z == 0 (ops.h:48)
            sCoef[lane] = wkk * (lane < 2 ? inv2sp : inv2sc); // cz_k (overwrites the raw-space coefficient slot)
        }
        bsync();
        if constexpr (FAST) {
            for (int t = tid; t < S * 20; t += kThreads) {
                const int i = t / 20, k = t - i * 20;
                float z = 0.f;
                if (k < 17) z = (float)((sOwn[i * kNDim + (k < 5 ? k : k + 2)] - sFastM[k]) * sFastI[k]);
                sFastZ[t] = z;
            }
        } else {
            for (int i = tid; i < S; i += kThreads) {
                double A = 0.0;
#pragma unroll
                for (int k = 0; k < 17; ++k) {
                    const double z = (sOwn[i * kNDim + (k < 5 ? k : k + 2)] - sFastM[k]) * sFastI[k];
                    const double t = sCoef[k] * z;
                    A = fma(t, z, A);
                    sOwnU[i * 18 + k] = -2.0 * t;
                }
                sOwnU[i * 18 + 17] = A;
            }
        }
        bsync();
    }
    bool bad = false;
    // own samples weighted per sweep over the neighbourhood (register budget: 4 at 3 waves/SIMD; the multi-wave kernels
    // run 1-2 waves/SIMD and take 8, halving the per-sample set-up, and gather one sample slot ahead)
    constexpr int kOwnBlock = (NW == 1 && K == 13) ? 16 : 8;
    double *sRed = sMI + 16 * wv; // 16 sums of a sweep meet here (the MI values are dead; one slot set per wave)
    // NW > 1: the sweeps are dealt round-robin to the waves of the pixel
    for (int i0 = kOwnBlock * wv; i0 < ((p.stage_mask & 8) ? S : 0); i0 += kOwnBlock * NW) {
        double sw[kOwnBlock], s0[kOwnBlock], s1[kOwnBlock], s2[kOwnBlock];
#pragma unroll
        for (int ii = 0; ii < kOwnBlock; ++ii) { sw[ii] = 0.0; s0[ii] = 0.0; s1[ii] = 0.0; s2[ii] = 0.0; }
        // raw values of the lane's next neighbourhood sample are gathered while the current one is weighted
        float pf[14];  // columns 0,1 (pFilm) and 7..18 (features)
        double pc[3];  // colours
        auto fetch17 = [&](int j) {
            if (j < n) {
                const uint32_t off = sOff[j];
#pragma unroll
                for (int k = 0; k < 14; ++k) pf[k] = p.planes[(uint64_t)(k < 2 ? k : k + 5) * p.plane_stride + off];
#pragma unroll
                for (int k = 0; k < 3; ++k) pc[k] = p.col_in[(uint64_t)k * p.plane_stride + off];
            }
        };
        if constexpr (NW > 1) fetch17(lane);
        if constexpr (!FAST) {
#pragma unroll 1
            for (int kk = 0; kk < K; ++kk) {
                const int j = lane + kWave * kk;
                if (j >= n) break;
                if constexpr (NW == 1) fetch17(j); // no register double-buffering: three waves per SIMD cover the gather latency
                double zj[17], cj[3];
                double Bj = 0.0;
#pragma unroll
                for (int k = 0; k < 17; ++k) {
                    const double xv = k < 2 ? (double)pf[k] : (k < 5 ? pc[k < 5 && k >= 2 ? k - 2 : 0] : (double)pf[k >= 5 ? k - 3 : 0]);
                    const double z = (xv - sFastM[k]) * sFastI[k];
                    zj[k] = z;
                    Bj = fma(sCoef[k] * z, z, Bj);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) cj[k] = pc[k];
                if constexpr (NW > 1) fetch17(j + kWave); // the next slot's gathers fly while this one is weighted
                // straight-line over the kOwnBlock own samples (a missing one re-reads the last row and its weight is
                // forced to 0) so their dependent chains interleave; the 17-term dot product runs as four partial sums
                // (skipping exp() where every lane's exponent underflows was tried twice -- a wave-uniform branch per own
                // sample, and one per sweep step over stored exponents: both spill 24 .. 140 registers at three waves per
                // SIMD and run slower, 75.9 vs 69.8 ms)
#pragma unroll
                for (int ii = 0; ii < kOwnBlock; ++ii) {
                    const int i = i0 + ii;
                    const bool live = i < S; // wave-uniform
                    const double2 *ui2 = reinterpret_cast<const double2 *>(sOwnU + min(i, S - 1) * 18);
                    double ui[18]; // the own row as nine 16-byte broadcast reads
#pragma unroll
                    for (int q = 0; q < 9; ++q) { const double2 v = ui2[q]; ui[2 * q] = v.x; ui[2 * q + 1] = v.y; }
                    double e0 = ui[17] + Bj, e1 = 0.0, e2 = 0.0, e3 = 0.0;
#pragma unroll
                    for (int k = 0; k < 16; k += 4) {
                        e0 = fma(ui[k], zj[k], e0);
                        e1 = fma(ui[k + 1], zj[k + 1], e1);
                        e2 = fma(ui[k + 2], zj[k + 2], e2);
                        e3 = fma(ui[k + 3], zj[k + 3], e3);
                    }
                    e0 = fma(ui[16], zj[16], e0);
                    const double E = (e0 + e1) + (e2 + e3);
                    double w = exp(-E);                         // rpf.cpp:667-670
                    w = live ? w : 0.0;
                    sw[ii] += w;                                // rpf.cpp:691
                    s0[ii] = fma(w, cj[0], s0[ii]);             // rpf.cpp:692 (raw neighbourhood colours)
                    s1[ii] = fma(w, cj[1], s1[ii]);
                    s2[ii] = fma(w, cj[2], s2[ii]);
                }
            }
        } else {
            // opt-in RPF_FLAG_FAST_WEIGHTS: per-pair arithmetic in fp32 on z-space values (x-M)/SD that are formed in
            // fp64 (so large world coordinates do not cancel in fp32), v_exp_f32, fp64 accumulation of the sums.
            // Filtered colours differ from the fp64 path by ~1e-6 relative (bar: 1e-4).
#pragma unroll 1
            for (int kk = 0; kk < K; ++kk) {
                const int j = lane + kWave * kk;
                if (j >= n) break;
                if constexpr (NW == 1) fetch17(j);
                float zj[17];
                double cj[3];
                zj[0] = (float)(((double)pf[0] - sFastM[0]) * sFastI[0]);
                zj[1] = (float)(((double)pf[1] - sFastM[1]) * sFastI[1]);
#pragma unroll
                for (int k = 0; k < 3; ++k) { cj[k] = pc[k]; zj[2 + k] = (float)((pc[k] - sFastM[2 + k]) * sFastI[2 + k]); }
#pragma unroll
                for (int k = 0; k < 12; ++k) zj[5 + k] = (float)(((double)pf[2 + k] - sFastM[5 + k]) * sFastI[5 + k]);
                if constexpr (NW > 1) fetch17(j + kWave);
#pragma unroll
                for (int ii = 0; ii < kOwnBlock; ++ii) {
                    const int i = i0 + ii;
                    if (i < S) {
                        const float *zi = sFastZ + i * 20; // rows padded to 20 floats: five 16-byte reads
                        float zo[20];
#pragma unroll
                        for (int q4 = 0; q4 < 5; ++q4) {
                            const float4 v = reinterpret_cast<const float4 *>(zi)[q4];
                            zo[4 * q4] = v.x; zo[4 * q4 + 1] = v.y; zo[4 * q4 + 2] = v.z; zo[4 * q4 + 3] = v.w;
                        }
                        float E = 0.f;
#pragma unroll
                        for (int k = 0; k < 17; ++k) {
                            const float d = zo[k] - zj[k];
                            E = fmaf(d * d, coefz[k], E);
                        }
                        const double w = (double)__expf(-E);
                        sw[ii] += w;
                        s0[ii] = fma(w, cj[0], s0[ii]);
                        s1[ii] = fma(w, cj[1], s1[ii]);
                        s2[ii] = fma(w, cj[2], s2[ii]);
                    }
                }
            }
        }
        // 16 wave sums by one transposed butterfly, gathered through LDS: [0..3] sum w, [4..7] r, [8..11] g, [12..15] b
#pragma unroll
        for (int hb = 0; hb < kOwnBlock; hb += 4) {
            double ga[16];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) { ga[ii] = sw[hb + ii]; ga[4 + ii] = s0[hb + ii]; ga[8 + ii] = s1[hb + ii]; ga[12 + ii] = s2[hb + ii]; }
            const double ta = xl::reduce16<xl::OpSum>(ga, lane);
            wsync();
            if ((lane & 3) == 0) sRed[xl::slot16(lane)] = ta;
            wsync();
            if (lane < 12) {
                const int ii = lane / 3, k = lane % 3;
                const int i = i0 + hb + ii;
                if (i < S) {
                    double prime = sRed[4 * (k + 1) + ii] / sRed[ii];     // rpf.cpp:700
                    if (isnan(prime)) {                                    // rpf.cpp:702: the reference exits here
                        bad = true;
                        if (p.policy == RPF_DEGEN_EPS) prime = sOwn[i * kNDim + kColC + k];
                    }
                    p.col_out[(uint64_t)k * p.plane_stride + pix * S + i] = prime;
                }
            }
        }
    }
    if constexpr (NW > 1) {
        if (__any(bad) && lane == 0) sBadFlag[0] = 1;
        __syncthreads();
        if (tid == 0 && sBadFlag[0]) {
            atomicAdd(&p.status[0], 1);
            atomicMin(&p.status[1], (int)pix);
        }
    } else if (__any(bad) && lane == 0) {
        atomicAdd(&p.status[0], 1);
        atomicMin(&p.status[1], (int)pix);
    }
}

// ---- neighbourhood-size binning -----------------------------------------------------------------------------------
// The cost and the LDS footprint of a pixel are set by its neighbourhood size N, and N is data dependent: box*box*S is
// only its ceiling (path-traced buffers keep little more than the S own samples).  When the ceiling is above
// what the one-wave kernels hold, N is counted first (stage 1b's test without the list), the pixels are dealt into one
// list per kernel family, and every family filters its own list with LDS sized for ITS capacity.
__global__ __launch_bounds__(256) void nbhd_count_kernel(PassParams p) {
    const int lane = threadIdx.x & (kWave - 1);
    const int W = p.W, H = p.H, S = p.S, b = p.b;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= (int64_t)(p.row_end - p.row_begin) * W) return; // wave-uniform
    const int y = p.row_begin + (int)(q / W), x = (int)(q % W);
    const uint64_t HW = (uint64_t)H * W, pix = (uint64_t)y * W + x;
    const int x0 = max(x - b, 0), x1 = min(x + b, W - 1);
    const int y0 = max(y - b, 0), y1 = min(y + b, H - 1);
    const int nyv = y1 - y0 + 1;
    const int centre_rank = (x - x0) * nyv + (y - y0);
    const int ncand = ((x1 - x0 + 1) * nyv - 1) * S;
    const uint32_t magic_S = div_magic((uint32_t)S), magic_ny = div_magic((uint32_t)nyv);
    double m12[kNFeat], lim12[kNFeat];
#pragma unroll
    for (int k = 0; k < kNFeat; ++k) {
        m12[k] = p.pmean[(uint64_t)k * HW + pix];
        lim12[k] = p.pstd[(uint64_t)k * HW + pix] * 3.0; // multiplyArray(std, 3), rpf.cpp:579
    }
    constexpr int kPF1 = 3;
    float fb[kPF1][kNFeat];
    auto issue1 = [&](int qq, float (&f)[kNFeat]) {
        if (qq < ncand) {
            int cell = (int)div_small((uint32_t)qq, magic_S);
            const int s = qq - cell * S;
            if (cell >= centre_rank) ++cell;          // rpf.cpp:565
            const int ix = (int)div_small((uint32_t)cell, magic_ny), iy = cell - ix * nyv;
            const uint32_t off = (uint32_t)(((uint64_t)(y0 + iy) * W + (x0 + ix)) * S + s);
#pragma unroll
            for (int k = 0; k < kNFeat; ++k) f[k] = p.planes[(uint64_t)(kColF + k) * p.plane_stride + off];
        }
    };
#pragma unroll
    for (int u = 0; u < kPF1; ++u) issue1(u * kWave + lane, fb[u]);
    int n = S;
#pragma unroll 1
    for (int q0 = 0; q0 < ncand; q0 += kWave * kPF1) {
#pragma unroll
        for (int u = 0; u < kPF1; ++u) {
            const int qb = q0 + u * kWave;
            if (qb < ncand) { // wave-uniform
                bool pass = (qb + lane) < ncand;
#pragma unroll
                for (int k = 0; k < kNFeat; ++k) {
                    const double a = fabs((double)fb[u][k] - m12[k]);
                    if (a >= lim12[k]) pass = false;       // allLessThan (ops.h:101-104)
                }
                const unsigned long long mask = __ballot(pass);
                if (p.masks != nullptr && lane == 0) p.masks[pix * p.mask_stride + (uint32_t)(qb >> 6)] = mask;
                n += __popcll(mask);
                issue1(qb + kWave * kPF1 + lane, fb[u]);
            }
        }
    }
    if (lane == 0) p.nbhd[pix] = n;
}

struct ClassCaps { int cap[kNumClasses]; };

// pixels in slab_pixel order -> one list per class (class = first capacity >= N); lists[c][*], counts[c]
__global__ __launch_bounds__(256) void classify_kernel(PassParams p, ClassCaps caps, uint32_t *lists, uint32_t *counts,
                                                       uint64_t list_stride) {
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
    }
#pragma unroll
    for (int c = 0; c < kNumClasses; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (m == 0ull) continue; // wave-uniform
        uint32_t base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&counts[c], (uint32_t)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, kWave);
        if (cls == c) lists[(uint64_t)c * list_stride + base + __popcll(m & ((1ull << lane) - 1ull))] = pix;
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
__global__ __launch_bounds__(256) void colour_from_planes_kernel(const float *planes, double *colour, uint64_t ps,
                                                                  uint64_t e0, uint64_t cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * cnt) return;
    const uint64_t c = i / cnt, e = e0 + (i - c * cnt);
    colour[c * ps + e] = (double)planes[(2 + c) * ps + e];
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

template <int K, bool TL, bool FAST, int NW>
hipError_t launch_filter_inst(const PassParams &p, const LdsLayout &L, unsigned grid, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *)filter_pixel_kernel<K, TL, FAST, NW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((filter_pixel_kernel<K, TL, FAST, NW>), dim3(grid), dim3(64 * NW), L.total, s, p, L);
    return hipGetLastError();
}
template <int K>
hipError_t launch_filter_k(const PassParams &p, const LdsLayout &L, bool t_in_lds, unsigned grid, hipStream_t s) {
    if constexpr (K >= 13) {
        if (L.nw == 4) {
            if (t_in_lds)
                return p.fast_weights ? launch_filter_inst<K, true, true, 4>(p, L, grid, s)
                                      : launch_filter_inst<K, true, false, 4>(p, L, grid, s);
            return p.fast_weights ? launch_filter_inst<K, false, true, 4>(p, L, grid, s)
                                  : launch_filter_inst<K, false, false, 4>(p, L, grid, s);
        }
    }
    if (p.fast_weights) {
        return t_in_lds ? launch_filter_inst<K, true, true, 1>(p, L, grid, s) : launch_filter_inst<K, false, true, 1>(p, L, grid, s);
    }
    return t_in_lds ? launch_filter_inst<K, true, false, 1>(p, L, grid, s) : launch_filter_inst<K, false, false, 1>(p, L, grid, s);
}

} // namespace

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

bool table_in_lds(int S, int nmax, int bmax, const Tuning &tun) {
    // One-wave kernels read the D table through L1: 3 .. 25 KiB of LDS per workgroup buy resident waves, which is what
    // those latency-bound kernels need.  The four-wave kernels run 1-2 workgroups per CU, every table look-up is a
    // round trip none of their few waves can cover, and one copy serves four waves: they keep the table in LDS
    // whenever that costs no resident workgroup (32 spp: 161 -> 206 ms without it).  Option "table_in_lds" overrides.
    if ((uint32_t)nmax * 8u > 65536u) return false;
    if (tun.table_in_lds >= 0) return tun.table_in_lds != 0;
    if (waves_per_pixel(nmax, tun) == 1) return false;
    const uint32_t without = lds_layout(S, nmax, bmax, false, tun).total, with = lds_layout(S, nmax, bmax, true, tun).total;
    return with <= (uint32_t)max_lds_per_block() && (uint32_t)max_lds_per_block() / with == (uint32_t)max_lds_per_block() / without;
}

LdsLayout lds_layout(int S, int nmax, int bmax, bool t_in_lds, const Tuning &tun) {
    LdsLayout L{};
    const int K = samples_per_lane(nmax);
    const uint32_t KW = (uint32_t)pack_words(K);
    uint32_t o = 0;
    L.off_T = o;
    if (t_in_lds) o += align_up((uint32_t)nmax * 8u, 16);
    else if (K <= 8) o += (uint32_t)kDHead * 8u; // the one-wave kernels keep the head of the D table in LDS
    L.off_stat = o; o += align_up(4 * kNDim * 8, 16);
    L.off_hx = o; o += align_up(kNDim * 8, 16);
    L.off_pair = o; o += align_up(kNPair * 8, 16);
    L.off_mi = L.off_pair; // the MI values overwrite the pair sums in place
    L.off_own = o; o += align_up((uint32_t)S * kNDim * 8u, 16);
    L.off_off = o; o += align_up((uint32_t)nmax * 4u, 16);
    L.off_union = o;
    const int nw = waves_per_pixel(nmax, tun);
    L.nw = (uint32_t)nw;
    const uint32_t stage = nw > 1 ? align_up((uint32_t)(nw - 2) * kNDim * (kStageChunk + 1) * 8u, 16) // one chunk per producer wave
                                  : align_up(kNDim * (kStageHalf + 1) * 8, 16);
    const uint32_t bins = K <= 8 ? align_up((uint32_t)kNDim * kWave * 5u, 16)          // 5-bit words + slot-6 bytes
                                 : align_up((uint32_t)kNDim * kWave * KW * 4u, 16);     // 5- or 6-bit fields in KW words
    uint32_t uni = bins > stage ? bins : stage;
    const uint32_t fastz = align_up((uint32_t)S * 18u * 8u, 16);                         // own rows of the weight stage
    if (fastz > uni) uni = fastz;
    o += uni;
    L.off_hist = o;
    uint32_t cells = (uint32_t)bmax * (uint32_t)bmax;
    if (K <= 8 && cells < 512u) cells = 512u; // mi_stage clears with unconditional 1-KiB stores
    if (cells < 256u) cells = 256u;           // the buffer doubles as scratch (stage 1b masks, stage 3c, flags)
    L.hist_stride = align_up(cells * 4u, 16);
    o += L.hist_stride * (uint32_t)nw;        // one histogram buffer per wave of the pixel
    if (tun.lds_pad > 0) o += (uint32_t)tun.lds_pad; // occupancy experiment knob
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
    hipLaunchKernelGGL(pixel_stats_kernel, dim3((unsigned)((pix1 - pix0 + 255) / 256)), dim3(256), 0, s, p, pix0, pix1);
    return hipGetLastError();
}

hipError_t launch_pixel_stats(const PassParams &p, hipStream_t s) { return launch_pixel_stats_rows(p, 0, p.H, s); }

hipError_t launch_filter_pass(const PassParams &p, const Tuning &tun, hipStream_t s, uint32_t *lds_bytes_out) {
    const bool t_in_lds = table_in_lds(p.S, p.nmax, p.bmax, tun);
    const LdsLayout L = lds_layout(p.S, p.nmax, p.bmax, t_in_lds, tun);
    if (lds_bytes_out) *lds_bytes_out = L.total;
    if ((int)L.total > max_lds_per_block()) return hipErrorInvalidValue;
    const int rows_own = p.row_end - p.row_begin;
    if (rows_own <= 0) return hipSuccess;
    if (p.pix_list != nullptr && p.list_count == 0) return hipSuccess;
    const int64_t band = p.pix_list ? (int64_t)((p.list_count + 7u) / 8u)
                                    : (int64_t)((rows_own + 7) / 8) * p.W; // pixels per XCD band (see slab_pixel)
    const unsigned grid = (unsigned)(band * 8);
    switch (samples_per_lane(p.nmax)) {
    case 1: return launch_filter_k<1>(p, L, t_in_lds, grid, s);
    case 2: return launch_filter_k<2>(p, L, t_in_lds, grid, s);
    case 4: return launch_filter_k<4>(p, L, t_in_lds, grid, s);
    case 7: return launch_filter_k<7>(p, L, t_in_lds, grid, s);
    case 13: return launch_filter_k<13>(p, L, t_in_lds, grid, s);
    case 25: return launch_filter_k<25>(p, L, t_in_lds, grid, s);
    case 49: return launch_filter_k<49>(p, L, t_in_lds, grid, s);
    default: return hipErrorInvalidValue;
    }
}

int class_capacity(int c) {
    static const int caps[kNumClasses] = {64, 128, 256, 448, 832, 1600, 3136}; // 64 * {1, 2, 4, 7, 13, 25, 49}
    return caps[c];
}

hipError_t launch_nbhd_count(const PassParams &p, hipStream_t s) {
    const int64_t npix = (int64_t)(p.row_end - p.row_begin) * p.W;
    if (npix <= 0) return hipSuccess;
    hipLaunchKernelGGL(nbhd_count_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_classify(const PassParams &p, uint32_t *lists, uint32_t *counts, hipStream_t s) {
    const int64_t total = (int64_t)((p.row_end - p.row_begin + 7) / 8) * p.W * 8;
    if (total <= 0) return hipSuccess;
    ClassCaps caps;
    for (int c = 0; c < kNumClasses; ++c) caps.cap[c] = class_capacity(c);
    hipLaunchKernelGGL(classify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, caps, lists, counts,
                       (uint64_t)p.H * p.W);
    return hipGetLastError();
}

hipError_t launch_colour_from_planes_span(const float *planes, double *colour, uint64_t ps, uint64_t e0, uint64_t cnt,
                                          hipStream_t s) {
    if (cnt == 0) return hipSuccess;
    hipLaunchKernelGGL(colour_from_planes_kernel, dim3((unsigned)((3 * cnt + 255) / 256)), dim3(256), 0, s, planes,
                       colour, ps, e0, cnt);
    return hipGetLastError();
}

hipError_t launch_colour_from_planes(const float *planes, double *colour, uint64_t ps, hipStream_t s) {
    return launch_colour_from_planes_span(planes, colour, ps, 0, ps, s);
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
