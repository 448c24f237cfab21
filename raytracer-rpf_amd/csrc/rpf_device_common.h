// rpf_device_common.h -- device-side helpers shared by every kernel translation unit (the misc kernels in
// rpf_kernels.hip and the per-layout / per-size-class parts rpf_impl_*.hip, which each include rpf_filter_impl.inc).
// Everything here has internal linkage: each TU carries its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <cstdlib>
#include <type_traits>

#include "rpf_internal.h"
#include "rpf_xlane.h"
#include "rpf_reflog.h"

namespace rpf {

namespace {

constexpr int kDHead = 128;   // entries of the D table the one-wave kernels keep in LDS
constexpr int kTFixBits = 44; // T[k] = k ln k is tabulated as round(T * 2^44): exact integer sums, |T| < 2^15 * 2^44

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

// Predicated 8-byte LDS store with the EXEC masking kept inside one asm block.  Written as `if (pred) *p = v;` the
// large-neighbourhood kernels get a control-flow join after the store, and hipcc 7.2's register allocator has placed
// spill code (scratch stores / v_accvgpr_write) at the top of such a join block IN FRONT OF the s_or that re-enables
// the masked lanes: those lanes lost live registers (scripts/check_spills.py).  With no compiler-visible branch there
// is no join block to put spill code into.
__device__ __forceinline__ void lds_store_u64_if(bool pred, uint64_t *p, uint64_t v) {
    const uint64_t mask = __builtin_amdgcn_ballot_w64(pred);
    const uint32_t addr = (uint32_t)reinterpret_cast<uintptr_t>(p); // LDS byte offset = low half of the flat address
    uint64_t saved;
    // s_and_saveexec WRITES SCC, and the statement must say so.  Rounds 2-3 shipped it without the clobber: hipcc then keeps a
    // compare alive across the store wherever that suits its schedule -- seen as wrong pair sums of every histogram group of an
    // anchor after its first in the one-wave K = 25 / 49 kernels, and as run-to-run differences with a table head at K = 13
    // (both layouts), while the instantiations that shipped happened to compare after the store (scripts/r03_diag_mi.py,
    // r03_bisect.sh, r03_bisect2.sh; gpurun_out/r3v: 73 of 99 pixels wrong without, 0 with the clobber).
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b64 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "s"(mask), "v"(addr), "v"(v) : "memory", "scc");
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
        // whole 1-KiB wave stores, the same number in every lane: the histogram buffers are padded to a multiple of
        // 1 KiB (lds_layout).  A per-lane bound makes this a divergent loop, and at the register pressure of the
        // large-K kernels hipcc parked spill copies behind its exit, where EXEC is empty (check_spills.py).
        const int nit = __builtin_amdgcn_readfirstlane((cells + kWave * 4 - 1) / (kWave * 4));
        // (no loop vectoriser here: it turned the 16-byte store per row into four 4-byte stores -- 24 LDS instructions per
        // histogram of the 32-spp class instead of 6)
#pragma clang loop vectorize(disable) interleave(disable)
        for (int it = 0; it < nit; ++it) *reinterpret_cast<uint4 *>(h + (it * kWave + lane) * 4) = make_uint4(0u, 0u, 0u, 0u);
    }
}

// ---- bin-id storage of one pixel ---------------------------------------------------------------------------
// PACK5 names the packing scheme (an int, historically a bool):
//   1  kernels with K <= 8 (B = floor(sqrt(N)) <= 22): sample slots 0..5 of a lane are 5-bit fields of ONE 32-bit
//      word per (column, lane) -- [19][64] words -- and slot 6 is a byte in a side array [19][64] behind them:
//      6 KiB per pixel, which is what lets 12 single-wave workgroups share a CU's LDS
//   7  K = 13 (B <= 28): slots 0..11 as 5-bit fields of TWO words per (column, lane), slot 12 a byte in a side array
//      [19][64] behind them (13 x 5 bits are one bit more than two words hold; a third word per (column, lane) was 3.6 KiB
//      per pixel and, with the table head the kernel no longer keeps, the difference between six and seven resident
//      workgroups per CU -- this kernel's time goes as 1 / workgroups)
//   5  other K <= 17 (B <= 32): 5-bit fields, six per word, KW = ceil(K/6) words per (column, lane)
//   6  larger K (B <= 64): 6-bit fields, five per word, KW = ceil(K/5) words per (column, lane)
__host__ __device__ constexpr int pack_scheme(int K) { return K <= 8 ? 1 : (K == 13 ? 7 : (K <= 17 ? 5 : 6)); }
__host__ __device__ constexpr int pack_words(int K) { return K <= 8 ? 1 : (K == 13 ? 2 : (K <= 17 ? (K + 5) / 6 : (K + 4) / 5)); }
// bytes of packed bin ids per (column, lane)
__host__ __device__ constexpr int pack_bytes(int K) { return K <= 8 ? 5 : (K == 13 ? 9 : 4 * pack_words(K)); }

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
        } else if constexpr (PACK5 == 7) {
            if (kk < 12) w[kk / 6] |= bin << (5 * (kk % 6)); else b6 = bin;
        } else {
            w[kk / SPW] |= bin << (BITS * (kk % SPW));
        }
    }
    __device__ __forceinline__ uint32_t get(int kk) const {
        if constexpr (PACK5 == 1) return kk < 6 ? ((w[0] >> (5 * kk)) & 31u) : b6;
        else if constexpr (PACK5 == 7) return kk < 12 ? ((w[kk / 6] >> (5 * (kk % 6))) & 31u) : b6;
        else return (w[kk / SPW] >> (BITS * (kk % SPW))) & ((1u << BITS) - 1u);
    }
};
// floor(n / d) for n * d < 2^32 with the precomputed M = floor((2^32 - 1) / d) + 1: one v_mul_hi_u32 instead of the
// ~25-instruction sequence hipcc emits for an integer division by a run-time divisor (exact: n (M d - 2^32) < 2^32)
__device__ __forceinline__ uint32_t div_magic(uint32_t d) { return 0xFFFFFFFFu / d + 1u; }
__device__ __forceinline__ uint32_t div_small(uint32_t n, uint32_t M) { return M ? __umulhi(n, M) : n; } // M == 0: d == 1

// XCD- and L2-aware pixel order of a slab.  Blocks with equal (blockIdx % 8) share an XCD and its 4 MiB L2: XCD r
// filters one contiguous band of rows, and walks it in vertical strips of p.strip_w pixels (row by row inside a strip),
// so the box-row window data of the pixels in flight on the XCD (and the rows shared with the next strip row) stay
// L2-resident instead of being re-fetched once per image row.  The strip width is set by the host so that `box` rows of
// a strip (+ halo columns) at ~88 B per sample stay well inside the L2: 128 px at 8 spp, 32 px at 32 spp (a 128-px strip
// at 32 spp fetched 7.8x the compulsory bytes).  (r, ql) -> pixel; false = no such pixel.
__device__ __forceinline__ bool slab_pixel(const PassParams &p, int r, int64_t ql, int &x, int &y) {
    const uint32_t kStripW = (uint32_t)p.strip_w;
    const int W = p.W;
    const int rows_own = p.row_end - p.row_begin;
    const int rows_band = (rows_own + 7) / 8;
    const int band_row0 = r * rows_band;
    const int band_rows = min(rows_band, rows_own - band_row0);
    if (band_rows <= 0) return false;
    if (ql >= (int64_t)band_rows * W) return false;
    const uint32_t q = (uint32_t)ql;                       // band_rows * W < 2^32 (host-checked: W*H*S < 2^32)
    const uint32_t full_strips = (uint32_t)W / kStripW;
    const uint32_t strip_px = kStripW * (uint32_t)band_rows;
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

} // namespace
} // namespace rpf
