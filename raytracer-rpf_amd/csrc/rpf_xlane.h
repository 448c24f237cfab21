// rpf_xlane.h -- wave64 cross-lane reductions for gfx950 that never touch LDS.
//
// hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS-crossbar instruction); the fused RPF kernel is bound by
// the LDS pipe (histogram atomics), so every reduction here is built from VALU data movers instead:
//   * across the two 32-lane halves:  v_permlane32_swap_b32   (dst' = {dst.lo, src.lo}, src' = {dst.hi, src.hi})
//   * across adjacent 16-lane rows:   v_permlane16_swap_b32   (dst' = {d.r0, s.r0, d.r2, s.r2}, src' = {d.r1, s.r1, d.r3, s.r3})
//   * inside a 16-lane row:           DPP row_mirror (i <-> 15-i), row_half_mirror (i <-> 7-i),
//                                     quad_perm [2,3,0,1] and [1,0,3,2]
// (semantics verified on MI355X: profiles/r01_dpp_probe.txt).  All functions need EXEC = all lanes.
//
// "Transposed butterfly": to total N per-lane accumulators over the wave, each step halves the number of
// accumulators a lane carries while doubling the lanes summed into each, so N accumulators cost ~N exchanges
// instead of N * log2(64).  Afterwards a lane holds the total of accumulator `slot(lane)`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace rpf {
namespace xl {

constexpr int kQuadXor1 = 0xB1;  // quad_perm [1,0,3,2]
constexpr int kQuadXor2 = 0x4E;  // quad_perm [2,3,0,1]
constexpr int kRowMirror = 0x140;
constexpr int kRowHalfMirror = 0x141;

struct OpSum {
    template <class T> __device__ __forceinline__ static T f(T a, T b) { return a + b; }
};
struct OpMin {
    __device__ __forceinline__ static float f(float a, float b) { return fminf(a, b); }
    __device__ __forceinline__ static double f(double a, double b) { return fmin(a, b); }
};
struct OpMax {
    __device__ __forceinline__ static float f(float a, float b) { return fmaxf(a, b); }
    __device__ __forceinline__ static double f(double a, double b) { return fmax(a, b); }
};

template <class T> struct Words;
template <> struct Words<float> {
    static constexpr int N = 1;
    __device__ __forceinline__ static void split(float v, uint32_t (&w)[1]) { w[0] = __float_as_uint(v); }
    __device__ __forceinline__ static float join(const uint32_t (&w)[1]) { return __uint_as_float(w[0]); }
};
template <> struct Words<uint32_t> {
    static constexpr int N = 1;
    __device__ __forceinline__ static void split(uint32_t v, uint32_t (&w)[1]) { w[0] = v; }
    __device__ __forceinline__ static uint32_t join(const uint32_t (&w)[1]) { return w[0]; }
};
template <> struct Words<double> {
    static constexpr int N = 2;
    __device__ __forceinline__ static void split(double v, uint32_t (&w)[2]) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        w[0] = (uint32_t)u; w[1] = (uint32_t)(u >> 32);
    }
    __device__ __forceinline__ static double join(const uint32_t (&w)[2]) {
        return __longlong_as_double((long long)(((unsigned long long)w[1] << 32) | w[0]));
    }
};
template <> struct Words<uint64_t> {
    static constexpr int N = 2;
    __device__ __forceinline__ static void split(uint64_t v, uint32_t (&w)[2]) { w[0] = (uint32_t)v; w[1] = (uint32_t)(v >> 32); }
    __device__ __forceinline__ static uint64_t join(const uint32_t (&w)[2]) { return ((uint64_t)w[1] << 32) | w[0]; }
};

template <int CTRL, class T>
__device__ __forceinline__ T dpp(T v) {
    uint32_t w[Words<T>::N], r[Words<T>::N];
    Words<T>::split(v, w);
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) r[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[i], CTRL, 0xF, 0xF, false);
    return Words<T>::join(r);
}

// lower 32 lanes get op(A.lo, A.hi), upper 32 lanes get op(B.lo, B.hi)   (lane-wise over the pairs L, L+32)
template <class Op, class T>
__device__ __forceinline__ T exch32(T a, T b) {
    uint32_t wa[Words<T>::N], wb[Words<T>::N], x[Words<T>::N], y[Words<T>::N];
    Words<T>::split(a, wa);
    Words<T>::split(b, wb);
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(wa[i], wb[i], false, false);
        x[i] = r[0]; y[i] = r[1];
    }
    return Op::f(Words<T>::join(x), Words<T>::join(y));
}
// even rows get op over the row pair of A, odd rows get op over the row pair of B
template <class Op, class T>
__device__ __forceinline__ T exch16(T a, T b) {
    uint32_t wa[Words<T>::N], wb[Words<T>::N], x[Words<T>::N], y[Words<T>::N];
    Words<T>::split(a, wa);
    Words<T>::split(b, wb);
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(wa[i], wb[i], false, false);
        x[i] = r[0]; y[i] = r[1];
    }
    return Op::f(Words<T>::join(x), Words<T>::join(y));
}
// inside a row: lanes whose `bit` is 0 keep A and receive A from the partner, lanes with `bit` set keep B
template <class Op, int CTRL, int BIT, class T>
__device__ __forceinline__ T exch_row(T a, T b, int lane) {
    const bool up = (lane & BIT) != 0;
    const T keep = up ? b : a, send = up ? a : b;
    return Op::f(keep, dpp<CTRL>(send));
}
// all-reduce over the low bits of the lane id (bits below 16: inside a row)
template <class Op, class T> __device__ __forceinline__ T allreduce_bit0(T v) { return Op::f(v, dpp<kQuadXor1>(v)); }
template <class Op, class T> __device__ __forceinline__ T allreduce_bits10(T v) {
    v = Op::f(v, dpp<kQuadXor2>(v));
    return Op::f(v, dpp<kQuadXor1>(v));
}
template <class Op, class T> __device__ __forceinline__ T allreduce_row(T v) {
    v = Op::f(v, dpp<kRowMirror>(v));
    v = Op::f(v, dpp<kRowHalfMirror>(v));
    return allreduce_bits10<Op>(v);
}
// whole-wave all-reduce of one value
template <class Op, class T> __device__ __forceinline__ T allreduce(T v) {
    v = exch32<Op>(v, v);
    v = exch16<Op>(v, v);
    return allreduce_row<Op>(v);
}

// ---- transposed butterflies ---------------------------------------------------------------------------
// 4 accumulators -> lane holds total of slot 2*(lane>=32) + (row odd)
__device__ __forceinline__ int slot4(int lane) { return ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1); }
template <class Op, class T>
__device__ __forceinline__ T reduce4(const T (&a)[4]) {
    const T b0 = exch32<Op>(a[0], a[2]), b1 = exch32<Op>(a[1], a[3]);
    return allreduce_row<Op>(exch16<Op>(b0, b1));
}
// 8 accumulators -> lane holds total of slot 4*b5 + 2*b4 + b3
__device__ __forceinline__ int slot8(int lane) { return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1); }
template <class Op, class T>
__device__ __forceinline__ T reduce8(const T (&a)[8], int lane) {
    T b4[4], b2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[i] = exch32<Op>(a[i], a[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) b2[i] = exch16<Op>(b4[i], b4[i + 2]);
    T v = exch_row<Op, kRowMirror, 8>(b2[0], b2[1], lane);
    v = Op::f(v, dpp<kRowHalfMirror>(v));
    return allreduce_bits10<Op>(v);
}
// 16 accumulators -> lane holds total of slot 8*b5 + 4*b4 + 2*b3 + b2 (b_k = bit k of the lane id)
__device__ __forceinline__ int slot16(int lane) {
    return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}
template <class Op, class T>
__device__ __forceinline__ T reduce16(const T (&a)[16], int lane) {
    T b8[8], b4[4], b2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) b8[i] = exch32<Op>(a[i], a[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[i] = exch16<Op>(b8[i], b8[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) b2[i] = exch_row<Op, kRowMirror, 8>(b4[i], b4[i + 2], lane);
    const T v = exch_row<Op, kRowHalfMirror, 4>(b2[0], b2[1], lane);
    return allreduce_bits10<Op>(v);
}
// 32 accumulators -> lane holds total of slot (lane >> 1) & 31
__device__ __forceinline__ int slot32(int lane) { return (lane >> 1) & 31; }
template <class Op, class T>
__device__ __forceinline__ T reduce32(const T (&a)[32], int lane) {
    T b16[16], b8[8], b4[4], b2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) b16[i] = exch32<Op>(a[i], a[i + 16]);
#pragma unroll
    for (int i = 0; i < 8; ++i) b8[i] = exch16<Op>(b16[i], b16[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[i] = exch_row<Op, kRowMirror, 8>(b8[i], b8[i + 4], lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) b2[i] = exch_row<Op, kRowHalfMirror, 4>(b4[i], b4[i + 2], lane);
    const T v = exch_row<Op, kQuadXor2, 2>(b2[0], b2[1], lane);
    return allreduce_bit0<Op>(v);
}

// compile-time loop: f(std::integral_constant<int, I>) for I = I0 .. N-1 (DPP controls are instruction immediates)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// acc += row[lane N of this lane's 16-lane row] * z : a wave-uniform fp64 operand that lives ACROSS the lanes of one register
// pair (replicated in the four rows) instead of in LDS or in 2 VGPRs per value.  gfx90a+ "DP ALU DPP": 64-bit VOP2 operations
// take row_newbcast (and nothing else); v_fmac_f64 is VOP2 on gfx950.  Needs EXEC = all lanes (a disabled source lane
// disables the destination lanes that read it).  NOP: software wait states in front of the first DPP read of a block (a VALU
// write of the source register -- an AGPR reload, a copy -- within two instructions is a hazard the assembler does not see
// inside inline asm).  Semantics pinned on MI355X by scripts/microbench/dpp_bcast_probe.hip.
template <int N, bool NOP = false>
__device__ __forceinline__ void fmac_rowbc(double &acc, double row, double z) {
    static_assert(N >= 0 && N < 16, "lane of a 16-lane row");
    if constexpr (NOP)
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(row), "v"(z), "n"(N));
    else
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(row), "v"(z), "n"(N));
}

// the value itself (v_mov_b64 is VOP1: DPP-capable)
template <int N, bool NOP = false>
__device__ __forceinline__ double rowbc(double row) {
    static_assert(N >= 0 && N < 16, "lane of a 16-lane row");
    double out;
    if constexpr (NOP)
        asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(row), "n"(N));
    else
        asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(row), "n"(N));
    return out;
}

} // namespace xl
} // namespace rpf
