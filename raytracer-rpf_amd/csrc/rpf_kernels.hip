// rpf_kernels.hip -- hand-written gfx950 kernels of the RPF pass.  Compiled with -ffp-contract=off: every
// fp64 operation whose rounding decides a DISCRETE outcome (3-sigma membership, histogram bin) is issued
// in the reference's operation order, so those outcomes are bit-identical to the CPU path; fused
// multiply-adds appear only where written as fma().
//
// Kernel map (reference lines -> kernel):
//   rpf.cpp:302-353 FillMeanAndStddev                       -> pixel_stats_kernel      (thread / pixel)
//   rpf.cpp:556-717 gather, normalise, ComputeCFWeights,
//                   weights, blend   + mi.cpp:5-90          -> filter_pixel_kernel     (one wave64 / pixel)
//   rpf.cpp:779-794 per-pixel reduction (box r=0.5)         -> reduce_kernel
//
// filter_pixel_kernel, one 64-lane workgroup (= one wavefront) per pixel, LDS-resident working set:
//   1b  candidates of the box window are tested 64 at a time in the reference's visiting order (own
//       samples, then x-major / y-minor neighbours); a ballot + prefix popcount appends the accepted
//       samples' offsets to an LDS list, so list order == reference neighbourhood order.
//   2   neighbourhood mean/std need the reference's sequential summation order to be bit-exact: chunks of
//       32 samples x 19 columns are staged through LDS (coalesced gathers), then 38 lanes each run one
//       in-order fp64 chain (19 sums, 19 sums of squares).
//   3   per column: z=(x-M)/SD, wave min/max, bin id (u8) into LDS; 19 marginal + 96 joint histograms by
//       LDS atomics; MI = (T[N] + sum T[J] - sum T[hx] - sum T[hy]) / N with T[k]=k ln k tabulated, which
//       is mi.cpp:79-86 rewritten over integer counts (no log evaluated on the device).
//   4   weights/blend for 8 own samples at a time over the lane-strided neighbourhood, fp64, wave
//       shuffle reductions.
#include <hip/hip_runtime.h>
#include <math.h>

#include "rpf_internal.h"

namespace rpf {

namespace {

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

// the workgroup is exactly one wavefront: the barrier is an LDS/memory ordering point, not a rendezvous
__device__ __forceinline__ void wsync() { __syncthreads(); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
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
__global__ __launch_bounds__(256) void pixel_stats_kernel(PassParams p) {
    const uint64_t HW = (uint64_t)p.H * p.W;
    const uint64_t pix = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
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

// ------------------------------------------------------------------------------------------------
// the fused per-pixel kernel
//   K         compile-time bound on samples per lane: K*64 >= nmax
//   T_IN_LDS  keep the k ln k table in LDS (small neighbourhoods) instead of reading it through L1
// ------------------------------------------------------------------------------------------------
template <int K, bool T_IN_LDS>
__global__ __launch_bounds__(64) void filter_pixel_kernel(PassParams p, LdsLayout L) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *sT = reinterpret_cast<double *>(smem + L.off_T);
    double *sStat = reinterpret_cast<double *>(smem + L.off_stat); // M[19], SD[19]
    double *sHX = reinterpret_cast<double *>(smem + L.off_hx);     // sum_i T[hx_i] per column
    double *sMI = reinterpret_cast<double *>(smem + L.off_mi);     // 96 MI values
    double *sOwn = reinterpret_cast<double *>(smem + L.off_own);   // raw own samples [S][19]
    uint32_t *sOff = reinterpret_cast<uint32_t *>(smem + L.off_off);
    double *sStage = reinterpret_cast<double *>(smem + L.off_union); // [19][kStageChunk+1] (aliases bins)
    uint8_t *sBins = smem + L.off_union;                             // [19][nmax_pad]
    uint32_t *sHist = reinterpret_cast<uint32_t *>(smem + L.off_hist);

    const int lane = threadIdx.x;
    const int W = p.W, H = p.H, S = p.S, b = p.b;

    // XCD-aware pixel assignment: blocks with equal (blockIdx % 8) share an XCD (and its L2); give each
    // XCD one contiguous band of the slab so concurrently resident pixels share window data in L2.
    const int64_t P = (int64_t)(p.row_end - p.row_begin) * W;
    const int64_t band = (P + 7) / 8;
    const int64_t q = (int64_t)(blockIdx.x & 7) * band + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= band || q >= P) return;
    const int y = p.row_begin + (int)(q / W);
    const int x = (int)(q % W);
    const uint64_t HW = (uint64_t)H * W;
    const uint64_t pix = (uint64_t)y * W + x;

    auto Tl = [&](uint32_t k) -> double { return T_IN_LDS ? sT[k] : p.tlogt[k]; };

    if (T_IN_LDS)
        for (int k = lane; k <= p.nmax; k += kWave) sT[k] = p.tlogt[k];

    // ---------------- stage 1b: neighbourhood membership (rpf.cpp:556-586) ----------------------
    const int x0 = max(x - b, 0), x1 = min(x + b, W - 1);
    const int y0 = max(y - b, 0), y1 = min(y + b, H - 1);
    const int nyv = y1 - y0 + 1;
    const int ncells = (x1 - x0 + 1) * nyv;
    const int centre_rank = (x - x0) * nyv + (y - y0);
    const int ncand = (ncells - 1) * S;

    for (int s = lane; s < S; s += kWave) sOff[s] = (uint32_t)(pix * S + s); // own samples first

    double m12[kNFeat], lim12[kNFeat];
#pragma unroll
    for (int k = 0; k < kNFeat; ++k) {
        m12[k] = p.pmean[(uint64_t)k * HW + pix];
        lim12[k] = p.pstd[(uint64_t)k * HW + pix] * 3.0; // multiplyArray(std, 3), rpf.cpp:579
    }
    int n = S;
    for (int q0 = 0; q0 < ncand; q0 += kWave) {
        const int qq = q0 + lane;
        bool pass = false;
        uint32_t off = 0;
        if (qq < ncand) {
            int cell = qq / S;
            const int s = qq - cell * S;
            if (cell >= centre_rank) ++cell;          // rpf.cpp:565: skip the centre pixel
            const int ix = cell / nyv;                 // xn outer ascending (rpf.cpp:562)
            const int iy = cell - ix * nyv;            // yn inner ascending (rpf.cpp:563)
            off = (uint32_t)(((uint64_t)(y0 + iy) * W + (x0 + ix)) * S + s);
            pass = true;
#pragma unroll
            for (int k = 0; k < kNFeat; ++k) {
                const double a = fabs((double)p.planes[(uint64_t)(kColF + k) * p.plane_stride + off] - m12[k]);
                if (a >= lim12[k]) pass = false;       // allLessThan: fails iff a >= b (ops.h:101-104)
            }
        }
        const unsigned long long mask = __ballot(pass);
        if (pass) sOff[n + __popcll(mask & ((1ull << lane) - 1ull))] = off;
        n += __popcll(mask);
    }
    wsync();
    if (lane == 0) p.nbhd[pix] = n;

    if (p.dbg.member_hash != nullptr && lane == 0) {
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
    // lanes 0..18 carry sum(x) of column `lane`, lanes 32..50 carry sum(x*x) of column `lane-32`.
    {
        double acc = 0.0;
        const int myc = lane & 31;
        const bool chain = myc < kNDim;
        const bool is_sq = lane >= 32;
        for (int j0 = 0; j0 < n; j0 += kStageChunk) {
            const int cnt = min(kStageChunk, n - j0);
            for (int e = lane; e < kNDim * kStageChunk; e += kWave) {
                const int c = e / kStageChunk, t = e % kStageChunk;
                if (t < cnt) sStage[c * (kStageChunk + 1) + t] = load_col(p, c, sOff[j0 + t]);
            }
            wsync();
            if (chain) {
                const double *src = sStage + myc * (kStageChunk + 1);
                if (!is_sq) {
                    for (int t = 0; t < cnt; ++t) acc = acc + src[t];                     // ops.h:121
                } else {
                    for (int t = 0; t < cnt; ++t) { const double v = src[t]; acc = acc + v * v; } // ops.h:138
                }
            }
            wsync();
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
        wsync();
    }

    // ---------------- stage 3a: normalise, min/max, bin ids (sd.h:229-232, mi.cpp:14-16) --------
    const int B = max(1, (int)sqrt((double)n)); // mi.cpp:54
    const double dB = (double)B;
    for (int c = 0; c < kNDim; ++c) {
        const double Mc = sStat[c], SDc = sStat[kNDim + c];
        double zr[K];
        double lo = INFINITY, hi = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int j = lane + kWave * kk;
            zr[kk] = 0.0;
            if (j < n) {
                const double xv = load_col(p, c, sOff[j]);
                const double a = xv - Mc;                    // subtractArrays
                const double z = (SDc == 0.0) ? 0.0 : a / SDc; // divideArrays, ops.h:48
                zr[kk] = z;
                lo = fmin(lo, z);
                hi = fmax(hi, z);
                if (j < S) sOwn[j * kNDim + c] = xv;
            }
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        const double range = hi - lo;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int j = lane + kWave * kk;
            if (j < n) {
                int bin = 0;
                if (hi != lo) {                              // mi.cpp:7 / 28 / 34
                    const double t = (zr[kk] - lo) / range * dB; // mi.cpp:14
                    bin = (int)t;
                    bin = min(bin, B - 1);
                    bin = max(bin, 0);
                }
                sBins[c * p.nmax_pad + j] = (uint8_t)bin;
            }
        }
    }
    wsync();
    if (p.dbg.bin_hash != nullptr && lane < kNDim) {
        uint32_t h = 2166136261u;
        for (int j = 0; j < n; ++j) h = fnv1a_u16(h, sBins[lane * p.nmax_pad + j]);
        p.dbg.bin_hash[pix * kNDim + lane] = h;
    }

    // ---------------- stage 3b: histograms -> mutual information (mi.cpp:45-90) -----------------
    for (int t = lane; t < B * B; t += kWave) sHist[t] = 0u; // the staging buffer of stage 2 aliased this region
    wsync();
    const double TN = Tl((uint32_t)n);
    const double dn = (double)n;
    for (int c = 0; c < kNDim; ++c) { // marginal term sum_i T[hx_i] of every column
        for (int j = lane; j < n; j += kWave) atomicAdd(&sHist[sBins[c * p.nmax_pad + j]], 1u);
        wsync();
        double a = 0.0;
        for (int t = lane; t < B; t += kWave) {
            const uint32_t h = sHist[t];
            sHist[t] = 0u;
            a += Tl(h);
        }
        a = wave_sum(a);
        if (lane == 0) sHX[c] = a;
        wsync();
    }
    for (int pr = 0; pr < kNPair; ++pr) {
        const int ca = c_pairs.a[pr], cb = c_pairs.b[pr];
        const uint8_t *ba = sBins + ca * p.nmax_pad;
        const uint8_t *bb = sBins + cb * p.nmax_pad;
        for (int j = lane; j < n; j += kWave) atomicAdd(&sHist[(uint32_t)ba[j] * B + bb[j]], 1u); // mi.cpp:39
        wsync();
        double a = 0.0;
        for (int t = lane; t < B * B; t += kWave) {
            const uint32_t h = sHist[t];
            sHist[t] = 0u;
            a += Tl(h);
        }
        a = wave_sum(a);
        // sum_ij pXY ln(pXY/(pX pY)) = (N ln N + sum J ln J - sum hx ln hx - sum hy ln hy) / N
        // A column whose samples all fall in one bin (constant feature, or B == 1) has pX == 1, so every
        // term of mi.cpp:84 is pXY*log(1): the reference returns exactly 0, and that exact zero decides
        // whether rpf.cpp:465/470 divide 0 by 0.  sum_i T[hx_i] == T[N] iff the column has a single bin.
        const double hxa = sHX[ca], hxb = sHX[cb];
        const double mi = (hxa == TN || hxb == TN) ? 0.0 : (TN + a - hxa - hxb) / dn;
        if (lane == 0) {
            sMI[pr] = mi;
            if (p.dbg.mi) p.dbg.mi[pix * kNPair + pr] = mi;
        }
        wsync();
    }

    // ---------------- stage 3c: alpha, beta, W_r_c (rpf.cpp:444-487), every lane redundantly ----
    double alpha[3], beta[kNFeat], wrc;
    {
        double Drf[kNFeat], Dpf[kNFeat], Dcf[kNFeat], Drc[3], Dpc[3], Dfc[3];
#pragma unroll
        for (int i = 0; i < kNFeat; ++i) {
            Drf[i] = 0.0 + sMI[i * 4 + 0] + sMI[i * 4 + 1]; // rpf.cpp:421
            Dpf[i] = 0.0 + sMI[i * 4 + 2] + sMI[i * 4 + 3]; // rpf.cpp:425
            Dcf[i] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int base = 48 + c * 16;
            Drc[c] = 0.0 + sMI[base + 0] + sMI[base + 1]; // rpf.cpp:432
            Dpc[c] = 0.0 + sMI[base + 2] + sMI[base + 3]; // rpf.cpp:436
            double f = 0.0;
#pragma unroll
            for (int j = 0; j < kNFeat; ++j) {
                f += sMI[base + 4 + j];                   // rpf.cpp:440
                Dcf[j] += sMI[base + 4 + j];
            }
            Dfc[c] = f;
        }
        double D_f_c = 0.0, D_r_c = 0.0, D_p_c = 0.0;     // rpf.cpp:449-456
#pragma unroll
        for (int i = 0; i < 3; ++i) { D_f_c += Dfc[i]; D_r_c += Drc[i]; D_p_c += Dpc[i]; }
        const double e = (p.policy == RPF_DEGEN_EPS) ? p.eps : 0.0;
        const double den = D_f_c + D_r_c + D_p_c + e;
        double wsum = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double w = Drc[i] / (Drc[i] + Dpc[i] + e); // rpf.cpp:470
            alpha[i] = 1 - w;                                // rpf.cpp:475
            wsum += w;
        }
        wrc = wsum / 3;                                      // rpf.cpp:487
#pragma unroll
        for (int k = 0; k < kNFeat; ++k) {
            double num; // what rpf.cpp:464 reads as D_f_ck[k] (3-element array indexed to 11: SURVEY F3)
            if (p.beta_map == RPF_BETA_PAPER) num = Dcf[k];
            else if (p.beta_map == RPF_BETA_REF_GCC11_O2) num = k < 3 ? Dfc[k < 3 ? k : 0] : (k < 8 ? 0.0 : Drf[k >= 8 ? k - 8 : 0]);
            else num = k < 3 ? Dfc[k < 3 ? k : 0] : (k < 4 ? 0.0 : Drf[k >= 4 ? k - 4 : 0]);
            const double Wc = num / den;                       // rpf.cpp:464
            const double Wr = Drf[k] / (Drf[k] + Dpf[k] + e);  // rpf.cpp:465
            beta[k] = (1 - Wr) * Wc;                           // rpf.cpp:479
        }
        if (lane == 0) {
            if (p.dbg.wrc) p.dbg.wrc[pix] = wrc;
            if (p.dbg.alpha)
                for (int i = 0; i < 3; ++i) p.dbg.alpha[pix * 3 + i] = alpha[i];
            if (p.dbg.beta)
                for (int i = 0; i < kNFeat; ++i) p.dbg.beta[pix * kNFeat + i] = beta[i];
        }
    }

    // ---------------- stage 4: weights and blend (rpf.cpp:627-717) ------------------------------
    // exponent of w_ij folded over raw values:  sum_k coef_k (x_ik - x_jk)^2  with
    // coef_k = weight_k / (SD_k^2 * 2 sigma^2); a column with SD_k == 0 normalises to z == 0 for every
    // sample (ops.h:48), so its term is weight_k * 0.
    double coef[17];
    {
        const double sigma_c2 = p.seed * p.seed / (1 - wrc) / (1 - wrc); // rpf.cpp:662
        const double inv2sc = 1.0 / (2 * sigma_c2);
        const double inv2sp = 1.0 / (2 * (p.sigma_p * p.sigma_p));       // rpf.cpp:664,668
        double wk[17];
        wk[0] = 1.0; wk[1] = 1.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) wk[2 + k] = alpha[k];
#pragma unroll
        for (int k = 0; k < kNFeat; ++k) wk[5 + k] = beta[k];
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const int col = k < 5 ? k : k + 2;
            const double sd = sStat[kNDim + col];
            const double s2 = k < 2 ? inv2sp : inv2sc;
            coef[k] = (sd == 0.0) ? (wk[k] * 0.0) * s2 : wk[k] / (sd * sd) * s2;
        }
    }
    bool bad = false;
    for (int i0 = 0; i0 < S; i0 += 8) {
        double sw[8], s0[8], s1[8], s2[8];
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) { sw[ii] = 0.0; s0[ii] = 0.0; s1[ii] = 0.0; s2[ii] = 0.0; }
#pragma unroll 1
        for (int kk = 0; kk < K; ++kk) {
            const int j = lane + kWave * kk;
            if (j >= n) break;
            const uint32_t off = sOff[j];
            double xj[17];
#pragma unroll
            for (int k = 0; k < 17; ++k) xj[k] = load_col(p, k < 5 ? k : k + 2, off);
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const int i = i0 + ii;
                if (i < S) {
                    const double *oi = sOwn + i * kNDim;
                    double E = 0.0;
#pragma unroll
                    for (int k = 0; k < 17; ++k) {
                        const double d = oi[k < 5 ? k : k + 2] - xj[k];
                        E = fma(d * d, coef[k], E);
                    }
                    const double w = exp(-E);               // rpf.cpp:667-670
                    sw[ii] += w;                            // rpf.cpp:691
                    s0[ii] = fma(w, xj[2], s0[ii]);         // rpf.cpp:692 (raw neighbourhood colours)
                    s1[ii] = fma(w, xj[3], s1[ii]);
                    s2[ii] = fma(w, xj[4], s2[ii]);
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            if (i < S) { // wave-uniform
                const double tw = wave_sum(sw[ii]);
                double c3[3] = {wave_sum(s0[ii]), wave_sum(s1[ii]), wave_sum(s2[ii])};
                if (lane < 3) {
                    double prime = (lane == 0 ? c3[0] : (lane == 1 ? c3[1] : c3[2])) / tw; // rpf.cpp:700
                    if (isnan(prime)) {                     // rpf.cpp:702: the reference exits here
                        bad = true;
                        if (p.policy == RPF_DEGEN_EPS) prime = sOwn[i * kNDim + kColC + lane];
                    }
                    p.col_out[(uint64_t)lane * p.plane_stride + pix * S + i] = prime;
                }
            }
        }
    }
    if (__any(bad) && lane == 0) {
        atomicAdd(&p.status[0], 1);
        atomicMin(&p.status[1], (int)pix);
    }
}

__global__ __launch_bounds__(256) void colour_from_planes_kernel(const float *planes, double *colour, uint64_t ps) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * ps) return;
    colour[i] = (double)planes[2 * ps + i];
}

__global__ __launch_bounds__(256) void copy_f64_kernel(const double *src, double *dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// rpf.cpp:783-794 with the default box reconstruction filter: pixel = sum_s(L*rayWeight) / S
__global__ __launch_bounds__(256) void reduce_kernel(const double *colour, const float *ray_weight, float *sample_rgb,
                                                      float *pixel_rgb, int W, int H, int S) {
    const uint64_t HW = (uint64_t)H * W;
    const uint64_t pix = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
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

template <int K>
hipError_t launch_filter_k(const PassParams &p, const LdsLayout &L, bool t_in_lds, unsigned grid, hipStream_t s) {
    hipError_t e;
    if (t_in_lds) {
        e = hipFuncSetAttribute((const void *)filter_pixel_kernel<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)L.total);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((filter_pixel_kernel<K, true>), dim3(grid), dim3(64), L.total, s, p, L);
    } else {
        e = hipFuncSetAttribute((const void *)filter_pixel_kernel<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)L.total);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((filter_pixel_kernel<K, false>), dim3(grid), dim3(64), L.total, s, p, L);
    }
    return hipGetLastError();
}

} // namespace

static uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

LdsLayout lds_layout(int S, int nmax, int nmax_pad, int bmax, bool t_in_lds) {
    LdsLayout L{};
    uint32_t o = 0;
    L.off_T = o;
    if (t_in_lds) o += align_up((uint32_t)(nmax + 1) * 8u, 16);
    L.off_stat = o; o += align_up(2 * kNDim * 8, 16);
    L.off_hx = o; o += align_up(kNDim * 8, 16);
    L.off_mi = o; o += align_up(kNPair * 8, 16);
    L.off_own = o; o += align_up((uint32_t)S * kNDim * 8u, 16);
    L.off_off = o; o += align_up((uint32_t)nmax * 4u, 16);
    L.off_union = o;
    const uint32_t stage = align_up(kNDim * (kStageChunk + 1) * 8, 16);
    const uint32_t bins = align_up((uint32_t)kNDim * (uint32_t)nmax_pad, 16);
    L.off_hist = o + bins;
    const uint32_t hist = align_up((uint32_t)bmax * (uint32_t)bmax * 4u, 16);
    const uint32_t uni = (bins + hist) > stage ? (bins + hist) : stage;
    o += uni;
    L.total = o;
    return L;
}

int max_lds_per_block() { return 160 * 1024; }

hipError_t launch_pixel_stats(const PassParams &p, hipStream_t s) {
    const uint64_t HW = (uint64_t)p.H * p.W;
    hipLaunchKernelGGL(pixel_stats_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_filter_pass(const PassParams &p, hipStream_t s, uint32_t *lds_bytes_out) {
    const bool t_in_lds = (uint32_t)(p.nmax + 1) * 8u <= 8192u;
    const LdsLayout L = lds_layout(p.S, p.nmax, p.nmax_pad, p.bmax, t_in_lds);
    if (lds_bytes_out) *lds_bytes_out = L.total;
    if ((int)L.total > max_lds_per_block()) return hipErrorInvalidValue;
    const int64_t P = (int64_t)(p.row_end - p.row_begin) * p.W;
    if (P <= 0) return hipSuccess;
    const int64_t band = (P + 7) / 8;
    const unsigned grid = (unsigned)(band * 8);
    const int per_lane = (p.nmax + kWave - 1) / kWave;
    if (per_lane <= 1) return launch_filter_k<1>(p, L, t_in_lds, grid, s);
    if (per_lane <= 2) return launch_filter_k<2>(p, L, t_in_lds, grid, s);
    if (per_lane <= 4) return launch_filter_k<4>(p, L, t_in_lds, grid, s);
    if (per_lane <= 7) return launch_filter_k<7>(p, L, t_in_lds, grid, s);
    if (per_lane <= 13) return launch_filter_k<13>(p, L, t_in_lds, grid, s);
    if (per_lane <= 25) return launch_filter_k<25>(p, L, t_in_lds, grid, s);
    if (per_lane <= 49) return launch_filter_k<49>(p, L, t_in_lds, grid, s);
    return hipErrorInvalidValue;
}

hipError_t launch_colour_from_planes(const float *planes, double *colour, uint64_t ps, hipStream_t s) {
    hipLaunchKernelGGL(colour_from_planes_kernel, dim3((unsigned)((3 * ps + 255) / 256)), dim3(256), 0, s, planes,
                       colour, ps);
    return hipGetLastError();
}

hipError_t launch_copy_f64(const double *src, double *dst, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(copy_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_reduce(const double *colour, const float *ray_weight, float *sample_rgb, float *pixel_rgb, int W,
                         int H, int S, hipStream_t s) {
    const uint64_t HW = (uint64_t)H * W;
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, s, colour, ray_weight,
                       sample_rgb, pixel_rgb, W, H, S);
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
