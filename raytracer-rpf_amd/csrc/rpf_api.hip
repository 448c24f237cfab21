// rpf_api.hip -- the C ABI of include/rpf_hip.h: context, HBM workspace, pass sequencing, status and
// counters.  The pass loop mirrors RPFIntegrator::Render (rpf.cpp:767-775): for each box size run
// FillMeanAndStddev (stage 1a) then the fused filter; filtered colours replace the film's colours
// (rpf.cpp:732) and feed the next pass.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rpf_internal.h"

using namespace rpf;

struct rpf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // grow-only HBM workspace
    char *d_planes = nullptr;    size_t cap_planes = 0;   // ndim planes of fp32 (or fp16)
    float *d_rayw = nullptr;     size_t cap_rayw = 0;
    double *d_colA = nullptr;    size_t cap_colA = 0;     // 3 planes fp64
    double *d_colB = nullptr;    size_t cap_colB = 0;
    double *d_pmean = nullptr;   size_t cap_pmean = 0;    // [12][H*W]
    double *d_pstd = nullptr;    size_t cap_pstd = 0;
    int32_t *d_nbhd = nullptr;   size_t cap_nbhd = 0;
    uint64_t *d_tfix = nullptr;  size_t cap_tfix = 0;     int tfix_n = 0;     // round(k ln k * 2^44), k = 0..n
    uint64_t *d_dfix = nullptr;  size_t cap_dfix = 0;                         // first differences
    float *d_srgb = nullptr;     size_t cap_srgb = 0;
    float *d_prgb = nullptr;     size_t cap_prgb = 0;
    double *d_carry = nullptr; size_t cap_carry = 0;       // split route of the 32- / 64-spp classes: statistics / weights between its three kernels
    int32_t *d_status = nullptr;                           // [0] bad count [1] first bad
    unsigned long long *d_nred = nullptr;                  // [0] sum N [1] max N
    uint32_t *d_lists = nullptr; size_t cap_lists = 0;     // size binning: [7][H*W] pixel lists
    int last_route = -1;                                   // last pass: 1 = count first, 0 = fused, 2 = size-binned (rpf_query_route)
    uint32_t *d_class_counts = nullptr;                    // [kNumClasses] list sizes + [2] the route probe's counts
    uint64_t *d_masks = nullptr; size_t cap_masks = 0;     // size binning: stage-1b acceptance masks [H*W][stride]
    char *d_big_list = nullptr;  size_t cap_big_list = 0;  // streaming kernel: member lists [slots][nmax] u32
    char *d_big_bins = nullptr;  size_t cap_big_bins = 0;  //                   bin ids [slots][ndim][nmax] u8
    uint8_t *d_flat = nullptr;   size_t cap_flat = 0;      // stage 1a by-product: pixels with a zero-variance feature [H*W]
    int32_t *d_nan_flag = nullptr;                         // ... and whether any feature mean of the buffer is NaN
    uint32_t *d_redo_list = nullptr; size_t cap_redo = 0;  // REF_ABORT: pixels handed to the reference-expression kernel [H*W]
    uint32_t *d_redo_count = nullptr;
    // membership depends on the features only, so within one call a pass with the same box and rows re-uses the
    // previous pass's masks and lists (reset at every API entry: the planes may change between calls)
    bool flat_fresh = false;   // d_flat / d_nan_flag describe the planes of the call in progress (stage 1a ran in it)
    bool bin_valid = false;
    int bin_box = 0, bin_r0 = 0, bin_r1 = 0;
    uint32_t bin_counts[kNumClasses] = {};
    // debug planes
    void *d_dbg[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap_dbg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // host-buffer entry (rpf_filter): row-band pipeline, uploads / downloads on their own streams
    hipStream_t s_up = nullptr, s_down = nullptr;
    std::vector<hipEvent_t> band_ev; // no-timing events, two per band
    rpf_counters counters{};
    Tuning tun;                      // rpf_set_option
};

namespace {

// Per-stage tracing hooks (the reference brackets its phases with ProfilePhase, core/stats.h:254): roctx ranges around the
// host-side enqueue of upload / stage 1a / count + classify / each size-class launch / redo / reduce / download, visible in
// `rocprofv3 --marker-trace --kernel-trace`.  The marker library is looked up at run time (the profiler preloads it; without
// it, or without the library on the machine, the hooks are two null checks): librpf_hip.so has no link-time dependency on it.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
const Roctx &roctx() { static const Roctx r; return r; }
struct Range {
    bool on;
    explicit Range(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

int32_t fail(rpf_ctx *c, int32_t st, const std::string &msg) {
    if (c) c->err = msg;
    return st;
}
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(ctx, RPF_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

template <class T>
int32_t ensure(rpf_ctx *ctx, T *&ptr, size_t &cap, size_t bytes) {
    if (bytes <= cap && ptr) return RPF_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&ptr, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? RPF_E_NOMEM : RPF_E_HIP,
                                     std::string("hipMalloc: ") + hipGetErrorString(e));
    cap = bytes;
    return RPF_OK;
}

// the sample-vector layout a descriptor names (0 in all three fields = the reference's 19 dims, fp32 planes)
SampleLayout layout_of(const rpf_desc *d) {
    SampleLayout l;
    if (d->n_random != 0 || d->n_feat != 0 || d->plane_dtype != 0) {
        l.nR = d->n_random ? d->n_random : 2;
        l.nF = d->n_feat ? d->n_feat : 12;
        l.f16 = d->plane_dtype == RPF_PLANES_F16 ? 1 : (d->plane_dtype == RPF_PLANES_F32 ? 0 : -1);
    }
    return l;
}

int32_t validate(rpf_ctx *ctx, const rpf_desc *d, bool need_boxes) {
    if (!ctx) return RPF_E_BADARG;
    if (!d) return fail(ctx, RPF_E_BADARG, "desc is NULL");
    if (d->W <= 0 || d->H <= 0 || d->S <= 0) return fail(ctx, RPF_E_BADARG, "W, H, S must be positive");
    if (d->row_begin < 0 || d->row_end > d->H || d->row_begin > d->row_end)
        return fail(ctx, RPF_E_BADARG, "row range must satisfy 0 <= row_begin <= row_end <= H");
    if ((uint64_t)d->W * d->H * d->S >= (1ull << 32))
        return fail(ctx, RPF_E_BADARG, "W*H*S must be < 2^32 per slab (split the image into row slabs)");
    if (d->beta_map < 0 || d->beta_map > RPF_BETA_PAPER) return fail(ctx, RPF_E_BADARG, "unknown beta_map");
    if (!layout_of(d).supported())
        return fail(ctx, RPF_E_UNSUPPORTED, "sample layout: kernels exist for n_random=2, n_feat=12, fp32 planes (the reference's "
                                            "19 dims) and n_random=4, n_feat=18, fp16 planes (27 dims)");
    if (d->degenerate_policy < 0 || d->degenerate_policy > RPF_DEGEN_EPS)
        return fail(ctx, RPF_E_BADARG, "unknown degenerate_policy");
    if (need_boxes) {
        if (d->n_box < 1 || d->n_box > RPF_MAX_BOXES) return fail(ctx, RPF_E_BADARG, "n_box must be 1..8");
        // The reference filters the WHOLE film in every pass (rpf.cpp:732 swaps the full film), so pass i+1 reads
        // filtered colours in every window row.  A strict sub-slab only filters its own rows: its halo rows would stay
        // unfiltered and the owned rows next to them would silently differ from the full-frame result.  Sub-slabs are
        // therefore driven one pass per call, with a colour-halo exchange in between (rpf_filter_multi does exactly
        // that inside one process; slabs.py across processes).
        if (d->n_box > 1 && (d->row_begin != 0 || d->row_end != d->H))
            return fail(ctx, RPF_E_BADARG, "n_box > 1 needs row_begin == 0 and row_end == H: a sub-slab must be filtered "
                                           "one pass per call with a colour-halo exchange in between (or rpf_filter_multi)");
        for (int i = 0; i < d->n_box; ++i)
            if (d->box_sizes[i] < 1 || (d->box_sizes[i] & 1) == 0)
                return fail(ctx, RPF_E_BADARG, "box sizes must be odd and positive (rpf.cpp:561)");
    }
    return RPF_OK;
}

// T[k] = k ln k in 2^-44 fixed point (computed in long double, rounded once) and its first differences
int32_t ensure_tables(rpf_ctx *ctx, int nmax) {
    if (ctx->d_tfix && ctx->tfix_n >= nmax + 1) return RPF_OK;
    std::vector<uint64_t> t((size_t)nmax + 1), d((size_t)nmax + 1, 0);
    t[0] = 0;
    for (int k = 1; k <= nmax; ++k) t[k] = (uint64_t)std::llroundl(std::ldexp((long double)k * std::log((long double)k), 44));
    for (int k = 0; k < nmax; ++k) d[k] = t[k + 1] - t[k];
    int32_t st;
    if ((st = ensure(ctx, ctx->d_tfix, ctx->cap_tfix, t.size() * sizeof(uint64_t)))) return st;
    if ((st = ensure(ctx, ctx->d_dfix, ctx->cap_dfix, d.size() * sizeof(uint64_t)))) return st;
    HIP_TRY(hipMemcpy(ctx->d_tfix, t.data(), t.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_dfix, d.data(), d.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    ctx->tfix_n = nmax + 1;
    return RPF_OK;
}

struct PassSetup {
    PassParams p;
    uint32_t lds = 0;
};

int32_t setup_pass(rpf_ctx *ctx, const rpf_desc *d, int box, const void *d_planes, const double *col_in,
                   double *col_out, const rpf_debug *dbg_dev, PassSetup &out) {
    if (box < 1 || (box & 1) == 0) return fail(ctx, RPF_E_BADARG, "box must be odd and positive");
    PassParams &p = out.p;
    std::memset(&p, 0, sizeof(p));
    p.W = d->W; p.H = d->H; p.S = d->S;
    p.lay = layout_of(d);
    const int kNFeat = p.lay.nF;
    p.row_begin = d->row_begin; p.row_end = d->row_end;
    p.box = box; p.b = (box - 1) / 2;
    p.beta_map = d->beta_map; p.policy = d->degenerate_policy;
    p.fast_weights = (d->flags & RPF_FLAG_FAST_WEIGHTS) ? 1 : 0;
    p.stage_mask = ctx->tun.stage_mask; // timing ablation knob (rpf_set_option); results are wrong unless -1
    p.screen = ctx->tun.screen;
    const int64_t nmax64 = (int64_t)box * box * d->S;
    if (nmax64 > kMaxNbhd) return fail(ctx, RPF_E_UNSUPPORTED, "box*box*S > 65535: neighbourhood too large (16-bit histogram cells, one-byte bin ids)");
    p.nmax = (int)nmax64;
    {   // XCD strip width: box rows x (strip + halo) px x S samples x ~88 B against a budget of L2 bytes.  Measured
        // (scripts/strip_sweep.sh, profiles/r02_strip_width.txt): the kernel time does not depend on it, the fetched bytes
        // do -- at 8 spp wide strips win (a quarter of the 4 MiB L2: 128 px), from 16 spp up narrow ones (the size-binned
        // route reads every window twice, count pass and filter pass, several hundred microseconds apart): 256 KiB.
        const int64_t per_px = (int64_t)box * d->S * 88;
        const int64_t budget = d->S > 8 ? (256 << 10) : (1 << 20);
        int w = (int)(budget / std::max<int64_t>(per_px, 1)) - 2 * ((box - 1) / 2);
        w = std::max(8, std::min(128, w));
        p.strip_w = w & ~7;
        if (ctx->tun.strip_w > 0) p.strip_w = ctx->tun.strip_w;
    }
    p.bmax = (int)std::sqrt((double)p.nmax);
    if (p.bmax < 1) p.bmax = 1;
    p.eps = d->eps; p.seed = d->sigma_seed;
    p.sigma_p = (double)(box / 4); // rpf.cpp:531: integer division
    p.plane_stride = (uint64_t)d->W * d->H * d->S;
    p.planes = d_planes; p.col_in = col_in; p.col_out = col_out;
    const size_t HW = (size_t)d->W * d->H;
    int32_t st;
    if ((st = ensure(ctx, ctx->d_pmean, ctx->cap_pmean, HW * kNFeat * sizeof(double)))) return st;
    if ((st = ensure(ctx, ctx->d_pstd, ctx->cap_pstd, HW * kNFeat * sizeof(double)))) return st;
    if ((st = ensure(ctx, ctx->d_nbhd, ctx->cap_nbhd, HW * sizeof(int32_t)))) return st;
    if ((st = ensure_tables(ctx, p.nmax))) return st;
    p.pmean = ctx->d_pmean; p.pstd = ctx->d_pstd; p.tfix = ctx->d_tfix; p.dfix = ctx->d_dfix;
    p.nbhd = ctx->d_nbhd; p.status = ctx->d_status;
    if ((st = ensure(ctx, ctx->d_flat, ctx->cap_flat, HW))) return st;
    p.flat = ctx->d_flat; p.nan_flag = ctx->d_nan_flag;
    if (dbg_dev) p.dbg = *dbg_dev;
    {   // LDS of the largest resident kernel this pass can launch (larger neighbourhoods stream: filter_pixel_big_kernel)
        const int nres = std::min(p.nmax, kMaxResident), bres = std::max(1, (int)std::sqrt((double)nres));
        out.lds = lds_layout(p.S, nres, bres, table_in_lds(p.S, nres, bres, ctx->tun, p.lay), ctx->tun, p.lay).total;
    }
    if ((int)out.lds > max_lds_per_block())
        return fail(ctx, RPF_E_UNSUPPORTED, "neighbourhood working set exceeds 160 KiB of LDS");
    return RPF_OK;
}

int32_t finish_counters(rpf_ctx *ctx, const rpf_desc *d, hipStream_t s);

// One fused-filter pass over rows [p.row_begin, p.row_end).  When box*box*S is above what the one-wave kernels hold
// (512 samples), the neighbourhood sizes are counted first and every kernel family filters its own pixel list with
// LDS sized for its capacity (rpf_kernels.hip, "neighbourhood-size binning"); option "binning" = 0/1 overrides.
// Needs stage 1a's planes (pmean / pstd) for those rows.  Synchronises the stream when it bins (list sizes).
// REF_ABORT: the pixels the resident kernels appended to the redo list (an MI table inside the fixed-point rounding band at
// a non-power-of-two N: the reference returns rounding residue there, rpf_filter_impl.inc stage 3b) are filtered again,
// whole, by the streaming kernel, which evaluates the reference's floating-point expression for such tables.  The list
// size stays on the device (no read-back): a fixed small grid whose workgroups find the list empty on ordinary frames.
int32_t launch_redo(rpf_ctx *ctx, const PassParams &p, hipStream_t s, int *launches) {
    if (p.redo_list == nullptr) return RPF_OK;
    Range rg("rpf:redo (reference-expression kernel)");
    PassParams q = p;
    q.pix_list = p.redo_list;
    q.list_count = 0;
    const size_t per_slot = (size_t)q.nmax * (4 + (size_t)p.lay.ndim());
    const uint32_t slots = (uint32_t)std::max<size_t>(8, std::min<size_t>(128, (64u << 20) / per_slot));
    int32_t st;
    if ((st = ensure(ctx, ctx->d_big_list, ctx->cap_big_list, (size_t)slots * q.nmax * 4))) return st;
    if ((st = ensure(ctx, ctx->d_big_bins, ctx->cap_big_bins, (size_t)slots * q.nmax * p.lay.ndim()))) return st;
    HIP_TRY(launch_filter_big(q, ctx->d_big_list, ctx->d_big_bins, slots, p.redo_count, s));
    if (launches) ++*launches;
    return RPF_OK;
}

int32_t launch_filter_binned(rpf_ctx *ctx, const PassParams &p_in, hipStream_t s, int *launches) {
    PassParams p = p_in;
    p.redo_list = nullptr; p.redo_count = nullptr;
    if (p.policy == RPF_DEGEN_REF_ABORT && !p.fast_weights && ctx->tun.stage_mask == -1) {
        int32_t e;
        if ((e = ensure(ctx, ctx->d_redo_list, ctx->cap_redo, (size_t)p.W * p.H * sizeof(uint32_t)))) return e;
        HIP_TRY(hipMemsetAsync(ctx->d_redo_count, 0, sizeof(uint32_t), s));
        p.redo_list = ctx->d_redo_list; p.redo_count = ctx->d_redo_count;
    }
    bool bin = p.nmax > 512;
    if (ctx->tun.binning >= 0) bin = ctx->tun.binning != 0;
    if (p.nmax > kMaxResident) bin = true; // the streaming kernel takes the pixels no resident kernel can hold
    if (p.dbg.nbhd_size || p.dbg.mi) { /* debug planes are written by whichever launch owns the pixel: fine */ }
    // small neighbourhoods (N <= 64) run on the packed kernels, several pixels per wave (rpf_packed_impl.inc); option "packed"
    const bool packed = ctx->tun.packed != 0 && !p.fast_weights && ctx->tun.stage_mask == -1 && ctx->tun.lds_pad == 0;
    const size_t HW = (size_t)p.W * p.H;
    int32_t st;
    if (!bin) {
        PassParams q = p;
        if (packed) {
            // the fused kernel finds N itself (stage 1b); a pixel with N <= 64 leaves its acceptance masks, joins the list of
            // its lane class and exits; the four packed launches read their list sizes on the device (no host read-back)
            q.mask_stride = (uint32_t)std::max<int64_t>(1, ((int64_t)(p.box * p.box - 1) * p.S + 63) / 64);
            if ((st = ensure(ctx, ctx->d_lists, ctx->cap_lists, (size_t)kNumClasses * HW * sizeof(uint32_t)))) return st;
            if ((st = ensure(ctx, ctx->d_masks, ctx->cap_masks, HW * q.mask_stride * sizeof(uint64_t)))) return st;
            HIP_TRY(hipMemsetAsync(ctx->d_class_counts, 0, kNumClasses * sizeof(uint32_t), s));
            q.reroute_masks = ctx->d_masks;
        }
        // Two routes, same results (tests: option "count_first" 0 / 1).  FUSED: filter_pixel_kernel runs stage 1b itself (its
        // twelve gathers per candidate hide behind the other stages of the pixels in flight) and re-routes the pixels it finds
        // small; the route of buffers whose neighbourhoods are large (the headline generator: N = 296 of 392).  COUNT FIRST:
        // stage 1b for the whole slab as its own launch (nbhd_count_kernel, two phases: a path-traced buffer rejects most
        // candidates on the first features), then the pixels are dealt by N -- the four packed lists, and the rest (N > 64) into
        // the list the fused kernel walks, rebuilding its member list from the masks; the route of buffers whose neighbourhoods
        // are small, where a one-wave workgroup per pixel that only finds out it has nothing to do is the whole cost (6.2 vs
        // 2.7 ms per 1080p frame).  Which one: a probe -- the test on a lattice of every 32nd pixel of every 32nd row (~2000
        // pixels of a 1080p frame, ~20 us + a 4-byte read-back): count first when half of them have N <= 64.
        uint32_t *glist = ctx->d_lists + (size_t)(kNumClasses - 1) * HW, *gcount = ctx->d_class_counts + (kNumClasses - 1);
        uint32_t *pcount = ctx->d_class_counts + kNumClasses;
        const uint32_t npix = (uint32_t)((size_t)(p.row_end - p.row_begin) * p.W);
        bool run_main = true;
        int count_first = packed ? ctx->tun.count_first : 0;
        uint32_t n_general = npix;
        const bool prelisted = packed && ctx->flat_fresh;
        if (packed && (count_first < 0 || prelisted)) {
            // pixels that stage 1a proved flat (a zero-variance feature, no NaN mean in the buffer: N = S) never reach the fused
            // kernel or the count pass: the others are listed in slab order (the list of the streaming class is free on this
            // route): a one-wave workgroup per flat pixel that only finds out it has nothing to do cost 5.8 ms per 1080p frame
            // of a captured-like buffer.
            Range rg("rpf:route probe + prelist (flat quads)");
            uint32_t probe[2] = {0, 0};
            const int step = 32;
            if (count_first < 0) {
                PassParams pr = q;
                pr.masks = nullptr; pr.reroute_masks = nullptr;
                if (!ctx->flat_fresh) pr.flat = nullptr;
                HIP_TRY(hipMemsetAsync(pcount, 0, 2 * sizeof(uint32_t), s));
                HIP_TRY(launch_nbhd_count(pr, step, pcount, nullptr, nullptr, 0, s));
                HIP_TRY(hipMemcpyAsync(probe, pcount, sizeof(probe), hipMemcpyDeviceToHost, s));
            }
            if (prelisted) {
                HIP_TRY(launch_prelist(q, glist, gcount, s));
                HIP_TRY(hipMemcpyAsync(&n_general, gcount, sizeof(n_general), hipMemcpyDeviceToHost, s));
            }
            HIP_TRY(hipStreamSynchronize(s)); // one read-back for both
            if (count_first < 0) {
                const uint32_t n_probe = (uint32_t)((p.W + step - 1) / step) * (uint32_t)((p.row_end - p.row_begin + step - 1) / step);
                const uint32_t n_probe_general = n_probe - std::min(n_probe, probe[1]);
                count_first = (n_probe_general != 0 && 2u * probe[0] >= n_probe_general) ? 1 : 0; // (only flat pixels: nothing to count)
            }
        }
        ctx->last_route = count_first;
        if (count_first == 1 && n_general != 0) {
            Range rg("rpf:stage 1b + classify");
            uint32_t *rlist = ctx->d_lists + (size_t)(kNumClasses - 2) * HW, *rcount = ctx->d_class_counts + (kNumClasses - 2);
            uint32_t n_rest = 0;
            q.reroute_masks = nullptr;
            q.masks = ctx->d_masks;
            if (!ctx->flat_fresh) q.flat = nullptr;
            if (prelisted && n_general < npix) { HIP_TRY(launch_nbhd_count(q, 1, nullptr, glist, gcount, n_general, s)); }
            else { HIP_TRY(launch_nbhd_count(q, 1, nullptr, nullptr, nullptr, 0, s)); }
            HIP_TRY(launch_classify(q, ctx->d_lists, ctx->d_class_counts, kNumPacked, kNumClasses - 2, s));
            HIP_TRY(hipMemcpyAsync(&n_rest, rcount, sizeof(n_rest), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s)); // the fused kernel's grid
            if (n_rest != 0) { q.pix_list = rlist; q.list_count = n_rest; }
            else run_main = false;
        } else if (prelisted) {
            if (n_general == 0) run_main = false;
            else if (n_general < npix) { q.pix_list = glist; q.list_count = n_general; }
        }
        if (run_main) {
            Range rg("rpf:filter_pixel_kernel");
            HIP_TRY(launch_filter_pass(q, ctx->tun, s, nullptr));
            if (launches) ++*launches;
        }
        q.pix_list = nullptr; q.list_count = 0;
        if (packed) {
            Range rg("rpf:packed kernels (N <= 8, 16, 32, 64)");
            if (!(count_first == 1 && n_general != 0)) { HIP_TRY(launch_classify(q, ctx->d_lists, ctx->d_class_counts, kNumPacked, -1, s)); }
            for (int c = 0; c < kNumPacked; ++c) {
                if (p.S > class_capacity(c)) continue; // N >= S: the list is empty by construction
                PassParams r = q;
                r.reroute_masks = nullptr;
                r.masks = ctx->d_masks;
                r.nmax = std::min(p.nmax, class_capacity(c));
                r.bmax = std::max(1, (int)std::sqrt((double)r.nmax));
                r.pix_list = ctx->d_lists + (size_t)c * HW;
                r.list_count = (uint32_t)((size_t)(p.row_end - p.row_begin) * p.W); // upper bound: sizes the grid
                HIP_TRY(launch_filter_packed(r, class_capacity(c), ctx->d_class_counts + c, s));
                if (launches) ++*launches;
            }
        }
        return launch_redo(ctx, p, s, launches);
    }
    ctx->last_route = 2;
    if ((st = ensure(ctx, ctx->d_lists, ctx->cap_lists, (size_t)kNumClasses * HW * sizeof(uint32_t)))) return st;
    // the count pass keeps its acceptance masks (one u64 per 64 candidates) so the filter kernels only rebuild the list
    PassParams pc = p;
    pc.mask_stride = (uint32_t)(((int64_t)(p.box * p.box - 1) * p.S + 63) / 64);
    if (pc.mask_stride == 0) pc.mask_stride = 1;
    if ((st = ensure(ctx, ctx->d_masks, ctx->cap_masks, HW * pc.mask_stride * sizeof(uint64_t)))) return st;
    pc.masks = ctx->d_masks;
    pc.carry = nullptr;
    if (p.nmax > class_capacity(kNumClasses - 4) && p.nmax <= kMaxResident && ctx->tun.split_weights != 0) {
        // the 32- and 64-spp classes run as three kernels (chains; bins + MI; weights): per-pixel hand-over buffer
        if ((st = ensure(ctx, ctx->d_carry, ctx->cap_carry, HW * (size_t)kCarryStride * sizeof(double)))) return st;
        pc.carry = ctx->d_carry;
    }
    uint32_t counts[kNumClasses];
    if (ctx->bin_valid && ctx->bin_box == p.box && ctx->bin_r0 == p.row_begin && ctx->bin_r1 == p.row_end) {
        std::memcpy(counts, ctx->bin_counts, sizeof(counts));
    } else {
        Range rg("rpf:count + classify (stage 1b test, size classes)");
        ctx->bin_valid = false;
        HIP_TRY(hipMemsetAsync(ctx->d_class_counts, 0, kNumClasses * sizeof(uint32_t), s));
        HIP_TRY(launch_nbhd_count(pc, 1, nullptr, nullptr, nullptr, 0, s));
        HIP_TRY(launch_classify(pc, ctx->d_lists, ctx->d_class_counts, kNumClasses, -1, s));
        HIP_TRY(hipMemcpyAsync(counts, ctx->d_class_counts, sizeof(counts), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::memcpy(ctx->bin_counts, counts, sizeof(counts));
        ctx->bin_box = p.box; ctx->bin_r0 = p.row_begin; ctx->bin_r1 = p.row_end;
        ctx->bin_valid = true;
    }
    // the four-wave kernels keep one acceptance mask per 64 candidates of the WINDOW in a 64-entry LDS array: windows
    // beyond 4096 candidates (boxes 17 and up) run their resident classes on the one-wave instantiations
    Tuning tun = ctx->tun;
    if ((int64_t)(p.box * p.box - 1) * p.S > 4096) tun.waves_per_pixel = 1;
    for (int c = 0; c < kNumClasses; ++c) {
        if (counts[c] == 0) continue;
        static const char *const kClassName[kNumClasses] = {
            "rpf:filter class N<=8", "rpf:filter class N<=16", "rpf:filter class N<=32", "rpf:filter class N<=64",
            "rpf:filter class N<=128", "rpf:filter class N<=256", "rpf:filter class N<=448",
            "rpf:filter class N<=832", "rpf:filter class N<=1600", "rpf:filter class N<=3136", "rpf:filter class streaming"};
        Range rg(kClassName[c]);
        PassParams q = pc;
        q.nmax = std::min(p.nmax, class_capacity(c));
        q.bmax = std::max(1, (int)std::sqrt((double)q.nmax));
        q.pix_list = ctx->d_lists + (size_t)c * HW;
        q.list_count = counts[c];
        if (c == kNumClasses - 1) { // neighbourhoods beyond the LDS-resident kernels: stream through global scratch
            const uint32_t slots = std::min<uint32_t>(counts[c], 1024u);
            if ((st = ensure(ctx, ctx->d_big_list, ctx->cap_big_list, (size_t)slots * q.nmax * 4))) return st;
            if ((st = ensure(ctx, ctx->d_big_bins, ctx->cap_big_bins, (size_t)slots * q.nmax * p.lay.ndim()))) return st;
            HIP_TRY(launch_filter_big(q, ctx->d_big_list, ctx->d_big_bins, slots, nullptr, s));
        } else if (c < kNumPacked && packed) {
            HIP_TRY(launch_filter_packed(q, class_capacity(c), nullptr, s));
        } else {
            HIP_TRY(launch_filter_pass(q, tun, s, nullptr));
        }
        if (launches) ++*launches;
    }
    return launch_redo(ctx, pc, s, launches);
}

// runs all passes of desc on device-resident buffers; colour ends up in d_colour
int32_t run_passes(rpf_ctx *ctx, const rpf_desc *d, const void *d_planes, double *d_colour, hipStream_t s) {
    const bool timing = (d->flags & RPF_FLAG_TIMING) != 0;
    const size_t ps = (size_t)d->W * d->H * d->S;
    int32_t st;
    ctx->bin_valid = false;
    if ((st = ensure(ctx, ctx->d_colB, ctx->cap_colB, 3 * ps * sizeof(double)))) return st;
    const int32_t init_status[2] = {0, INT_MAX};
    HIP_TRY(hipMemcpyAsync(ctx->d_status, init_status, sizeof(init_status), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(ctx->d_nred, 0, 2 * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(ctx->d_nan_flag, 0, sizeof(int32_t), s)); // stage 1a (pass 0) refills it, and the flat plane
    ctx->flat_fresh = true;
    rpf_counters &c = ctx->counters;
    c = rpf_counters{};
    c.first_bad_pixel = -1;
    float ms_filter = 0.f, ms_stats = 0.f;
    if (timing) HIP_TRY(hipEventRecord(ctx->ev[0], s));
    const size_t row = (size_t)d->W * d->S;
    double *cin = d_colour, *cout = ctx->d_colB; // ping-pong: the filtered colours replace the film's (rpf.cpp:732)
    for (int i = 0; i < d->n_box; ++i) {
        PassSetup ps_;
        if ((st = setup_pass(ctx, d, d->box_sizes[i], d_planes, cin, cout, nullptr, ps_))) return st;
        // rows outside the slab (halo) pass through unchanged
        HIP_TRY(launch_copy_colour_span(cin, cout, ps, 0, (uint64_t)d->row_begin * row, s));
        HIP_TRY(launch_copy_colour_span(cin, cout, ps, (uint64_t)d->row_end * row, (uint64_t)(d->H - d->row_end) * row, s));
        if (timing) HIP_TRY(hipEventRecord(ctx->ev[1], s));
        // stage 1a depends on the features only: formed once (the reference recomputes identical values per pass)
        if (i == 0) {
            Range rg("rpf:stage 1a pixel_stats");
            HIP_TRY(launch_pixel_stats(ps_.p, s));
        }
        if (timing) HIP_TRY(hipEventRecord(ctx->ev[2], s));
        if ((st = launch_filter_binned(ctx, ps_.p, s, &c.filter_kernel_launches))) return st;
        if (timing) {
            HIP_TRY(hipEventRecord(ctx->ev[3], s));
            HIP_TRY(hipEventSynchronize(ctx->ev[3]));
            float a = 0.f, b = 0.f;
            HIP_TRY(hipEventElapsedTime(&a, ctx->ev[1], ctx->ev[2]));
            HIP_TRY(hipEventElapsedTime(&b, ctx->ev[2], ctx->ev[3]));
            ms_stats += a;
            ms_filter += b;
        }
        std::swap(cin, cout);
    }
    if (cin != d_colour) HIP_TRY(hipMemcpyAsync(d_colour, cin, 3 * ps * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (timing) {
        HIP_TRY(hipEventRecord(ctx->ev[3], s));
        HIP_TRY(hipEventSynchronize(ctx->ev[3]));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[3]));
        c.device_total_ms = t;
    }
    c.filter_kernel_ms = ms_filter;
    c.stats_kernel_ms = ms_stats;
    return finish_counters(ctx, d, s);
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- host-buffer entry: row-band pipeline ----------------------------------------------------------------
// rpf_filter() receives the film in host memory.  The planes are [dim][y][x][s], so a band of rows is one
// contiguous span per plane: the image is cut into ~8 row bands, band j+1 is uploaded (s_up) while band j is
// filtered (compute stream), and in the last pass band j is reduced and downloaded (s_down) while band j+1 is
// filtered.  A band can be filtered once the b halo rows below it are resident, i.e. once the next band is up.
// The per-pixel feature statistics (stage 1a) depend on the features only, so they are formed once, in pass 0
// (the reference recomputes identical values every pass, rpf.cpp:529).
struct Band { int r0, r1; };

int32_t ensure_band_events(rpf_ctx *ctx, size_t n) {
    while (ctx->band_ev.size() < n) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->band_ev.push_back(e);
    }
    return RPF_OK;
}

int32_t finish_counters(rpf_ctx *ctx, const rpf_desc *d, hipStream_t s) {
    rpf_counters &c = ctx->counters;
    HIP_TRY(launch_nbhd_reduce(ctx->d_nbhd, d->W, d->row_begin, d->row_end, ctx->d_nred, s));
    int32_t hst[2];
    unsigned long long nred[2];
    uint32_t redo = 0;
    HIP_TRY(hipMemcpyAsync(hst, ctx->d_status, sizeof(hst), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(nred, ctx->d_nred, sizeof(nred), hipMemcpyDeviceToHost, s));
    if (d->degenerate_policy == RPF_DEGEN_REF_ABORT)
        HIP_TRY(hipMemcpyAsync(&redo, ctx->d_redo_count, sizeof(redo), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c.redo_pixels = (int32_t)redo;
    c.samples_filtered = (int64_t)(d->row_end - d->row_begin) * d->W * d->S * d->n_box;
    c.options_active = ctx->tun.is_default() ? 0 : 1;
    c.sum_nbhd = (int64_t)nred[0];
    c.max_nbhd = (int32_t)nred[1];
    c.nonfinite_pixels = hst[0];
    c.first_bad_pixel = hst[0] ? hst[1] : -1;
    if (hst[0] && d->degenerate_policy == RPF_DEGEN_REF_ABORT) {
        char buf[160];
        std::snprintf(buf, sizeof(buf), "non-finite filtered colour at pixel (x=%d, y=%d); %d pixel(s) affected "
                      "(the reference exits here, rpf.cpp:702-705)", hst[1] % d->W, hst[1] / d->W, hst[0]);
        return fail(ctx, RPF_E_NONFINITE, buf);
    }
    return RPF_OK;
}

int32_t run_host_pipeline(rpf_ctx *ctx, const rpf_desc *d, const void *planes_v, const float *ray_weight,
                          float *sample_rgb_out, float *pixel_rgb_out) {
    const int W = d->W, H = d->H, S = d->S;
    const size_t row = (size_t)W * S, ps = row * H;
    const SampleLayout lay = layout_of(d);
    const int kNDim = lay.ndim();
    const size_t pb = lay.plane_bytes();
    const char *planes = static_cast<const char *>(planes_v);
    hipStream_t s = ctx->stream, up = ctx->s_up, down = ctx->s_down;
    int32_t st;
    ctx->bin_valid = false;
    if ((st = ensure(ctx, ctx->d_colB, ctx->cap_colB, 3 * ps * sizeof(double)))) return st;

    // bands: about eight, never thinner than the widest halo of the first / last pass (or 16 rows)
    const int b_first = (d->box_sizes[0] - 1) / 2, b_last = (d->box_sizes[d->n_box - 1] - 1) / 2;
    const int min_rows = std::max(16, std::max(b_first, b_last));
    int nb = std::min(8, std::max(1, H / min_rows));
    const int bh = (H + nb - 1) / nb;
    nb = (H + bh - 1) / bh;
    std::vector<Band> bands(nb);
    for (int j = 0; j < nb; ++j) bands[j] = Band{j * bh, std::min(H, (j + 1) * bh)};
    if ((st = ensure_band_events(ctx, 2 * (size_t)nb))) return st;
    hipEvent_t *ev_up = ctx->band_ev.data(), *ev_done = ctx->band_ev.data() + nb;

    const int32_t init_status[2] = {0, INT_MAX};
    HIP_TRY(hipMemcpyAsync(ctx->d_status, init_status, sizeof(init_status), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(ctx->d_nred, 0, 2 * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(ctx->d_nan_flag, 0, sizeof(int32_t), s));
    ctx->flat_fresh = true;
    rpf_counters &c = ctx->counters;
    c = rpf_counters{};
    c.first_bad_pixel = -1;

    double *cin = ctx->d_colA, *cout = ctx->d_colB;
    const bool want_out = sample_rgb_out || pixel_rgb_out;
    const double t0 = now_ms();
    for (int i = 0; i < d->n_box; ++i) {
        const bool first = i == 0, last = i == d->n_box - 1;
        PassSetup ps_;
        if ((st = setup_pass(ctx, d, d->box_sizes[i], ctx->d_planes, cin, cout, nullptr, ps_))) return st;
        // rows outside the slab pass through (rpf.cpp filters whole films; slabs are this build's multi-GPU cut)
        auto pass_through = [&](int r0, int r1) -> hipError_t {
            const int a0 = r0, a1 = std::min(r1, d->row_begin), b0 = std::max(r0, d->row_end), b1 = r1;
            hipError_t e = hipSuccess;
            if (a1 > a0) e = launch_copy_colour_span(cin, cout, ps, (uint64_t)a0 * row, (uint64_t)(a1 - a0) * row, s);
            if (e == hipSuccess && b1 > b0)
                e = launch_copy_colour_span(cin, cout, ps, (uint64_t)b0 * row, (uint64_t)(b1 - b0) * row, s);
            return e;
        };
        auto filter_rows = [&](int r0, int r1) -> int32_t {
            PassParams q = ps_.p;
            q.row_begin = std::max(r0, d->row_begin);
            q.row_end = std::min(r1, d->row_end);
            HIP_TRY(pass_through(r0, r1));
            if (q.row_end > q.row_begin) return launch_filter_binned(ctx, q, s, &c.filter_kernel_launches);
            return RPF_OK;
        };
        auto emit_rows = [&](int j) -> int32_t { // last pass: reduce + download band j
            if (!want_out) return RPF_OK;
            Range rg("rpf:reduce + download band");
            const Band &bd = bands[j];
            HIP_TRY(launch_reduce_rows(cout, ray_weight ? ctx->d_rayw : nullptr, sample_rgb_out ? ctx->d_srgb : nullptr,
                                       pixel_rgb_out ? ctx->d_prgb : nullptr, W, H, S, bd.r0, bd.r1, s));
            HIP_TRY(hipEventRecord(ev_done[j], s));
            HIP_TRY(hipStreamWaitEvent(down, ev_done[j], 0));
            const size_t o = (size_t)bd.r0 * row, n = (size_t)(bd.r1 - bd.r0) * row;
            if (sample_rgb_out)
                for (int k = 0; k < 3; ++k)
                    HIP_TRY(hipMemcpyAsync(sample_rgb_out + k * ps + o, ctx->d_srgb + k * ps + o, n * sizeof(float),
                                           hipMemcpyDeviceToHost, down));
            if (pixel_rgb_out)
                HIP_TRY(hipMemcpyAsync(pixel_rgb_out + (size_t)bd.r0 * W * 3, ctx->d_prgb + (size_t)bd.r0 * W * 3,
                                       (size_t)(bd.r1 - bd.r0) * W * 3 * sizeof(float), hipMemcpyDeviceToHost, down));
            return RPF_OK;
        };
        if (!first && !last) { // middle passes: one launch over the slab
            if ((st = filter_rows(0, H))) return st;
        } else {
            for (int j = 0; j < nb; ++j) {
                const Band &bd = bands[j];
                if (first) {
                    Range rg("rpf:upload band + stage 1a");
                    const size_t o = (size_t)bd.r0 * row, n = (size_t)(bd.r1 - bd.r0) * row;
                    for (int k = 0; k < kNDim; ++k)
                        HIP_TRY(hipMemcpyAsync(ctx->d_planes + (k * ps + o) * pb, planes + (k * ps + o) * pb, n * pb,
                                               hipMemcpyHostToDevice, up));
                    if (ray_weight)
                        HIP_TRY(hipMemcpyAsync(ctx->d_rayw + o, ray_weight + o, n * sizeof(float), hipMemcpyHostToDevice, up));
                    HIP_TRY(hipEventRecord(ev_up[j], up));
                    HIP_TRY(hipStreamWaitEvent(s, ev_up[j], 0));
                    HIP_TRY(launch_colour_from_planes_span(ctx->d_planes, lay.f16 != 0, cin, ps, o, n, s));
                    HIP_TRY(launch_pixel_stats_rows(ps_.p, bd.r0, bd.r1, s));
                    if (j >= 1) { // band j-1 has its lower halo now
                        if ((st = filter_rows(bands[j - 1].r0, bands[j - 1].r1))) return st;
                        if (last && (st = emit_rows(j - 1))) return st;
                    }
                } else {
                    if ((st = filter_rows(bd.r0, bd.r1))) return st;
                    if ((st = emit_rows(j))) return st;
                }
            }
            if (first) {
                if ((st = filter_rows(bands[nb - 1].r0, bands[nb - 1].r1))) return st;
                if (last && (st = emit_rows(nb - 1))) return st;
            }
        }
        std::swap(cin, cout);
    }
    // cin now names the buffer holding the final colours; keep the convention "result in d_colA"
    if (cin != ctx->d_colA) std::swap(ctx->d_colA, ctx->d_colB), std::swap(ctx->cap_colA, ctx->cap_colB);
    const int32_t fst = finish_counters(ctx, d, s);
    HIP_TRY(hipStreamSynchronize(down));
    HIP_TRY(hipStreamSynchronize(up));
    c.device_total_ms = (float)(now_ms() - t0); // wall clock of the overlapped upload + passes + download
    return fst;
}

} // namespace

extern "C" {

const char *rpf_version(void) { return "rpf_hip 0.1 (gfx950)"; }

const char *rpf_status_string(int32_t s) {
    switch (s) {
    case RPF_OK: return "RPF_OK";
    case RPF_E_BADARG: return "RPF_E_BADARG";
    case RPF_E_HIP: return "RPF_E_HIP";
    case RPF_E_NONFINITE: return "RPF_E_NONFINITE";
    case RPF_E_NOMEM: return "RPF_E_NOMEM";
    case RPF_E_UNSUPPORTED: return "RPF_E_UNSUPPORTED";
    case RPF_E_NODEVICE: return "RPF_E_NODEVICE";
    default: return "RPF_E_?";
    }
}

int32_t rpf_create(rpf_ctx **out, int32_t device) {
    if (!out) return RPF_E_BADARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RPF_E_NODEVICE;
    if (device < 0 || device >= n) return RPF_E_BADARG;
    rpf_ctx *ctx = new rpf_ctx();
    ctx->device = device;
    *out = ctx; // returned even on failure so that rpf_last_error() can be read; caller destroys it
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->s_up, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->s_down, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void **)&ctx->d_status, 2 * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->d_nred, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void **)&ctx->d_class_counts, (kNumClasses + 2) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->d_nan_flag, sizeof(int32_t)));
    HIP_TRY(hipMemset(ctx->d_nan_flag, 0, sizeof(int32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->d_redo_count, sizeof(uint32_t)));
    HIP_TRY(hipMemset(ctx->d_redo_count, 0, sizeof(uint32_t)));
    for (auto &e : ctx->ev) HIP_TRY(hipEventCreate(&e));
    return RPF_OK;
}

void rpf_destroy(rpf_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void *bufs[] = {ctx->d_planes, ctx->d_rayw, ctx->d_colA, ctx->d_colB, ctx->d_pmean, ctx->d_pstd, ctx->d_nbhd,
                    ctx->d_tfix, ctx->d_dfix, ctx->d_srgb, ctx->d_prgb, ctx->d_status, ctx->d_nred, ctx->d_lists,
                    ctx->d_class_counts, ctx->d_masks, ctx->d_big_list, ctx->d_big_bins, ctx->d_carry, ctx->d_redo_list,
                    ctx->d_redo_count, ctx->d_flat, ctx->d_nan_flag};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (void *b : ctx->d_dbg)
        if (b) (void)hipFree(b);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->band_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t q : {ctx->s_up, ctx->s_down, ctx->stream})
        if (q) (void)hipStreamDestroy(q);
    delete ctx;
}

const char *rpf_last_error(const rpf_ctx *ctx) { return ctx ? ctx->err.c_str() : "ctx is NULL"; }

int64_t rpf_lds_bytes_required(int32_t S, int32_t box) {
    if (S <= 0 || box <= 0) return -1;
    const int64_t nmax = (int64_t)box * box * S;
    if (nmax > 49 * 64) return -1;
    const int nm = (int)nmax;
    int bmax = (int)std::sqrt((double)nm);
    if (bmax < 1) bmax = 1;
    const Tuning tun;
    const SampleLayout lay;
    return lds_layout(S, nm, bmax, table_in_lds(S, nm, bmax, tun, lay), tun, lay).total;
}

int32_t rpf_colour_from_planes_device(rpf_ctx *ctx, const rpf_desc *d, const void *d_planes, double *d_colour,
                                      void *stream) {
    int32_t st = validate(ctx, d, false);
    if (st) return st;
    if (!d_planes || !d_colour) return fail(ctx, RPF_E_BADARG, "NULL device pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream; // NULL = the legacy default stream: ordered after the caller's own work
    HIP_TRY(launch_colour_from_planes(d_planes, layout_of(d).f16 != 0, d_colour, (uint64_t)d->W * d->H * d->S, s));
    return RPF_OK;
}

int32_t rpf_reduce_device(rpf_ctx *ctx, const rpf_desc *d, const double *d_colour, const float *d_ray_weight,
                          float *d_sample_rgb_out, float *d_pixel_rgb_out, void *stream) {
    int32_t st = validate(ctx, d, false);
    if (st) return st;
    if (!d_colour) return fail(ctx, RPF_E_BADARG, "NULL device pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream; // NULL = the legacy default stream: ordered after the caller's own work
    HIP_TRY(launch_reduce(d_colour, d_ray_weight, d_sample_rgb_out, d_pixel_rgb_out, d->W, d->H, d->S, s));
    return RPF_OK;
}

int32_t rpf_filter_device(rpf_ctx *ctx, const rpf_desc *d, const void *d_planes, double *d_colour, void *stream) {
    int32_t st = validate(ctx, d, true);
    if (st) return st;
    if (!d_planes || !d_colour) return fail(ctx, RPF_E_BADARG, "NULL device pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream; // NULL = the legacy default stream: ordered after the caller's own work
    return run_passes(ctx, d, d_planes, d_colour, s);
}

int32_t rpf_host_alloc(rpf_ctx *ctx, uint64_t bytes, void **out) {
    if (!ctx) return RPF_E_BADARG;
    if (!out) return fail(ctx, RPF_E_BADARG, "out is NULL");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(out, bytes ? (size_t)bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess)
        return fail(ctx, e == hipErrorOutOfMemory ? RPF_E_NOMEM : RPF_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return RPF_OK;
}

int32_t rpf_host_free(rpf_ctx *ctx, void *ptr) { // ctx may be NULL (a buffer can outlive its context)
    if (!ptr) return RPF_OK;
    const hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) return fail(ctx, RPF_E_HIP, std::string("hipHostFree: ") + hipGetErrorString(e));
    return RPF_OK;
}

int32_t rpf_set_option(rpf_ctx *ctx, const char *name, int64_t value) {
    if (!ctx) return RPF_E_BADARG;
    if (!name) return fail(ctx, RPF_E_BADARG, "option name is NULL");
    const std::string n(name);
    Tuning &t = ctx->tun;
    if (n == "stage_mask") t.stage_mask = (int32_t)value;
    else if (n == "binning" && value >= -1 && value <= 1) t.binning = (int32_t)value;
    else if (n == "waves_per_pixel" && (value == 0 || value == 1 || value == 4)) t.waves_per_pixel = (int32_t)value;
    else if (n == "table_in_lds" && value >= -1 && value <= 1) t.table_in_lds = (int32_t)value;
    else if (n == "screen" && value >= 0 && value <= 1) t.screen = (int32_t)value;
    else if (n == "split_weights" && value >= -1 && value <= 1) t.split_weights = (int32_t)value;
    else if (n == "strip_w" && value >= 0 && value <= 4096 && value % 8 == 0) t.strip_w = (int32_t)value;
    else if (n == "packed" && value >= -1 && value <= 1) t.packed = (int32_t)value;
    else if (n == "split_chunk" && value >= 0 && value <= (1 << 30)) t.split_chunk = (int32_t)value;
    else if (n == "count_first" && value >= -1 && value <= 1) t.count_first = (int32_t)value;
    else if (n == "lds_pad" && value >= 0 && value <= 160 * 1024) t.lds_pad = (int32_t)value;
    else return fail(ctx, RPF_E_BADARG, "unknown option or value out of range: " + n);
    ctx->bin_valid = false;
    return RPF_OK;
}

int32_t rpf_filter(rpf_ctx *ctx, const rpf_desc *d, const void *planes, const float *ray_weight,
                   float *sample_rgb_out, float *pixel_rgb_out) {
    return rpf_filter_ex(ctx, d, planes, nullptr, ray_weight, sample_rgb_out, pixel_rgb_out, nullptr);
}

int32_t rpf_filter_ex(rpf_ctx *ctx, const rpf_desc *d, const void *planes, const double *colour64_in,
                      const float *ray_weight, float *sample_rgb_out, float *pixel_rgb_out, double *colour64_out) {
    int32_t st = validate(ctx, d, true);
    if (st) return st;
    if (!planes) return fail(ctx, RPF_E_BADARG, "planes is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t ps = (size_t)d->W * d->H * d->S, HW = (size_t)d->W * d->H;
    const SampleLayout lay = layout_of(d);
    const size_t plane_total = (size_t)lay.ndim() * ps * lay.plane_bytes();
    if ((st = ensure(ctx, ctx->d_planes, ctx->cap_planes, plane_total))) return st;
    if ((st = ensure(ctx, ctx->d_colA, ctx->cap_colA, 3 * ps * sizeof(double)))) return st;
    if (ray_weight && (st = ensure(ctx, ctx->d_rayw, ctx->cap_rayw, ps * sizeof(float)))) return st;
    if (sample_rgb_out && (st = ensure(ctx, ctx->d_srgb, ctx->cap_srgb, 3 * ps * sizeof(float)))) return st;
    if (pixel_rgb_out && (st = ensure(ctx, ctx->d_prgb, ctx->cap_prgb, 3 * HW * sizeof(float)))) return st;
    // the band pipeline needs asynchronous copies, i.e. page-locked buffers on the host side (rpf_host_alloc or the
    // caller's own hipHostMalloc / hipHostRegister); with pageable memory every copy blocks the submitting thread
    // and the serial order is faster (scripts/host_path.py)
    auto pinned = [](const void *p) {
        if (!p) return true;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    if (!(d->flags & (RPF_FLAG_TIMING | RPF_FLAG_NO_OVERLAP)) && !colour64_in && !colour64_out && pinned(planes) &&
        pinned(ray_weight) && pinned(sample_rgb_out) && pinned(pixel_rgb_out))
        return run_host_pipeline(ctx, d, planes, ray_weight, sample_rgb_out, pixel_rgb_out);
    // serial variant (per-kernel event timing needs it): upload, passes, download
    const double t0 = now_ms();
    {
        Range rg("rpf:upload");
        HIP_TRY(hipMemcpyAsync(ctx->d_planes, planes, plane_total, hipMemcpyHostToDevice, s));
        if (ray_weight) HIP_TRY(hipMemcpyAsync(ctx->d_rayw, ray_weight, ps * sizeof(float), hipMemcpyHostToDevice, s));
        if (colour64_in) HIP_TRY(hipMemcpyAsync(ctx->d_colA, colour64_in, 3 * ps * sizeof(double), hipMemcpyHostToDevice, s));
        else HIP_TRY(launch_colour_from_planes(ctx->d_planes, lay.f16 != 0, ctx->d_colA, ps, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    const double t1 = now_ms();
    const int32_t fst = run_passes(ctx, d, ctx->d_planes, ctx->d_colA, s);
    if (fst != RPF_OK && fst != RPF_E_NONFINITE) return fst;
    const double t2 = now_ms();
    if (colour64_out) {
        HIP_TRY(hipMemcpyAsync(colour64_out, ctx->d_colA, 3 * ps * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (sample_rgb_out || pixel_rgb_out) {
        Range rg("rpf:reduce + download");
        HIP_TRY(launch_reduce(ctx->d_colA, ray_weight ? ctx->d_rayw : nullptr, sample_rgb_out ? ctx->d_srgb : nullptr,
                              pixel_rgb_out ? ctx->d_prgb : nullptr, d->W, d->H, d->S, s));
        if (sample_rgb_out)
            HIP_TRY(hipMemcpyAsync(sample_rgb_out, ctx->d_srgb, 3 * ps * sizeof(float), hipMemcpyDeviceToHost, s));
        if (pixel_rgb_out)
            HIP_TRY(hipMemcpyAsync(pixel_rgb_out, ctx->d_prgb, 3 * HW * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    ctx->counters.h2d_ms = (float)(t1 - t0);
    ctx->counters.d2h_ms = (float)(now_ms() - t2);
    return fst;
}

int32_t rpf_stage_pixel_stats(rpf_ctx *ctx, const rpf_desc *d, const void *planes, double *mean, double *stddev) {
    int32_t st = validate(ctx, d, false);
    if (st) return st;
    if (!planes || !mean || !stddev) return fail(ctx, RPF_E_BADARG, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t ps = (size_t)d->W * d->H * d->S, HW = (size_t)d->W * d->H;
    const SampleLayout lay = layout_of(d);
    const int kNFeat = lay.nF;
    const size_t plane_total = (size_t)lay.ndim() * ps * lay.plane_bytes();
    if ((st = ensure(ctx, ctx->d_planes, ctx->cap_planes, plane_total))) return st;
    if ((st = ensure(ctx, ctx->d_pmean, ctx->cap_pmean, HW * kNFeat * sizeof(double)))) return st;
    if ((st = ensure(ctx, ctx->d_pstd, ctx->cap_pstd, HW * kNFeat * sizeof(double)))) return st;
    HIP_TRY(hipMemcpyAsync(ctx->d_planes, planes, plane_total, hipMemcpyHostToDevice, s));
    PassParams p{};
    p.lay = lay;
    p.W = d->W; p.H = d->H; p.S = d->S; p.policy = d->degenerate_policy;
    p.plane_stride = ps; p.planes = ctx->d_planes; p.pmean = ctx->d_pmean; p.pstd = ctx->d_pstd;
    HIP_TRY(launch_pixel_stats(p, s));
    std::vector<double> m(HW * kNFeat), sd(HW * kNFeat);
    HIP_TRY(hipMemcpyAsync(m.data(), ctx->d_pmean, m.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(sd.data(), ctx->d_pstd, sd.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t pix = 0; pix < HW; ++pix) // device planes are [12][H*W]; the ABI is pixel-major
        for (int k = 0; k < kNFeat; ++k) {
            mean[pix * kNFeat + k] = m[(size_t)k * HW + pix];
            stddev[pix * kNFeat + k] = sd[(size_t)k * HW + pix];
        }
    return RPF_OK;
}

int32_t rpf_filter_pass_debug(rpf_ctx *ctx, const rpf_desc *d, int32_t box, const void *planes,
                              const double *colour_in, double *colour_out, const rpf_debug *dbg) {
    int32_t st = validate(ctx, d, false);
    if (st) return st;
    if (!planes || !colour_out) return fail(ctx, RPF_E_BADARG, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t ps = (size_t)d->W * d->H * d->S, HW = (size_t)d->W * d->H;
    const SampleLayout lay = layout_of(d);
    const size_t kNDim = (size_t)lay.ndim(), kNFeat = (size_t)lay.nF, kNPair = (size_t)lay.npair();
    const size_t plane_total = kNDim * ps * lay.plane_bytes();
    if ((st = ensure(ctx, ctx->d_planes, ctx->cap_planes, plane_total))) return st;
    if ((st = ensure(ctx, ctx->d_colA, ctx->cap_colA, 3 * ps * sizeof(double)))) return st;
    if ((st = ensure(ctx, ctx->d_colB, ctx->cap_colB, 3 * ps * sizeof(double)))) return st;
    HIP_TRY(hipMemcpyAsync(ctx->d_planes, planes, plane_total, hipMemcpyHostToDevice, s));
    if (colour_in)
        HIP_TRY(hipMemcpyAsync(ctx->d_colA, colour_in, 3 * ps * sizeof(double), hipMemcpyHostToDevice, s));
    else
        HIP_TRY(launch_colour_from_planes(ctx->d_planes, lay.f16 != 0, ctx->d_colA, ps, s));
    // debug planes
    const size_t dbg_bytes[9] = {HW * 4, HW * kNDim * 8, HW * kNDim * 8, HW * kNPair * 8, HW * 3 * 8,
                                 HW * kNFeat * 8, HW * 8, HW * kNDim * 4, HW * 4};
    void *host_dbg[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (dbg) {
        void *tmp[9] = {dbg->nbhd_size, dbg->mean, dbg->stddev, dbg->mi, dbg->alpha, dbg->beta, dbg->wrc,
                        dbg->bin_hash, dbg->member_hash};
        std::memcpy(host_dbg, tmp, sizeof(tmp));
    }
    rpf_debug dev{};
    void **dev_slots[9] = {(void **)&dev.nbhd_size, (void **)&dev.mean, (void **)&dev.stddev, (void **)&dev.mi,
                           (void **)&dev.alpha, (void **)&dev.beta, (void **)&dev.wrc, (void **)&dev.bin_hash,
                           (void **)&dev.member_hash};
    for (int i = 0; i < 9; ++i) {
        if (!host_dbg[i]) continue;
        if ((st = ensure(ctx, ctx->d_dbg[i], ctx->cap_dbg[i], dbg_bytes[i]))) return st;
        HIP_TRY(hipMemsetAsync(ctx->d_dbg[i], 0, dbg_bytes[i], s));
        *dev_slots[i] = ctx->d_dbg[i];
    }
    const int32_t init_status[2] = {0, INT_MAX};
    HIP_TRY(hipMemcpyAsync(ctx->d_status, init_status, sizeof(init_status), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(ctx->d_nred, 0, 2 * sizeof(unsigned long long), s));
    PassSetup ps_;
    if ((st = setup_pass(ctx, d, box, ctx->d_planes, ctx->d_colA, ctx->d_colB, &dev, ps_))) return st;
    HIP_TRY(launch_copy_f64(ctx->d_colA, ctx->d_colB, 3 * ps, s));
    const bool timing = (d->flags & RPF_FLAG_TIMING) != 0;
    HIP_TRY(hipMemsetAsync(ctx->d_nan_flag, 0, sizeof(int32_t), s));
    ctx->flat_fresh = true;
    HIP_TRY(launch_pixel_stats(ps_.p, s));
    if (timing) HIP_TRY(hipEventRecord(ctx->ev[0], s));
    int n_launch = 0;
    ctx->bin_valid = false;
    if ((st = launch_filter_binned(ctx, ps_.p, s, &n_launch))) return st;
    if (timing) HIP_TRY(hipEventRecord(ctx->ev[1], s));
    HIP_TRY(launch_nbhd_reduce(ctx->d_nbhd, d->W, d->row_begin, d->row_end, ctx->d_nred, s));
    HIP_TRY(hipMemcpyAsync(colour_out, ctx->d_colB, 3 * ps * sizeof(double), hipMemcpyDeviceToHost, s));
    if (dbg && dbg->nbhd_size) // N is always produced in the context's own plane
        HIP_TRY(hipMemcpyAsync(ctx->d_dbg[0], ctx->d_nbhd, HW * 4, hipMemcpyDeviceToDevice, s));
    for (int i = 0; i < 9; ++i)
        if (host_dbg[i]) HIP_TRY(hipMemcpyAsync(host_dbg[i], ctx->d_dbg[i], dbg_bytes[i], hipMemcpyDeviceToHost, s));
    int32_t hst[2];
    unsigned long long nred[2];
    uint32_t redo = 0;
    HIP_TRY(hipMemcpyAsync(hst, ctx->d_status, sizeof(hst), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(nred, ctx->d_nred, sizeof(nred), hipMemcpyDeviceToHost, s));
    if (d->degenerate_policy == RPF_DEGEN_REF_ABORT)
        HIP_TRY(hipMemcpyAsync(&redo, ctx->d_redo_count, sizeof(redo), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    rpf_counters &c = ctx->counters;
    c = rpf_counters{};
    c.redo_pixels = (int32_t)redo;
    c.samples_filtered = (int64_t)(d->row_end - d->row_begin) * d->W * d->S;
    c.options_active = ctx->tun.is_default() ? 0 : 1;
    c.sum_nbhd = (int64_t)nred[0];
    c.max_nbhd = (int32_t)nred[1];
    c.nonfinite_pixels = hst[0];
    c.first_bad_pixel = hst[0] ? hst[1] : -1;
    c.filter_kernel_launches = n_launch;
    if (timing) HIP_TRY(hipEventElapsedTime(&c.filter_kernel_ms, ctx->ev[0], ctx->ev[1]));
    if (hst[0] && d->degenerate_policy == RPF_DEGEN_REF_ABORT)
        return fail(ctx, RPF_E_NONFINITE, "non-finite filtered colour (the reference exits here, rpf.cpp:702-705)");
    return RPF_OK;
}

int32_t rpf_feature_images(rpf_ctx *ctx, const rpf_desc *d, const void *planes, double *images_out) {
    int32_t st = validate(ctx, d, false);
    if (st) return st;
    if (!planes || !images_out) return fail(ctx, RPF_E_BADARG, "NULL pointer");
    if (!layout_of(d).is_ref19()) return fail(ctx, RPF_E_UNSUPPORTED, "visualizeSF's six images are defined for the reference's 19-dim layout");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t ps = (size_t)d->W * d->H * d->S, HW = (size_t)d->W * d->H;
    const size_t kNDim = RPF_NDIM;
    if ((st = ensure(ctx, ctx->d_planes, ctx->cap_planes, kNDim * ps * sizeof(float)))) return st;
    if ((st = ensure(ctx, ctx->d_dbg[1], ctx->cap_dbg[1], (18 * HW + 18) * sizeof(double)))) return st;
    double *d_img = (double *)ctx->d_dbg[1];
    unsigned long long *d_max = (unsigned long long *)(d_img + 18 * HW);
    HIP_TRY(hipMemcpyAsync(ctx->d_planes, planes, kNDim * ps * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(launch_feature_images(reinterpret_cast<const float *>(ctx->d_planes), d->W, d->H, d->S, d_img, d_max, s));
    HIP_TRY(hipMemcpyAsync(images_out, d_img, 18 * HW * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RPF_OK;
}

int32_t rpf_selftest_udiv(rpf_ctx *ctx, uint64_t n, uint64_t seed, int32_t mode, uint64_t *mismatches) {
    if (!ctx || !mismatches) return RPF_E_BADARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->d_nred, 0, 2 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(launch_udiv_selftest(n, seed, mode, ctx->d_nred, ctx->stream));
    unsigned long long r = 0;
    HIP_TRY(hipMemcpyAsync(&r, ctx->d_nred, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *mismatches = r;
    return RPF_OK;
}

int32_t rpf_query_counters(rpf_ctx *ctx, rpf_counters *out) {
    if (!ctx || !out) return RPF_E_BADARG;
    *out = ctx->counters;
    return RPF_OK;
}

int32_t rpf_query_nbhd(rpf_ctx *ctx, int32_t *nbhd_out, int64_t count) {
    if (!ctx) return RPF_E_BADARG;
    if (!nbhd_out || count <= 0) return fail(ctx, RPF_E_BADARG, "nbhd_out is NULL or count <= 0");
    if (!ctx->d_nbhd || (size_t)count * sizeof(int32_t) > ctx->cap_nbhd)
        return fail(ctx, RPF_E_BADARG, "no neighbourhood plane of that size: run a filter call first (count = W*H)");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(nbhd_out, ctx->d_nbhd, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost));
    return RPF_OK;
}

int32_t rpf_query_route(rpf_ctx *ctx, int32_t *route_out) {
    if (!ctx) return RPF_E_BADARG;
    if (!route_out) return fail(ctx, RPF_E_BADARG, "route_out is NULL");
    *route_out = ctx->last_route;
    return RPF_OK;
}

} // extern "C"

// ---- one process, several GPUs: row slabs behind the ABI ---------------------------------------------------------
// The reference's caller is one process (RPFIntegrator::Render, rpf.cpp:737-805); rpf_multi lets that one caller use
// every GPU of the node.  The image is cut into contiguous row slabs, one per entry of `devices` (an entry may repeat:
// two slabs on one GPU rehearse the multi-GPU path on a one-GPU box); slab g holds its rows plus `halo` rows of each
// neighbour, halo = max over the box list of (box-1)/2 (rpf.cpp:561).  Features never change, so their halo travels
// with the upload; colours change every pass, so before pass i >= 1 every slab's halo rows are refreshed from the
// neighbour's OWNED boundary rows with hipMemcpyPeerAsync (xGMI when peer access is available, staged otherwise; a
// plain device copy when both slabs share a GPU).  Passes run concurrently, one host thread per slab.
struct rpf_multi {
    std::vector<rpf_ctx *> ctx;
    std::vector<int> dev;
    std::string err;
    rpf_counters counters{};
};

namespace {

struct MSlab { int a, b, ht, hb; int rows() const { return ht + (b - a) + hb; } }; // owned image rows [a,b), halo rows held

int32_t mfail(rpf_multi *m, int32_t st, const std::string &msg) {
    if (m) m->err = msg;
    return st;
}

} // namespace

extern "C" {

int32_t rpf_multi_create(rpf_multi **out, const int32_t *devices, int32_t n_devices) {
    if (!out) return RPF_E_BADARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RPF_E_NODEVICE;
    rpf_multi *m = new rpf_multi();
    *out = m; // returned even on failure so that rpf_multi_last_error() can be read
    std::vector<int> devs;
    if (devices && n_devices > 0) devs.assign(devices, devices + n_devices);
    else for (int i = 0; i < n; ++i) devs.push_back(i); // NULL / 0: every visible device
    for (int d : devs) {
        rpf_ctx *c = nullptr;
        const int32_t st = rpf_create(&c, d);
        if (st != RPF_OK) {
            const std::string e = c ? c->err : std::string("no such device");
            if (c) rpf_destroy(c);
            return mfail(m, st, "rpf_create(device " + std::to_string(d) + "): " + e);
        }
        m->ctx.push_back(c);
        m->dev.push_back(d);
    }
    // direct peer copies between neighbouring slabs where the hardware offers them (failure = staged copies: still correct)
    for (size_t g = 0; g + 1 < devs.size(); ++g) {
        const int a = devs[g], b = devs[g + 1];
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) { (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0); }
        if (hipDeviceCanAccessPeer(&can, b, a) == hipSuccess && can) { (void)hipSetDevice(b); (void)hipDeviceEnablePeerAccess(a, 0); }
        (void)hipGetLastError(); // "already enabled" is not an error here
    }
    return RPF_OK;
}

void rpf_multi_destroy(rpf_multi *m) {
    if (!m) return;
    for (rpf_ctx *c : m->ctx) rpf_destroy(c);
    delete m;
}

const char *rpf_multi_last_error(const rpf_multi *m) { return m ? m->err.c_str() : "multi context is NULL"; }
int32_t rpf_multi_device_count(const rpf_multi *m) { return m ? (int32_t)m->ctx.size() : 0; }

int32_t rpf_multi_set_option(rpf_multi *m, const char *name, int64_t value) {
    if (!m) return RPF_E_BADARG;
    for (rpf_ctx *c : m->ctx) {
        const int32_t st = rpf_set_option(c, name, value);
        if (st != RPF_OK) return mfail(m, st, c->err);
    }
    return RPF_OK;
}

int32_t rpf_multi_query_counters(rpf_multi *m, rpf_counters *out) {
    if (!m || !out) return RPF_E_BADARG;
    *out = m->counters;
    return RPF_OK;
}

int32_t rpf_multi_filter(rpf_multi *m, const rpf_desc *d, const void *planes_v, const float *ray_weight,
                         float *sample_rgb_out, float *pixel_rgb_out) {
    if (!m || m->ctx.empty()) return RPF_E_BADARG;
    {
        const int32_t st = validate(m->ctx[0], d, true);
        if (st != RPF_OK) return mfail(m, st, m->ctx[0]->err);
    }
    if (!planes_v) return mfail(m, RPF_E_BADARG, "planes is NULL");
    if (d->row_begin != 0 || d->row_end != d->H)
        return mfail(m, RPF_E_BADARG, "rpf_multi_filter filters the whole image (row_begin = 0, row_end = H): the slabs are its own");
    const int G = (int)m->ctx.size(), W = d->W, H = d->H, S = d->S;
    int halo = 0;
    for (int i = 0; i < d->n_box; ++i) halo = std::max(halo, (d->box_sizes[i] - 1) / 2);
    std::vector<MSlab> sl(G);
    for (int g = 0; g < G; ++g) {
        sl[g].a = (int)((int64_t)g * H / G);
        sl[g].b = (int)((int64_t)(g + 1) * H / G);
        sl[g].ht = std::min(halo, sl[g].a);
        sl[g].hb = std::min(halo, H - sl[g].b);
        if (G > 1 && sl[g].b - sl[g].a < halo)
            return mfail(m, RPF_E_BADARG, "a row slab is thinner than the halo its neighbours need (H / devices < (box-1)/2): use fewer devices");
    }
    const SampleLayout lay = layout_of(d);
    const int ND = lay.ndim();
    const size_t pb = lay.plane_bytes(), row = (size_t)W * S, ps_img = row * H;
    const char *planes = static_cast<const char *>(planes_v);
    std::vector<double *> cin(G), cout(G);
    std::vector<int32_t> status(G, RPF_OK);
    std::vector<rpf_desc> sd(G, *d);

    // ---- upload: every slab's rows (+ halo rows) of every plane; colours seeded on the device ------------------------
    auto per_slab = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (int g = 0; g < G; ++g) th.emplace_back([&, g] { status[g] = fn(g); });
        for (auto &t : th) t.join();
        for (int g = 0; g < G; ++g)
            if (status[g] != RPF_OK && status[g] != RPF_E_NONFINITE) return mfail(m, status[g], "slab " + std::to_string(g) + ": " + m->ctx[g]->err);
        return (int32_t)RPF_OK;
    };
    int32_t st = per_slab([&](int g) -> int32_t {
        rpf_ctx *ctx = m->ctx[g];
        HIP_TRY(hipSetDevice(ctx->device));
        const MSlab &q = sl[g];
        const size_t ps = row * q.rows(), HW = (size_t)W * q.rows();
        rpf_desc &ds = sd[g];
        ds.H = q.rows(); ds.row_begin = q.ht; ds.row_end = q.ht + (q.b - q.a); ds.n_box = 1;
        int32_t e;
        if ((e = ensure(ctx, ctx->d_planes, ctx->cap_planes, (size_t)ND * ps * pb))) return e;
        if ((e = ensure(ctx, ctx->d_colA, ctx->cap_colA, 3 * ps * sizeof(double)))) return e;
        if ((e = ensure(ctx, ctx->d_colB, ctx->cap_colB, 3 * ps * sizeof(double)))) return e;
        if (ray_weight && (e = ensure(ctx, ctx->d_rayw, ctx->cap_rayw, ps * sizeof(float)))) return e;
        if (sample_rgb_out && (e = ensure(ctx, ctx->d_srgb, ctx->cap_srgb, 3 * ps * sizeof(float)))) return e;
        if (pixel_rgb_out && (e = ensure(ctx, ctx->d_prgb, ctx->cap_prgb, 3 * HW * sizeof(float)))) return e;
        hipStream_t s = ctx->stream;
        const size_t o = (size_t)(q.a - q.ht) * row;
        for (int k = 0; k < ND; ++k)
            HIP_TRY(hipMemcpyAsync(ctx->d_planes + (size_t)k * ps * pb, planes + ((size_t)k * ps_img + o) * pb, ps * pb,
                                   hipMemcpyHostToDevice, s));
        if (ray_weight) HIP_TRY(hipMemcpyAsync(ctx->d_rayw, ray_weight + o, ps * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(launch_colour_from_planes(ctx->d_planes, lay.f16 != 0, ctx->d_colA, ps, s));
        const int32_t init_status[2] = {0, INT_MAX};
        HIP_TRY(hipMemcpyAsync(ctx->d_status, init_status, sizeof(init_status), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(ctx->d_nan_flag, 0, sizeof(int32_t), s));
        HIP_TRY(hipStreamSynchronize(s));
        ctx->flat_fresh = true;
        ctx->bin_valid = false;
        ctx->counters = rpf_counters{};
        cin[g] = ctx->d_colA; cout[g] = ctx->d_colB;
        return RPF_OK;
    });
    if (st != RPF_OK) return st;

    float ms_filter = 0.f;
    int launches = 0;
    for (int i = 0; i < d->n_box; ++i) {
        const int box = d->box_sizes[i];
        // ---- colour halo refresh from the neighbours' owned rows (pass 0: the upload already carried it) ----------
        if (i > 0 && G > 1) {
            Range rg("rpf:colour halo refresh (peer copies)");
            for (int g = 0; g + 1 < G; ++g) {
                rpf_ctx *up = m->ctx[g], *dn = m->ctx[g + 1];
                const size_t ps_u = row * sl[g].rows(), ps_d = row * sl[g + 1].rows();
                const size_t hb = (size_t)sl[g].hb * row, ht = (size_t)sl[g + 1].ht * row; // == halo rows on both sides
                for (int c = 0; c < 3; ++c) {
                    // bottom halo of slab g <- first owned rows of slab g+1
                    const double *src1 = cin[g + 1] + c * ps_d + (size_t)sl[g + 1].ht * row;
                    double *dst1 = cin[g] + c * ps_u + (size_t)(sl[g].ht + sl[g].b - sl[g].a) * row;
                    // top halo of slab g+1 <- last owned rows of slab g
                    const double *src2 = cin[g] + c * ps_u + (size_t)(sl[g].ht + sl[g].b - sl[g].a) * row - ht;
                    double *dst2 = cin[g + 1] + c * ps_d;
                    hipError_t e1, e2;
                    if (up->device == dn->device) {
                        (void)hipSetDevice(up->device);
                        e1 = hipMemcpyAsync(dst1, src1, hb * sizeof(double), hipMemcpyDeviceToDevice, up->stream);
                        e2 = hipMemcpyAsync(dst2, src2, ht * sizeof(double), hipMemcpyDeviceToDevice, up->stream);
                    } else {
                        (void)hipSetDevice(up->device); // each copy is queued with its stream's device current
                        e1 = hipMemcpyPeerAsync(dst1, up->device, src1, dn->device, hb * sizeof(double), up->stream);
                        (void)hipSetDevice(dn->device);
                        e2 = hipMemcpyPeerAsync(dst2, dn->device, src2, up->device, ht * sizeof(double), dn->stream);
                    }
                    if (e1 != hipSuccess || e2 != hipSuccess)
                        return mfail(m, RPF_E_HIP, std::string("halo copy: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
                }
            }
            for (int g = 0; g < G; ++g) { // every copy has landed before any slab starts the pass
                (void)hipSetDevice(m->ctx[g]->device);
                if (hipStreamSynchronize(m->ctx[g]->stream) != hipSuccess) return mfail(m, RPF_E_HIP, "halo copy synchronise");
            }
        }
        // ---- the pass, all slabs concurrently ------------------------------------------------------------------------
        std::vector<float> ms(G, 0.f);
        std::vector<int> nl(G, 0);
        st = per_slab([&](int g) -> int32_t {
            rpf_ctx *ctx = m->ctx[g];
            HIP_TRY(hipSetDevice(ctx->device));
            hipStream_t s = ctx->stream;
            const MSlab &q = sl[g];
            const size_t ps = row * q.rows();
            PassSetup pp;
            int32_t e;
            if ((e = setup_pass(ctx, &sd[g], box, ctx->d_planes, cin[g], cout[g], nullptr, pp))) return e;
            // halo rows pass through (they are refreshed from the neighbour before the next pass)
            HIP_TRY(launch_copy_colour_span(cin[g], cout[g], ps, 0, (uint64_t)q.ht * row, s));
            HIP_TRY(launch_copy_colour_span(cin[g], cout[g], ps, (uint64_t)(q.ht + q.b - q.a) * row, (uint64_t)q.hb * row, s));
            if (i == 0) HIP_TRY(launch_pixel_stats(pp.p, s)); // stage 1a depends on the features only
            HIP_TRY(hipEventRecord(ctx->ev[0], s));
            if ((e = launch_filter_binned(ctx, pp.p, s, &nl[g]))) return e;
            HIP_TRY(hipEventRecord(ctx->ev[1], s));
            HIP_TRY(hipEventSynchronize(ctx->ev[1]));
            HIP_TRY(hipEventElapsedTime(&ms[g], ctx->ev[0], ctx->ev[1]));
            return RPF_OK;
        });
        if (st != RPF_OK) return st;
        float mx = 0.f;
        for (int g = 0; g < G; ++g) { mx = std::max(mx, ms[g]); launches += nl[g]; std::swap(cin[g], cout[g]); }
        ms_filter += mx;
    }

    // ---- reduce + download the owned rows; merge status and counters -----------------------------------------------
    rpf_counters tot{};
    tot.first_bad_pixel = -1;
    std::vector<rpf_counters> cs(G);
    st = per_slab([&](int g) -> int32_t {
        rpf_ctx *ctx = m->ctx[g];
        HIP_TRY(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        const MSlab &q = sl[g];
        const size_t ps = row * q.rows(), own0 = (size_t)q.ht * row, own_n = (size_t)(q.b - q.a) * row;
        HIP_TRY(hipMemsetAsync(ctx->d_nred, 0, 2 * sizeof(unsigned long long), s));
        if (sample_rgb_out || pixel_rgb_out) {
            HIP_TRY(launch_reduce_rows(cin[g], ray_weight ? ctx->d_rayw : nullptr, sample_rgb_out ? ctx->d_srgb : nullptr,
                                       pixel_rgb_out ? ctx->d_prgb : nullptr, W, q.rows(), S, q.ht, q.ht + (q.b - q.a), s));
            if (sample_rgb_out)
                for (int c = 0; c < 3; ++c)
                    HIP_TRY(hipMemcpyAsync(sample_rgb_out + c * ps_img + (size_t)q.a * row, ctx->d_srgb + c * ps + own0,
                                           own_n * sizeof(float), hipMemcpyDeviceToHost, s));
            if (pixel_rgb_out)
                HIP_TRY(hipMemcpyAsync(pixel_rgb_out + (size_t)q.a * W * 3, ctx->d_prgb + (size_t)q.ht * W * 3,
                                       (size_t)(q.b - q.a) * W * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
        }
        rpf_desc one = sd[g];
        one.n_box = d->n_box; // samples_filtered counts every pass
        const int32_t fst = finish_counters(ctx, &one, s); // synchronises
        cs[g] = ctx->counters;
        return fst;
    });
    if (st != RPF_OK) return st;
    bool bad = false;
    for (int g = 0; g < G; ++g) {
        const rpf_counters &c = cs[g];
        tot.samples_filtered += c.samples_filtered;
        tot.sum_nbhd += c.sum_nbhd;
        tot.nonfinite_pixels += c.nonfinite_pixels;
        tot.max_nbhd = std::max(tot.max_nbhd, c.max_nbhd);
        tot.options_active |= c.options_active;
        tot.redo_pixels += c.redo_pixels;
        if (c.first_bad_pixel >= 0) { // slab-local y*W+x -> image index
            const int yl = c.first_bad_pixel / W, x = c.first_bad_pixel % W;
            const int gi = (sl[g].a - sl[g].ht + yl) * W + x;
            if (tot.first_bad_pixel < 0 || gi < tot.first_bad_pixel) tot.first_bad_pixel = gi;
        }
        bad = bad || status[g] == RPF_E_NONFINITE;
    }
    tot.filter_kernel_ms = ms_filter; // per pass: the slowest slab
    tot.filter_kernel_launches = launches;
    m->counters = tot;
    if (bad) {
        char buf[160];
        std::snprintf(buf, sizeof(buf), "non-finite filtered colour at pixel (x=%d, y=%d); %lld pixel(s) affected (the reference exits here, "
                      "rpf.cpp:702-705)", tot.first_bad_pixel % W, tot.first_bad_pixel / W, (long long)tot.nonfinite_pixels);
        return mfail(m, RPF_E_NONFINITE, buf);
    }
    return RPF_OK;
}

} // extern "C"
