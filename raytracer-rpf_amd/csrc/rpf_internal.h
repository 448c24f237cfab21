// rpf_internal.h -- shared between the kernel TU (rpf_kernels.hip) and the C-ABI TU (rpf_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rpf_hip.h"

namespace rpf {

constexpr int kWave = 64;

// Which sample-vector layout a call uses (rpf_desc n_random / n_feat / plane_dtype).  Columns: [0,2) pFilm | [2,5) colour
// | [5,5+nR) random parameters | [5+nR, 5+nR+nF) features (sd.h:62-94 is nR = 2, nF = 12).  The kernels exist for the
// reference layout with fp32 planes and for BASELINE configs[4]'s 27 dims (nR = 4, nF = 18) with fp16 planes.
struct SampleLayout {
    int32_t nR = 2, nF = 12, f16 = 0;
    int ndim() const { return 5 + nR + nF; }
    int npair() const { return nF * (nR + 2) + 3 * (nR + 2 + nF); } // rpf.cpp:416-442 generalised
    int nwt() const { return 5 + nF; }                              // weighted columns of stage 4
    size_t plane_bytes() const { return f16 ? 2 : 4; }
    bool is_ref19() const { return nR == 2 && nF == 12 && !f16; }
    bool supported() const { return is_ref19() || (nR == 4 && nF == 18 && f16 == 1); }
};
constexpr int kStageChunk = 64; // samples gathered per step of the in-order (reference-order) sums
constexpr int kStageHalf = 32;  // ... and staged through LDS this many at a time

// everything one pass needs, passed by value to the kernels
struct PassParams {
    int32_t W, H, S;
    int32_t row_begin, row_end;
    int32_t box, b;        // b = (box-1)/2, rpf.cpp:561
    int32_t beta_map, policy;
    int32_t fast_weights;  // RPF_FLAG_FAST_WEIGHTS: fp32 pair arithmetic in stage 4
    int32_t stage_mask;    // diagnostics only (rpf_set_option "stage_mask"): bit0 stats chain, bit1 bins, bit2 MI, bit3 weights; -1 = all
    int32_t screen;        // far-pair screen of the four-wave kernels: 0 off, 1 on
    int32_t strip_w;       // pixels per XCD strip of the pixel walk (slab_pixel)
    int32_t nmax;          // box*box*S: capacity of a neighbourhood
    int32_t bmax;          // floor(sqrt(nmax)): max histogram bins per axis
    double eps, seed, sigma_p;
    uint64_t plane_stride; // H*W*S
    SampleLayout lay;
    const void *planes;    // ndim planes of fp32 (or fp16: lay.f16); the colour planes are only read to seed d_colour
    const double *col_in;  // 3 fp64 planes
    double *col_out;       // 3 fp64 planes
    const double *pmean;   // [12][H*W] stage 1a
    const double *pstd;    // [12][H*W]
    const uint64_t *tfix;  // round(k ln k * 2^44), k = 0..nmax
    const uint64_t *dfix;  // tfix[k+1] - tfix[k], k = 0..nmax-1
    int32_t *nbhd;         // [H*W] N per pixel (always written)
    double *carry;            // split route of the 32- / 64-spp classes: per-pixel statistics and weights between its three kernels (kCarryStride doubles per pixel), or null
    const uint32_t *pix_list; // size-binned launch: the pixels (y*W+x) this launch filters, or null = every pixel of the slab
    uint32_t list_count;
    uint64_t *masks;       // size-binned launch: acceptance masks of stage 1b, [H*W][mask_stride] (one per 64 candidates,
    uint32_t mask_stride;  //   written by nbhd_count_kernel, re-used by the filter kernels), or null
    // unbinned route (box*box*S <= 512): filter_pixel_kernel leaves the pixels whose neighbourhood turns out small (N <= 64) to
    // the packed kernels -- it writes its acceptance masks and N and exits; classify_kernel then deals those pixels into the
    // lane-class lists (one atomic per wave and class: a per-pixel append on one counter cost 11 ns per pixel, 22 ms a frame)
    uint64_t *reroute_masks;   // [H*W][mask_stride], or null = no re-routing
    // stage 1a's by-product: flat[pix] = 1 when some feature of the pixel has sigma == 0 (and a finite mean) -- the strict 3-sigma
    // test then rejects every finite candidate (flat_quad_shortcut) -- and *nan_flag != 0 when any feature mean of the buffer is
    // NaN (the one kind of candidate that would still pass).  flat && !*nan_flag proves N = S without touching a sample.
    uint8_t *flat;         // [H*W], or null
    int32_t *nan_flag;     // [1]
    uint32_t *redo_list;   // REF_ABORT: pixels whose MI stage met a table inside the rounding band at a non-power-of-two N are
    uint32_t *redo_count;  //   appended here and filtered again by filter_pixel_big_kernel (reference expression); or null
    int32_t *status;       // [0] count of NaN pixels, [1] lowest bad pixel index (atomicMin)
    rpf_debug dbg;         // device pointers, any may be null
};

// Per-context tuning / diagnostic overrides (rpf_set_option).  Defaults = the library's own choices; nothing here is
// read from the environment.  stage_mask != -1 skips stages (timing ablation: results are wrong) and is reported in
// rpf_counters.options_active.
struct Tuning {
    int32_t waves_per_pixel = 0; // 0 auto, 1 or 4
    int32_t table_in_lds = -1;   // -1 auto, 0 / 1
    int32_t lds_pad = 0;         // extra LDS bytes per workgroup (occupancy experiments)
    int32_t binning = -1;        // -1 auto (box*box*S > 512), 0 / 1
    int32_t stage_mask = -1;     // bit0 stats chain, bit1 bins, bit2 MI, bit3 weights; -1 = all
    int32_t screen = 1;          // far-pair screen (stage 4, four-wave kernels): 0 off, 1 on; same results
    int32_t split_weights = -1;  // 32- / 64-spp classes as three kernels (chains; bins + MI; weights): -1 auto (on), 0 off, 1 on; same results
    int32_t split_chunk = 0;     // split route: run its three launches chunk by chunk over this many list entries (0 = the whole list at once); same results
    int32_t strip_w = 0;         // pixels per XCD strip of the pixel walk: 0 auto (by box and spp), else a multiple of 8; same results
    int32_t count_first = -1;    // box*box*S <= 512: stage 1b as its own launch ahead of the filter kernels (the small-N route): -1 auto (probe), 0 off, 1 on; same results
    int32_t packed = -1;         // small neighbourhoods (N <= 64) on the packed kernels, several pixels per wave: -1 auto (on), 0 off, 1 on
    bool is_default() const { return waves_per_pixel == 0 && table_in_lds == -1 && lds_pad == 0 && binning == -1 && stage_mask == -1 && screen == 1 && strip_w == 0 && split_chunk == 0 && count_first == -1 && split_weights == -1 && packed == -1; }
};

struct LdsLayout {
    uint32_t off_T, off_stat, off_hx, off_pair, off_mi, off_own, off_off, off_union, off_hist, total;
    uint32_t hist_stride; // bytes of one wave's histogram buffer
    uint32_t nw;          // waves per pixel (1 or 4)
};
LdsLayout lds_layout(int S, int nmax, int bmax, bool t_in_lds, const Tuning &tun, const SampleLayout &lay);
LdsLayout lds_layout_weights(int S, int nmax, const SampleLayout &lay, int nw); // the weight kernel of the split route (32- / 64-spp classes)
LdsLayout lds_layout_chains(int S, int nmax, const SampleLayout &lay, int nw);  // ... and its chain kernel (4 or 8 waves per pixel)
constexpr int kCarryStride = 136; // doubles per pixel of PassParams::carry (>= kCarry of either layout)
int samples_per_lane(int nmax); // the K the filter kernel is instantiated with (0 = unsupported)
bool table_in_lds(int S, int nmax, int bmax, const Tuning &tun, const SampleLayout &lay);
int waves_per_pixel(int nmax, const Tuning &tun);

hipError_t launch_pixel_stats(const PassParams &p, hipStream_t s);
hipError_t launch_pixel_stats_rows(const PassParams &p, int r0, int r1, hipStream_t s);
hipError_t launch_filter_pass(const PassParams &p, const Tuning &tun, hipStream_t s, uint32_t *lds_bytes_out);
// neighbourhood-size binning (large box*box*S): count N per pixel, then deal the pixels into one list per kernel family
constexpr int kNumPacked = 4;       // lane classes of the packed kernels: N <= 8, 16, 32, 64 (8, 4, 2, 1 pixels per wave)
constexpr int kNumClasses = 11;     // four packed lane classes (one-wave K = 1 kernel when the packed route is off), six more
                                    // LDS-resident kernel families, the streaming kernel for larger neighbourhoods
constexpr int kMaxResident = 3136;  // largest neighbourhood the LDS-resident kernels hold (64 lanes x 49 samples)
constexpr int kMaxNbhd = 65535;     // the streaming kernel: 16-bit histogram cells, one-byte bin ids
int class_capacity(int c);          // 8, 16, 32, 64, 128, 256, 448, 832, 1600, 3136, 65535
// the packed kernels (rpf_packed_impl.inc): pixels of p.pix_list with N <= lanes_per_pixel; count_dev != null: the list size
// is read on the device and p.list_count is only its upper bound
hipError_t launch_filter_packed(const PassParams &p, int lanes_per_pixel, const uint32_t *count_dev, hipStream_t s);
// stage 1b's test (N and the acceptance masks): for every pixel of the slab (step 1, list null); for the entries of `list`
// (size on the device in *list_count, at most list_max: sizes the grid); or for the points of a lattice of pitch `step`, with
// probe[0] += how many of them have N <= 64 without being proven flat and probe[1] += how many are proven flat
hipError_t launch_nbhd_count(const PassParams &p, int step, uint32_t *probe, const uint32_t *list, const uint32_t *list_count,
                             uint32_t list_max, hipStream_t s);
// the streaming kernel (neighbourhoods of the last size class): global scratch of `slots` workgroups,
// list [slots][nmax] u32 and bins [slots][ndim][nmax] u8
// count_dev != null: the size of p.pix_list is read from device memory (redo mode: no host read-back), grid = slots
hipError_t launch_filter_big(const PassParams &p, void *list, void *bins, uint32_t slots, const uint32_t *count_dev, hipStream_t s);
// max_class < kNumClasses: pixels of that class and above join the list of rest_class (-1: they are left out)
hipError_t launch_classify(const PassParams &p, uint32_t *lists /*[kNumClasses][H*W]*/, uint32_t *counts /*[kNumClasses], zeroed*/,
                           int max_class, int rest_class, hipStream_t s);
hipError_t launch_colour_from_planes(const void *planes, bool f16, double *colour, uint64_t plane_stride, hipStream_t s);
hipError_t launch_colour_from_planes_span(const void *planes, bool f16, double *colour, uint64_t plane_stride, uint64_t e0,
                                          uint64_t cnt, hipStream_t s);
hipError_t launch_copy_f64(const double *src, double *dst, uint64_t n, hipStream_t s);
hipError_t launch_copy_colour_span(const double *src, double *dst, uint64_t plane_stride, uint64_t e0, uint64_t cnt,
                                   hipStream_t s);
hipError_t launch_reduce_rows(const double *colour, const float *ray_weight, float *sample_rgb, float *pixel_rgb, int W,
                              int H, int S, int r0, int r1, hipStream_t s);
hipError_t launch_reduce(const double *colour, const float *ray_weight, float *sample_rgb, float *pixel_rgb, int W,
                         int H, int S, hipStream_t s);
// unbinned route: deals the pixels of the slab (slab_pixel order) into the list of those that need the fused kernel's own
// stage 1b; a proven flat pixel gets nbhd = S instead (the packed kernels take it from there)
hipError_t launch_prelist(const PassParams &p, uint32_t *list, uint32_t *count /* zeroed */, hipStream_t s);
hipError_t launch_nbhd_reduce(const int32_t *nbhd, int W, int row_begin, int row_end, unsigned long long *out2,
                              hipStream_t s);
hipError_t launch_feature_images(const float *planes, int W, int H, int S, double *out, unsigned long long *maxbits, hipStream_t s);
int max_lds_per_block();
hipError_t launch_udiv_selftest(uint64_t n, uint64_t seed, int mode, unsigned long long *d_mismatch, hipStream_t s);

} // namespace rpf
