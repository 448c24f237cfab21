/*
 * rpf_hip.h -- C ABI of librpf_hip.so: the MI355X (gfx950) Random Parameter Filtering pass.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference (tux550/RayTracer-RPF, a
 * pbrt-v3 fork) has no plugin ABI: its integrator calls the private member
 *     void RPFIntegrator::ApplyRPFFilter(SamplingFilm&, const int tileSize, int box_size)
 *                                                   /root/reference/src/custom/rpf.h:91-95, rpf.cpp:497-733
 * once per box size from RPFIntegrator::Render (rpf.cpp:767-775) and then reduces the filtered samples
 * into the film (rpf.cpp:779-804).  A maintainer replaces the body of ApplyRPFFilter (or the loop in
 * Render) with one call to rpf_filter(); INTEGRATION.md shows the binding.  No pbrt type crosses this
 * boundary: plain pointers, sizes and POD structs only, no exceptions, integer status codes.
 *
 * Data layout on both sides of the ABI: 19 SoA planes of fp32, plane d at base + d*H*W*S, element
 * (y, x, s) at ((y*W)+x)*S + s, dims = SampleData::data (sd.h:62-94):
 *     0,1 pFilm | 2,3,4 L rgb | 5,6 pLens | 7..9 n0 | 10..12 p0 | 13..15 n1 | 16..18 p1
 * (the reference stores the same 19 values as doubles in samples[x][y][s], sample_film.cpp:32-42; they
 * are fp32-valued because pbrt's Float is float.)  Colours travel between passes as fp64 planes on the
 * device, exactly as the reference carries them in SampleData doubles.
 *
 * Threading: one caller thread per rpf_ctx at a time; distinct contexts are independent.
 * Multi-GPU: one context per device; a context filters a ROW SLAB [row_begin,row_end) of a buffer that also holds
 * the halo rows it needs.  A strict sub-slab is filtered ONE pass per call (n_box == 1) with a colour-halo exchange
 * between passes: raytracer-rpf_amd/slabs.py does that across processes (RCCL send/recv).
 */
#ifndef RPF_HIP_H
#define RPF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPF_NDIM 19   /* the reference's sample vector (sd.h:21-49); other layouts: rpf_desc.n_random / n_feat */
#define RPF_NFEAT 12
#define RPF_NPAIR 96
#define RPF_MAX_BOXES 8
/* general layout: columns [0,2) pFilm | [2,5) colour | [5,5+nR) random parameters | [5+nR,5+nR+nF) features */
#define RPF_NDIM_OF(nR, nF) (5 + (nR) + (nF))
#define RPF_NPAIR_OF(nR, nF) ((nF) * ((nR) + 2) + 3 * ((nR) + 2 + (nF))) /* MI pairs, rpf.cpp:416-442 generalised */
enum { RPF_PLANES_F32 = 0, RPF_PLANES_F16 = 1 };

typedef struct rpf_ctx rpf_ctx;

/* status codes; replaces the reference's exit(1) on NaN (rpf.cpp:702-705) and its silent preconditions */
typedef enum rpf_status {
    RPF_OK = 0,
    RPF_E_BADARG = 1,      /* malformed descriptor / NULL pointer / even box / S<=0 ...            */
    RPF_E_HIP = 2,         /* a HIP runtime call failed; text in rpf_last_error()                  */
    RPF_E_NONFINITE = 3,   /* REF_ABORT policy: a filtered colour came out NaN (reference aborts)  */
    RPF_E_NOMEM = 4,
    RPF_E_UNSUPPORTED = 5, /* neighbourhood too large for the LDS-resident kernel  */
    RPF_E_NODEVICE = 6
} rpf_status;

/* which D term feeds W_c_fk[k] for k = 0..11.  rpf.cpp:464 indexes the 3-element D_f_ck with k < 12
 * (undefined behaviour); the presets name what each build of the reference reads. */
typedef enum rpf_beta_map {
    RPF_BETA_REF_GCC11_O3 = 0, /* {Dfc[0..2], 0, Drf[0..7]}         g++ 11.4 -O3 (CMake Release)   */
    RPF_BETA_REF_GCC11_O2 = 1, /* {Dfc[0..2], 0,0,0,0,0, Drf[0..3]} g++ 11.4 -O0/-O2               */
    RPF_BETA_PAPER = 2         /* sum_c MI(c_c, f_k): the formula in the comment at rpf.cpp:459    */
} rpf_beta_map;

typedef enum rpf_degenerate_policy {
    RPF_DEGEN_REF_ABORT = 0, /* IEEE propagation exactly as the reference; NaN colour => RPF_E_NONFINITE */
    RPF_DEGEN_EPS = 1        /* documented deviation: +eps in the three denominators of rpf.cpp:464-470,
                                negative variances clamped to 0, NaN colours fall back to the input    */
} rpf_degenerate_policy;

enum {
    RPF_FLAG_NONE = 0,
    RPF_FLAG_TIMING = 1,       /* bracket kernels with hipEvents (rpf_query_counters) */
    RPF_FLAG_FAST_WEIGHTS = 2, /* opt-in: the S x N pair weights of stage 4 (rpf.cpp:637-678) are evaluated in fp32 on
                                  fp64-formed normalised values, with the hardware exp; everything that decides
                                  discrete outcomes (membership, bins, MI) is unchanged.  Colours move by ~1e-6
                                  relative (bar 1e-4).  Default off: fp64 throughout, like the reference. */
    RPF_FLAG_NO_OVERLAP = 4    /* rpf_filter(): upload, filter and download one after the other instead of the
                                  row-band pipeline (same results; for A/B timing).  RPF_FLAG_TIMING implies it. */
};

typedef struct rpf_desc {
    int32_t W;                         /* pixels per row                                            */
    int32_t H;                         /* rows present in the buffers (owned + halo)                */
    int32_t S;                         /* samples per pixel, identical for every pixel              */
    int32_t row_begin;                 /* first row to filter                                       */
    int32_t row_end;                   /* one past the last row to filter                           */
    int32_t n_box;                     /* number of passes (> 1 only when the slab is the whole buffer) */
    int32_t box_sizes[RPF_MAX_BOXES];  /* odd box sizes, one per pass (reference: {7}, rpf.cpp:767)  */
    int32_t beta_map;                  /* rpf_beta_map                                              */
    int32_t degenerate_policy;         /* rpf_degenerate_policy                                     */
    int32_t flags;
    double eps;                        /* RPF_DEGEN_EPS epsilon (1e-10)                             */
    double sigma_seed;                 /* rpf.cpp:533: 0.002                                        */
    /* Sample-vector layout.  All three 0 = the reference's: 2 random parameters (pLens), 12 features, fp32 planes.
     * Kernels also exist for n_random = 4, n_feat = 18 with RPF_PLANES_F16 (27 dims, fp16 feature storage: BASELINE
     * configs[4]); anything else returns RPF_E_UNSUPPORTED.  With fp16 planes every `planes` pointer of this header
     * addresses 16-bit IEEE halves instead of floats; colours are still carried as fp64, outputs stay fp32.
     * Per-pixel debug planes are then sized by RPF_NDIM_OF / RPF_NPAIR_OF / n_feat. */
    int32_t n_random;
    int32_t n_feat;
    int32_t plane_dtype;               /* RPF_PLANES_F32 / RPF_PLANES_F16                           */
    int32_t reserved;
} rpf_desc;

/* per-pixel stage outputs for parity tests; every pointer may be NULL; indexed [y*W+x] */
typedef struct rpf_debug {
    int32_t *nbhd_size;    /* [H*W]      N, stage 1b (rpf.cpp:556-586)                              */
    double *mean;          /* [H*W*19]   neighbourhood mean,  stage 2 (rpf.cpp:600)                  */
    double *stddev;        /* [H*W*19]   neighbourhood std,   stage 2 (rpf.cpp:601)                  */
    double *mi;            /* [H*W*96]   MI values in ComputeCFWeights call order (rpf.cpp:416-442)  */
    double *alpha;         /* [H*W*3]    rpf.cpp:474-476                                             */
    double *beta;          /* [H*W*12]   rpf.cpp:478-480                                             */
    double *wrc;           /* [H*W]      rpf.cpp:483-487                                             */
    uint32_t *bin_hash;    /* [H*W*19]   FNV-1a over each column's histogram bin ids (mi.cpp:14-16)  */
    uint32_t *member_hash; /* [H*W]      FNV-1a over the neighbourhood member list, in order         */
} rpf_debug;

typedef struct rpf_counters {
    int64_t samples_filtered;   /* (row_end-row_begin)*W*S*n_box of the last call                    */
    int64_t sum_nbhd;           /* sum over filtered pixels of N, last pass                         */
    int64_t nonfinite_pixels;   /* pixels whose filtered colour was NaN, all passes                 */
    int32_t max_nbhd;
    int32_t first_bad_pixel;    /* y*W+x, -1 if none                                                */
    float filter_kernel_ms;     /* RPF_FLAG_TIMING: sum over passes of the fused filter kernel      */
    float stats_kernel_ms;      /* RPF_FLAG_TIMING: sum over passes of the per-pixel stats kernel   */
    float device_total_ms;      /* RPF_FLAG_TIMING: first launch to last kernel end                 */
    float h2d_ms;               /* rpf_filter(): host->HBM marshalling, wall clock                  */
    float d2h_ms;               /* rpf_filter(): HBM->host                                          */
    int32_t filter_kernel_launches;
    int32_t options_active;     /* 1 when any rpf_set_option override was in force (diagnostic runs)   */
    int32_t redo_pixels;        /* RPF_DEGEN_REF_ABORT, last pass: pixels filtered a second time with the reference's own
                                   floating-point MI expression (an exactly independent histogram pair at a non-power-of-two
                                   N: mi.cpp:79-86 returns rounding residue there, not 0)                */
} rpf_counters;

const char *rpf_version(void);
const char *rpf_status_string(int32_t status);

/* lifetime.  device = HIP device ordinal of this process. */
int32_t rpf_create(rpf_ctx **out, int32_t device);
void rpf_destroy(rpf_ctx *ctx);
const char *rpf_last_error(const rpf_ctx *ctx);

/*
 * The drop-in call: replaces the `for (box_size : box_sizes) ApplyRPFFilter(...)` loop of
 * rpf.cpp:767-775 plus the per-pixel reduction of rpf.cpp:779-794 (box reconstruction filter r=0.5).
 *   planes          host, 19 fp32 planes [19][H][W][S]
 *   ray_weight      host, [H][W][S] fp32 (SampleData::rayWeight, sd.h:60) or NULL (= 1)
 *   sample_rgb_out  host, 3 fp32 planes [3][H][W][S] of filtered sample colours, or NULL
 *   pixel_rgb_out   host, [H][W][3] fp32 mean over s of colour*rayWeight (rows outside the slab: unfiltered), or NULL
 */
int32_t rpf_filter(rpf_ctx *ctx, const rpf_desc *desc, const void *planes, const float *ray_weight,
                   float *sample_rgb_out, float *pixel_rgb_out);

/* rpf_filter() with the sample colours carried as doubles across the boundary, as the reference carries them in
 * SampleData (sd.h:205-208 getColorI / setColorI; the film that one ApplyRPFFilter call leaves is the input of the next,
 * rpf.cpp:732, 767-775).  A caller that keeps the reference's call shape -- one ApplyRPFFilter(film, tile, box) per box
 * size -- uses this so that no colour is rounded to fp32 between passes:
 *   colour64_in   host, 3 fp64 planes [3][H][W][S], or NULL (= planes 2..4 of `planes`)
 *   colour64_out  host, 3 fp64 planes of filtered colours, or NULL
 * With either pointer set the call runs upload, passes, download one after the other (no row-band overlap). */
int32_t rpf_filter_ex(rpf_ctx *ctx, const rpf_desc *desc, const void *planes, const double *colour64_in,
                      const float *ray_weight, float *sample_rgb_out, float *pixel_rgb_out, double *colour64_out);

/* Per-context tuning / diagnostic overrides (nothing in the library reads the environment).  Names:
 *   "stage_mask"       -1 = all stages (default); other values SKIP stages for timing ablations: results are WRONG
 *   "binning"          -1 auto (size-binned launches when box*box*S > 512), 0 off, 1 on
 *   "waves_per_pixel"  0 auto, 1, 4 (4 needs box*box*S > 832)
 *   "table_in_lds"     -1 auto, 0, 1
 *   "lds_pad"          extra LDS bytes per workgroup (lowers occupancy)
 *   "split_weights"    32- and 64-spp size classes: three launches (in-order chains; bins + MI; weights) instead of one
 *                      kernel, the light stages at two to three times the occupancy: -1 auto (default: on), 0 off, 1 on;
 *                      same results bit for bit
 *   "strip_w"          pixels per XCD strip of the pixel walk: 0 auto (default), else a multiple of 8; same results
 *   "packed"           neighbourhoods of N <= 64 samples on the packed kernels (8 / 4 / 2 / 1 pixels per wavefront): -1 auto
 *                      (default: on), 0 off (every pixel gets a whole wavefront), 1 on.  Every stage output up to alpha / beta /
 *                      W_r_c is the same bits either way; colours agree to rounding (~1e-16 relative)
 *   "count_first"      passes with box*box*S <= 512: stage 1b as its own launch ahead of the filter kernels (the route of
 *                      small-neighbourhood buffers) or inside filter_pixel_kernel: -1 auto (default: a ~2000-pixel probe
 *                      decides per pass), 0 fused, 1 count first.  Same results bit for bit; rpf_query_route tells.
 *   "screen"           far-pair screen of the weight stage (four-wave kernels): 1 on (default), 0 off.  Both settings
 *                      give the same filtered colours bit for bit.
 * rpf_counters.options_active tells whether a result was produced under any override. */
int32_t rpf_set_option(rpf_ctx *ctx, const char *name, int64_t value);

/* Page-locked host memory for the buffers handed to rpf_filter(): a feature producer that writes its samples straight
 * into such planes (instead of the reference's heap SamplingFilm, sample_film.cpp:6-43) gets full-rate DMA and real
 * overlap of the upload with the first filter pass.  Pageable buffers work too, more slowly. */
int32_t rpf_host_alloc(rpf_ctx *ctx, uint64_t bytes, void **out);
int32_t rpf_host_free(rpf_ctx *ctx, void *ptr); /* ctx may be NULL */

/* Same pass structure with every buffer already resident in HBM (device pointers).  d_colour is 3 fp64
 * planes [3][H][W][S], read as the input colours and overwritten with the filtered ones; planes 2..4 of
 * d_planes are ignored.  Runs on `stream`: the hipStream_t on which the caller produced the buffers (NULL = the
 * legacy default stream, e.g. PyTorch's default stream), so the pass is ordered after that work without an explicit
 * synchronisation.  Returns after the stream has drained (the status and the counters are read back). */
int32_t rpf_filter_device(rpf_ctx *ctx, const rpf_desc *desc, const void *d_planes, double *d_colour,
                          void *stream);

/* fp32 colour planes (planes 2..4 of d_planes) -> fp64 colour planes; and back, plus the pixel mean */
int32_t rpf_colour_from_planes_device(rpf_ctx *ctx, const rpf_desc *desc, const void *d_planes,
                                      double *d_colour, void *stream);
int32_t rpf_reduce_device(rpf_ctx *ctx, const rpf_desc *desc, const double *d_colour, const float *d_ray_weight,
                          float *d_sample_rgb_out, float *d_pixel_rgb_out, void *stream);

/* ---- stage-level entry points (host buffers; used by the parity tests) ---------------------------- */

/* stage 1a, FillMeanAndStddev (rpf.cpp:302-353): mean/std [H*W*12] fp64, pixel-major */
int32_t rpf_stage_pixel_stats(rpf_ctx *ctx, const rpf_desc *desc, const void *planes, double *mean, double *stddev);

/* one pass with one box size; colour_in (3 fp64 planes) may be NULL (= planes 2..4); colour_out 3 fp64
 * planes; dbg host pointers, any may be NULL */
int32_t rpf_filter_pass_debug(rpf_ctx *ctx, const rpf_desc *desc, int32_t box, const void *planes,
                              const double *colour_in, double *colour_out, const rpf_debug *dbg);

/* counters of the most recent rpf_filter / rpf_filter_device / rpf_filter_pass_debug call */
int32_t rpf_query_counters(rpf_ctx *ctx, rpf_counters *out);

/* neighbourhood size N of every pixel (rpf.cpp:586: the neighbourhood vector's size) as the last pass of the most recent
 * call left it: nbhd_out host, int32 [H*W] (rows outside the filtered slab: whatever an earlier call left there);
 * count must equal desc W*H of that call.  For workload statistics (mean / percentiles of N). */
int32_t rpf_query_nbhd(rpf_ctx *ctx, int32_t *nbhd_out, int64_t count);

/* which kernel route the last pass of the most recent call took (a performance decision, the results are the same bits):
 * 0 = fused (filter_pixel_kernel runs stage 1b itself), 1 = count first (stage 1b as its own launch, then the packed
 * small-neighbourhood kernels take most pixels: the route of path-traced buffers), 2 = size-binned
 * (box*box*S > 512), -1 = no pass yet.  Option "count_first" (0 / 1) overrides the probe that chooses between 0 and 1. */
int32_t rpf_query_route(rpf_ctx *ctx, int32_t *route_out);

/* visualizeSF (rpf.cpp:37-101, visualization/vis.cpp:34-51): the reference's six debug images, without the EXR
 * writer: per-pixel mean over the S samples of n0, n1, p0, p1, (pFilm.x, pFilm.y, 0), (pLens.x, pLens.y, 0), each
 * channel divided by its maximum over the image.  images_out: host, fp64 [6][H][W][3] in that order. */
int32_t rpf_feature_images(rpf_ctx *ctx, const rpf_desc *desc, const void *planes, double *images_out);

/* device self-test: the kernels divide by wave-uniform divisors with a hoisted reciprocal (3 instructions per
 * quotient); this compares n pseudo-random quotients bit-for-bit with the compiler's IEEE fp64 division.
 * mode 0: operand magnitudes of the filter (2^-40..2^40); mode 1: 2^-600..2^600 (exercises the fallback). */
int32_t rpf_selftest_udiv(rpf_ctx *ctx, uint64_t n, uint64_t seed, int32_t mode, uint64_t *mismatches);

/* ---- one caller, every GPU of the node -------------------------------------------------------------------------
 * The reference's caller is a single process (RPFIntegrator::Render, rpf.cpp:737-805, reached from api.cpp:1620).  A
 * rpf_multi owns one context per entry of `devices` (NULL / 0 = every visible device; an ordinal may repeat: two slabs
 * on one GPU rehearse the path on a one-GPU box) and rpf_multi_filter() is rpf_filter() for the whole image: it cuts
 * the image into contiguous row slabs (rows [g*H/G, (g+1)*H/G) on entry g), uploads each slab with the halo rows of its
 * neighbours ((box-1)/2 rows, rpf.cpp:561-571), runs every pass on all slabs concurrently, and refreshes the colour
 * halo between passes with peer-to-peer copies of the neighbour's owned boundary rows (features never change, so their
 * halo travels with the upload).  Results are bit-identical to rpf_filter() on one device -- VERIFIED with two and three slab
 * contexts on ONE device (plain device copies for the halo); the branch taken between two different ordinals
 * (hipDeviceEnablePeerAccess + hipMemcpyPeerAsync) has not executed on hardware in this repository's development (one-GPU
 * boxes): tests/test_gpu_parity.py::test_multi_context_row_slabs_equal_one_context carries (0, 1) cases that run wherever a
 * second GPU is visible, and bench.py's `multi_inprocess` leg uses every visible device.  desc->row_begin / row_end
 * must name the whole image; a slab must own at least (box-1)/2 rows.  One caller thread per rpf_multi. */
typedef struct rpf_multi rpf_multi;
int32_t rpf_multi_create(rpf_multi **out, const int32_t *devices, int32_t n_devices);
void rpf_multi_destroy(rpf_multi *m);
const char *rpf_multi_last_error(const rpf_multi *m);
int32_t rpf_multi_device_count(const rpf_multi *m);
int32_t rpf_multi_set_option(rpf_multi *m, const char *name, int64_t value);
int32_t rpf_multi_filter(rpf_multi *m, const rpf_desc *desc, const void *planes, const float *ray_weight,
                         float *sample_rgb_out, float *pixel_rgb_out);
/* merged over the slabs: sums, maxima, the lowest offending image pixel; filter_kernel_ms = sum over passes of the
 * slowest slab's kernel time */
int32_t rpf_multi_query_counters(rpf_multi *m, rpf_counters *out);

/* LDS bytes per workgroup the fused kernel needs for (S, box); > device limit => RPF_E_UNSUPPORTED */
int64_t rpf_lds_bytes_required(int32_t S, int32_t box);

#ifdef __cplusplus
}
#endif
#endif
