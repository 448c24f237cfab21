// rpf_host.h -- host-side mirror (C++) of the reference's interface for the hot path, above the C ABI.
//
// The reference is compiled C++ with no plugin ABI: RPFIntegrator::Render (rpf.cpp:737-805) owns a
// SamplingFilm (sample_film.h:28-45: samples[x][y] = std::vector<SampleData>, SampleData = 19 doubles +
// Float rayWeight, sd.h:51-60) and calls the private member
//     void ApplyRPFFilter(SamplingFilm &samplingFilm, const int tileSize, int box_size)   (rpf.h:91-95)
// once per box size.  This header reproduces those shapes without any pbrt type so that the call site reads
// the same; the body marshals AoS doubles -> SoA fp32 planes and calls librpf_hip.so.  A pbrt maintainer
// keeps pbrt's own SampleData/SamplingFilm and only copies the body of RPFFilter::ApplyRPFFilter (see
// INTEGRATION.md).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rpf_hip.h"

namespace rpf_host {

constexpr int SD_N_FEATURES = 12, SD_N_POSITION = 2, SD_N_RANDOM = 2, SD_N_COLOR = 3; // sd.h:40-43

// same field order and meaning as pbrt::SampleData (sd.h:51-60); the getters the filter path uses
struct SampleData {
    double data[RPF_NDIM];
    float rayWeight;
    double getColorI(int i) const { return data[2 + i]; }      // sd.h:205
    void setColorI(int i, double v) { data[2 + i] = v; }       // sd.h:208
    double getFeatureI(int i) const { return data[7 + i]; }    // sd.h:155
};
typedef std::vector<SampleData> SampleDataSet;                 // sd.h:239

// same container as pbrt::SamplingFilm (sample_film.h:28-45): x-major, samples[x - x0][y - y0]
struct SamplingFilm {
    std::vector<std::vector<SampleDataSet>> samples;
    int x0 = 0, y0 = 0; // pixelBounds.pMin
    SamplingFilm(int width, int height, int x0_ = 0, int y0_ = 0)
        : samples(width, std::vector<SampleDataSet>(height)), x0(x0_), y0(y0_) {}
    int getWidth() const { return (int)samples.size(); }                       // sample_film.cpp:68
    int getHeight() const { return samples.empty() ? 0 : (int)samples[0].size(); } // sample_film.cpp:72
    void AddSample(int px, int py, const SampleData &s) { samples[px - x0][py - y0].push_back(s); } // sample_film.cpp:44
};

class RPFFilter;

// Optional scene-file parameters of the "rpf" integrator.  CreateRPFIntegrator (rpf.cpp:941-966) reads only the path
// tracer's keys and the filter's constants are hard-coded (box sizes {7}, rpf.cpp:767; the commented alternative
// {55, 35, 17, 7} beside it); these two keys make them reachable from a scene file while a file without them renders
// exactly as before ("Config / flags"):
//     "integer boxsizes" [ 7 ]      one filter pass per entry, in order; odd, >= 1, at most RPF_MAX_BOXES entries
//     "string backend"   "hip"      "hip": this library; "reference": pbrt's own CPU ApplyRPFFilter loop stays in charge
//                                   (nothing of this library runs -- there is no CPU path inside it)
// Parse() takes what pbrt's ParamSet hands out -- FindInt("boxsizes", &n) and FindOneString("backend", "hip") -- and
// returns "" or a message for pbrt's Error().
struct RPFParams {
    enum Backend { HIP = 0, REFERENCE = 1 };
    std::vector<int> boxsizes{7};
    Backend backend = HIP;
    static std::string Parse(const int *boxsizes, int n_boxsizes, const char *backend, RPFParams *out);
};

// Page-locked array handed out by the filter's context (rpf_host_alloc); grow-only, released with its owner.
template <class T>
class PinnedArray {
  public:
    PinnedArray() = default;
    ~PinnedArray() { release(); }
    PinnedArray(const PinnedArray &) = delete;
    PinnedArray &operator=(const PinnedArray &) = delete;
    bool resize(rpf_ctx *ctx, size_t n) { // contents are not preserved
        if (n <= cap_) { n_ = n; return true; }
        release();
        void *p = nullptr;
        if (rpf_host_alloc(ctx, (uint64_t)n * sizeof(T), &p) != RPF_OK) return false;
        ptr_ = static_cast<T *>(p); cap_ = n_ = n;
        return true;
    }
    void release() { if (ptr_) rpf_host_free(nullptr, ptr_); ptr_ = nullptr; cap_ = n_ = 0; }
    T *data() { return ptr_; }
    const T *data() const { return ptr_; }
    size_t size() const { return n_; }
    T &operator[](size_t i) { return ptr_[i]; }
    const T &operator[](size_t i) const { return ptr_[i]; }
  private:
    T *ptr_ = nullptr;
    size_t cap_ = 0, n_ = 0;
};

// the feature producer's film as SoA fp32 planes in page-locked memory, [19][y][x][s] -- what
// FillSampleFilm / SamplingTile::addSample / MergeSamplingTile (rpf.cpp:220-299, sample_film.cpp:44-66) would fill
// instead of the heap AoS-double SamplingFilm.  Every pixel holds exactly S samples (the filter's precondition).
// AddSample needs no mutex: pbrt's render tiles are disjoint, so a pixel is only ever written by one thread.
class PlaneFilm {
  public:
    PlaneFilm(RPFFilter &owner, int width, int height, int spp, int x0 = 0, int y0 = 0);
    bool ok() const { return planes_.size() != 0; }
    int getWidth() const { return W_; }
    int getHeight() const { return H_; }
    int samplesPerPixel() const { return S_; }
    // appends to pixel (px,py); false if the pixel is outside the film or already holds S samples
    bool AddSample(int px, int py, const SampleData &s);
    bool complete() const; // every pixel has S samples
    float *planes() { return planes_.data(); }          // [19][H][W][S]
    float *rayWeight() { return rayw_.data(); }         // [H][W][S]
    const float *filtered() const { return srgb_.data(); } // [3][H][W][S], valid after RPFFilter::FilterAndReduce
    float filteredColor(int px, int py, int s, int c) const {
        return srgb_[(size_t)c * W_ * H_ * S_ + ((size_t)(py - y0_) * W_ + (px - x0_)) * S_ + s];
    }
  private:
    friend class RPFFilter;
    int W_, H_, S_, x0_, y0_;
    PinnedArray<float> planes_, rayw_, srgb_;
    std::vector<uint16_t> count_;
};

class RPFFilter {
  public:
    explicit RPFFilter(int device = 0);
    ~RPFFilter();
    RPFFilter(const RPFFilter &) = delete;
    RPFFilter &operator=(const RPFFilter &) = delete;

    // the reference hard-codes these (rpf.cpp:533, 767); defaults reproduce it
    int beta_map = RPF_BETA_REF_GCC11_O3;
    int degenerate_policy = RPF_DEGEN_REF_ABORT;
    double eps = 1e-10, sigma_seed = 0.002;

    // Drop-in for RPFIntegrator::ApplyRPFFilter (rpf.cpp:497-733): on return the colour columns (data[2..4]) of
    // every sample hold the filtered colours, everything else is untouched (rpf.cpp:715, 732).  tileSize is
    // accepted for signature compatibility (the GPU path does not tile on the host).  Returns an rpf_status;
    // where the reference prints "PRIME ERROR" and calls exit(1) (rpf.cpp:702-705) this returns
    // RPF_E_NONFINITE and leaves the NaNs in place for the caller to report.
    int ApplyRPFFilter(SamplingFilm &samplingFilm, const int tileSize, int box_size);

    // The whole post-sampling part of Render(): every box size (rpf.cpp:767-775) and the film reduction with
    // the default box reconstruction filter (rpf.cpp:779-794): pixel_rgb[(y*W+x)*3+c], may be NULL.
    int FilterAndReduce(SamplingFilm &samplingFilm, const std::vector<int> &box_sizes, std::vector<float> *pixel_rgb);
    // Same on a PlaneFilm: no marshalling at all, the pinned planes go straight to rpf_filter()'s band pipeline;
    // filtered sample colours land in film.filtered().
    int FilterAndReduce(PlaneFilm &film, const std::vector<int> &box_sizes, std::vector<float> *pixel_rgb);
    rpf_ctx *context() { return ctx_; }

    const std::string &last_error() const { return err_; }
    const rpf_counters &counters() const { return counters_; }

  private:
    int run(SamplingFilm &film, const std::vector<int> &boxes, std::vector<float> *pixel_rgb);
    rpf_ctx *ctx_ = nullptr;
    std::string err_;
    rpf_counters counters_{};
    PinnedArray<float> planes_, rayw_; // staging for the AoS film
    PinnedArray<double> col64_;        // ... and its colours, carried as doubles in both directions (rpf_filter_ex)
};

} // namespace rpf_host

// OR-ed into the beta_map argument of rpf_host_apply_filter_aos: run the box list as the reference's Render loop does,
// one RPFFilter::ApplyRPFFilter(film, 16, box) call per box size on the same film (rpf.cpp:767-775)
#define RPF_HOST_PER_BOX_CALLS 0x100

extern "C" {
// test / FFI doorway: aos is double [W][H][S][19] in SamplingFilm order, ray_weight float [W][H][S] (may be NULL);
// colours are filtered in place (as doubles).  Returns an rpf_status.
int32_t rpf_host_apply_filter_aos(double *aos, const float *ray_weight, int32_t W, int32_t H, int32_t S,
                                  const int32_t *box_sizes, int32_t n_box, int32_t beta_map, int32_t policy,
                                  int32_t device, float *pixel_rgb_out, char *err, int32_t err_len);
// same input, but the samples are pushed into a PlaneFilm by concurrent 16x16-tile producers (as pbrt's render
// tiles would); sample_rgb_out float [3][H][W][S], pixel_rgb_out float [H][W][3], either may be NULL.
// RPFParams::Parse through a C doorway: boxes_out[RPF_MAX_BOXES]; returns 0 or -1 with the message in err
int32_t rpf_host_parse_params(const int32_t *boxsizes, int32_t n_boxsizes, const char *backend, int32_t *boxes_out,
                              int32_t *n_out, int32_t *backend_out, char *err, int32_t err_len);
int32_t rpf_host_planefilm_filter(const double *aos, const float *ray_weight, int32_t W, int32_t H, int32_t S,
                                  const int32_t *box_sizes, int32_t n_box, int32_t beta_map, int32_t policy,
                                  int32_t device, float *sample_rgb_out, float *pixel_rgb_out, char *err, int32_t err_len);
}
