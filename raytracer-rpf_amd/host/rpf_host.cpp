// rpf_host.cpp -- see rpf_host.h.  Plain C++ (no HIP): everything device-side happens behind the C ABI.
#include "rpf_host.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace rpf_host {

RPFFilter::RPFFilter(int device) {
    const int32_t st = rpf_create(&ctx_, device);
    if (st != RPF_OK) {
        err_ = std::string(rpf_status_string(st)) + ": " + (ctx_ ? rpf_last_error(ctx_) : "no HIP device");
        if (ctx_) rpf_destroy(ctx_);
        ctx_ = nullptr;
    }
}

RPFFilter::~RPFFilter() {
    if (ctx_) rpf_destroy(ctx_);
}

PlaneFilm::PlaneFilm(RPFFilter &owner, int width, int height, int spp, int x0, int y0)
    : W_(width), H_(height), S_(spp), x0_(x0), y0_(y0) {
    if (width <= 0 || height <= 0 || spp <= 0 || spp > 65535 || !owner.context()) return;
    const size_t ps = (size_t)W_ * H_ * S_;
    if (!planes_.resize(owner.context(), RPF_NDIM * ps) || !rayw_.resize(owner.context(), ps) ||
        !srgb_.resize(owner.context(), 3 * ps)) {
        planes_.release();
        return;
    }
    count_.assign((size_t)W_ * H_, 0);
}

bool PlaneFilm::AddSample(int px, int py, const SampleData &s) {
    const int x = px - x0_, y = py - y0_;
    if (x < 0 || y < 0 || x >= W_ || y >= H_) return false;
    const size_t pix = (size_t)y * W_ + x;
    const int k = count_[pix];
    if (k >= S_) return false;
    const size_t ps = (size_t)W_ * H_ * S_, o = pix * S_ + k;
    for (int d = 0; d < RPF_NDIM; ++d) planes_[(size_t)d * ps + o] = (float)s.data[d];
    rayw_[o] = s.rayWeight;
    count_[pix] = (uint16_t)(k + 1);
    return true;
}

bool PlaneFilm::complete() const {
    for (uint16_t c : count_)
        if (c != S_) return false;
    return !count_.empty();
}

int RPFFilter::FilterAndReduce(PlaneFilm &film, const std::vector<int> &boxes, std::vector<float> *pixel_rgb) {
    if (!ctx_) return RPF_E_NODEVICE;
    if (!film.ok()) { err_ = "PlaneFilm allocation failed"; return RPF_E_NOMEM; }
    if (!film.complete()) { err_ = "PlaneFilm: some pixel holds fewer than S samples"; return RPF_E_BADARG; }
    if (boxes.empty() || boxes.size() > RPF_MAX_BOXES) { err_ = "1..8 box sizes"; return RPF_E_BADARG; }
    rpf_desc d;
    std::memset(&d, 0, sizeof(d));
    d.W = film.W_; d.H = film.H_; d.S = film.S_;
    d.row_begin = 0; d.row_end = film.H_;
    d.n_box = (int32_t)boxes.size();
    for (size_t i = 0; i < boxes.size(); ++i) d.box_sizes[i] = boxes[i];
    d.beta_map = beta_map; d.degenerate_policy = degenerate_policy;
    d.eps = eps; d.sigma_seed = sigma_seed;
    if (pixel_rgb) pixel_rgb->resize((size_t)film.W_ * film.H_ * 3);
    const int32_t st = rpf_filter(ctx_, &d, film.planes_.data(), film.rayw_.data(), film.srgb_.data(),
                                  pixel_rgb ? pixel_rgb->data() : nullptr);
    rpf_query_counters(ctx_, &counters_);
    if (st != RPF_OK) err_ = std::string(rpf_status_string(st)) + ": " + rpf_last_error(ctx_);
    return st;
}

int RPFFilter::ApplyRPFFilter(SamplingFilm &film, const int /*tileSize*/, int box_size) {
    return run(film, std::vector<int>{box_size}, nullptr);
}

int RPFFilter::FilterAndReduce(SamplingFilm &film, const std::vector<int> &boxes, std::vector<float> *pixel_rgb) {
    return run(film, boxes, pixel_rgb);
}

int RPFFilter::run(SamplingFilm &film, const std::vector<int> &boxes, std::vector<float> *pixel_rgb) {
    if (!ctx_) return RPF_E_NODEVICE; // constructor failed; err_ says why.  There is no CPU fallback.
    const int W = film.getWidth(), H = film.getHeight();
    if (W <= 0 || H <= 0) { err_ = "empty SamplingFilm"; return RPF_E_BADARG; }
    const size_t S = film.samples[0][0].size();
    if (S == 0) { err_ = "pixel (0,0) has no samples"; return RPF_E_BADARG; }
    // the reference silently assumes every pixel holds the same number of samples (getMean on an empty
    // vector otherwise, ops.h:116); the ABI makes it a checked precondition
    for (int x = 0; x < W; ++x)
        for (int y = 0; y < H; ++y)
            if (film.samples[x][y].size() != S) {
                char b[128];
                std::snprintf(b, sizeof(b), "pixel (%d,%d) holds %zu samples, expected %zu", x + film.x0, y + film.y0,
                              film.samples[x][y].size(), S);
                err_ = b;
                return RPF_E_BADARG;
            }
    if (boxes.empty() || boxes.size() > RPF_MAX_BOXES) { err_ = "1..8 box sizes"; return RPF_E_BADARG; }

    // marshal AoS doubles [x][y][s][19] -> SoA fp32 planes [19][y][x][s] (values are fp32-valued: pbrt Float)
    const size_t ps = (size_t)W * H * S;
    if (!planes_.resize(ctx_, RPF_NDIM * ps) || !rayw_.resize(ctx_, ps) || !col64_.resize(ctx_, 3 * ps)) {
        err_ = "page-locked staging allocation failed";
        return RPF_E_NOMEM;
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const SampleDataSet &px = film.samples[x][y];
            const size_t base = ((size_t)y * W + x) * S;
            for (size_t s = 0; s < S; ++s) {
                for (int d = 0; d < RPF_NDIM; ++d) planes_[(size_t)d * ps + base + s] = (float)px[s].data[d];
                // colours cross the boundary as the doubles the film holds (sd.h:205): after a previous
                // ApplyRPFFilter call they are no longer fp32-valued, and rounding them would move histogram bins
                for (int c = 0; c < 3; ++c) col64_[(size_t)c * ps + base + s] = px[s].getColorI(c);
                rayw_[base + s] = px[s].rayWeight;
            }
        }

    rpf_desc d;
    std::memset(&d, 0, sizeof(d));
    d.W = W; d.H = H; d.S = (int32_t)S;
    d.row_begin = 0; d.row_end = H;
    d.n_box = (int32_t)boxes.size();
    for (size_t i = 0; i < boxes.size(); ++i) d.box_sizes[i] = boxes[i];
    d.beta_map = beta_map; d.degenerate_policy = degenerate_policy;
    d.eps = eps; d.sigma_seed = sigma_seed;
    if (pixel_rgb) pixel_rgb->resize((size_t)W * H * 3);
    const int32_t st = rpf_filter_ex(ctx_, &d, planes_.data(), col64_.data(), rayw_.data(), nullptr,
                                     pixel_rgb ? pixel_rgb->data() : nullptr, col64_.data());
    rpf_query_counters(ctx_, &counters_);
    if (st != RPF_OK) err_ = std::string(rpf_status_string(st)) + ": " + rpf_last_error(ctx_);
    if (st != RPF_OK && st != RPF_E_NONFINITE) return st;

    // write the filtered colours back into the film (rpf.cpp:715, 732); other columns untouched
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            SampleDataSet &px = film.samples[x][y];
            const size_t base = ((size_t)y * W + x) * S;
            for (size_t s = 0; s < S; ++s)
                for (int c = 0; c < 3; ++c) px[s].setColorI(c, col64_[(size_t)c * ps + base + s]);
        }
    return st;
}

std::string RPFParams::Parse(const int *boxsizes, int n_boxsizes, const char *backend, RPFParams *out) {
    RPFParams r; // defaults = the reference's hard-coded constants
    if (boxsizes != nullptr && n_boxsizes > 0) {
        if (n_boxsizes > RPF_MAX_BOXES)
            return "\"boxsizes\" holds " + std::to_string(n_boxsizes) + " values; at most " + std::to_string(RPF_MAX_BOXES) + " filter passes are supported";
        r.boxsizes.assign(boxsizes, boxsizes + n_boxsizes);
        for (int b : r.boxsizes)
            if (b < 1 || (b & 1) == 0)
                return "\"boxsizes\" value " + std::to_string(b) + " is not an odd positive window width (rpf.cpp:561 centres the window on the pixel)";
    }
    const std::string be = backend ? backend : "hip";
    if (be == "hip") r.backend = HIP;
    else if (be == "reference") r.backend = REFERENCE;
    else return "unknown \"backend\" \"" + be + "\" (expected \"hip\" or \"reference\")";
    if (out) *out = r;
    return std::string();
}

} // namespace rpf_host

extern "C" int32_t rpf_host_parse_params(const int32_t *boxsizes, int32_t n_boxsizes, const char *backend, int32_t *boxes_out,
                                         int32_t *n_out, int32_t *backend_out, char *err, int32_t err_len) {
    rpf_host::RPFParams prm;
    const std::string e = rpf_host::RPFParams::Parse(boxsizes, n_boxsizes, backend, &prm);
    if (!e.empty()) {
        if (err && err_len > 0) std::snprintf(err, err_len, "%s", e.c_str());
        return -1;
    }
    if (boxes_out) for (size_t i = 0; i < prm.boxsizes.size(); ++i) boxes_out[i] = prm.boxsizes[i];
    if (n_out) *n_out = (int32_t)prm.boxsizes.size();
    if (backend_out) *backend_out = (int32_t)prm.backend;
    return 0;
}

extern "C" int32_t rpf_host_apply_filter_aos(double *aos, const float *ray_weight, int32_t W, int32_t H, int32_t S,
                                             const int32_t *box_sizes, int32_t n_box, int32_t beta_map, int32_t policy,
                                             int32_t device, float *pixel_rgb_out, char *err, int32_t err_len) {
    using namespace rpf_host;
    if (!aos || !box_sizes || W <= 0 || H <= 0 || S <= 0) return RPF_E_BADARG;
    SamplingFilm film(W, H);
    for (int x = 0; x < W; ++x)
        for (int y = 0; y < H; ++y) {
            SampleDataSet &px = film.samples[x][y];
            px.resize(S);
            for (int s = 0; s < S; ++s) {
                const size_t o = (((size_t)x * H + y) * S + s);
                std::memcpy(px[s].data, aos + o * RPF_NDIM, sizeof(double) * RPF_NDIM);
                px[s].rayWeight = ray_weight ? ray_weight[o] : 1.0f;
            }
        }
    RPFFilter f(device);
    f.beta_map = beta_map & 0xff;
    f.degenerate_policy = policy;
    std::vector<float> pix;
    int st = RPF_OK;
    if (beta_map & RPF_HOST_PER_BOX_CALLS) {
        // the reference's own call shape (rpf.cpp:767-775): one ApplyRPFFilter per box size on the same film
        for (int i = 0; i < n_box && (st == RPF_OK); ++i) st = f.ApplyRPFFilter(film, 16, box_sizes[i]);
    } else if (n_box == 1 && !pixel_rgb_out) st = f.ApplyRPFFilter(film, 16, box_sizes[0]);
    else st = f.FilterAndReduce(film, std::vector<int>(box_sizes, box_sizes + n_box), pixel_rgb_out ? &pix : nullptr);
    if (err && err_len > 0) std::snprintf(err, err_len, "%s", f.last_error().c_str());
    if (st != RPF_OK && st != RPF_E_NONFINITE) return st;
    for (int x = 0; x < W; ++x)
        for (int y = 0; y < H; ++y)
            for (int s = 0; s < S; ++s) {
                const size_t o = (((size_t)x * H + y) * S + s);
                std::memcpy(aos + o * RPF_NDIM, film.samples[x][y][s].data, sizeof(double) * RPF_NDIM);
            }
    if (pixel_rgb_out && !pix.empty()) std::memcpy(pixel_rgb_out, pix.data(), pix.size() * sizeof(float));
    return st;
}

extern "C" int32_t rpf_host_planefilm_filter(const double *aos, const float *ray_weight, int32_t W, int32_t H, int32_t S,
                                             const int32_t *box_sizes, int32_t n_box, int32_t beta_map, int32_t policy,
                                             int32_t device, float *sample_rgb_out, float *pixel_rgb_out, char *err,
                                             int32_t err_len) {
    using namespace rpf_host;
    if (!aos || !box_sizes || W <= 0 || H <= 0 || S <= 0) return RPF_E_BADARG;
    RPFFilter f(device);
    f.beta_map = beta_map;
    f.degenerate_policy = policy;
    PlaneFilm film(f, W, H, S);
    int st = RPF_OK;
    if (!film.ok()) {
        st = f.context() ? RPF_E_NOMEM : RPF_E_NODEVICE;
    } else {
        // 16x16 producer tiles, as RPFIntegrator::FillSampleFilm's ParallelFor2D hands them out (rpf.cpp:220-299)
        const int tx = (W + 15) / 16, ty = (H + 15) / 16;
        bool all = true;
#pragma omp parallel for schedule(dynamic) reduction(&& : all)
        for (int t = 0; t < tx * ty; ++t) {
            const int x0 = (t % tx) * 16, y0 = (t / tx) * 16;
            for (int y = y0; y < std::min(y0 + 16, (int)H); ++y)
                for (int x = x0; x < std::min(x0 + 16, (int)W); ++x)
                    for (int s = 0; s < S; ++s) {
                        const size_t o = (((size_t)x * H + y) * S + s);
                        SampleData sd;
                        std::memcpy(sd.data, aos + o * RPF_NDIM, sizeof(double) * RPF_NDIM);
                        sd.rayWeight = ray_weight ? ray_weight[o] : 1.0f;
                        all = film.AddSample(x, y, sd) && all;
                    }
        }
        std::vector<float> pix;
        st = all ? f.FilterAndReduce(film, std::vector<int>(box_sizes, box_sizes + n_box), pixel_rgb_out ? &pix : nullptr)
                 : (int)RPF_E_BADARG;
        if (st == RPF_OK || st == RPF_E_NONFINITE) {
            if (sample_rgb_out) std::memcpy(sample_rgb_out, film.filtered(), (size_t)3 * W * H * S * sizeof(float));
            if (pixel_rgb_out) std::memcpy(pixel_rgb_out, pix.data(), pix.size() * sizeof(float));
        }
    }
    if (err && err_len > 0) std::snprintf(err, err_len, "%s", f.last_error().c_str());
    return st;
}
