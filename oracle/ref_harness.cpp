/*
 * ref_harness.cpp -- extern "C" doorway onto the REAL reference code that compiles here without any
 * stand-in: /root/reference/src/custom/mi.cpp (MutualInformation + histograms, libstdc++ only) and the
 * header-only templates of /root/reference/src/custom/ops.h.  Built by oracle/Makefile into
 * oracle/_ref/libref_mi.so straight from the sources where they lie (nothing is copied into this repo;
 * this file only #includes them by path, so it builds in the build container only).
 *
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/rpf_oracle.c and to generate tests/golden/*.
 *
 * The rest of the path (rpf.cpp, sd.h, sample_film.cpp) includes pbrt.h -> <glog/logging.h> and
 * visualization/vis.h -> <ImfRgbaFile.h>; both third-party trees are absent (empty submodule dirs), so
 * those files are unbuildable here without stand-ins and are NOT part of this build.
 */
#include <array>
#include <vector>

#include "custom/mi.h"  /* /root/reference/src/custom/mi.h  */
#include "custom/ops.h" /* /root/reference/src/custom/ops.h */

template <size_t N>
static void mean_std(const double *rows, int n, double *mean, double *sd) {
    std::vector<std::array<double, N>> v(n);
    for (int i = 0; i < n; ++i)
        for (size_t c = 0; c < N; ++c) v[i][c] = rows[(size_t)i * N + c];
    std::array<double, N> m = pbrt::getMean(v);      /* ops.h:111 */
    std::array<double, N> s = pbrt::getStdDev(v, m); /* ops.h:128 */
    for (size_t c = 0; c < N; ++c) { mean[c] = m[c]; sd[c] = s[c]; }
}
/* sumArray(multiplyArrays(squareArray(subtractArrays(a,b)), w)) as rpf.cpp:654-670 composes it */
template <size_t N>
static double wsq(const double *a, const double *b, const double *w) {
    std::array<double, N> x, y, ww;
    for (size_t i = 0; i < N; ++i) { x[i] = a[i]; y[i] = b[i]; ww[i] = w ? w[i] : 1.0; }
    auto sq = pbrt::squareArray(pbrt::subtractArrays(x, y));
    return w ? pbrt::sumArray(pbrt::multiplyArrays(sq, ww)) : pbrt::sumArray(sq);
}
extern "C" {

/* mi.cpp:45 */
double ref_mutual_information(const double *x, const double *y, int n) {
    std::vector<double> a(x, x + n), b(y, y + n);
    return MutualInformation(a, b);
}

/* mi.cpp:5 */
void ref_histogram(const double *x, int n, int bins, double lo, double hi, int *out) {
    std::vector<double> a(x, x + n);
    std::vector<int> h = computeHistogram(a, bins, lo, hi);
    for (int i = 0; i < bins; ++i) out[i] = h[i];
}

/* mi.cpp:23 */
void ref_joint_histogram(const double *x, const double *y, int n, int bx, int by, double lox, double hix,
                         double loy, double hiy, int *out) {
    std::vector<double> a(x, x + n), b(y, y + n);
    std::vector<std::vector<int>> j = computeJointHistogram(a, b, bx, by, lox, hix, loy, hiy);
    for (int i = 0; i < bx; ++i)
        for (int k = 0; k < by; ++k) out[i * by + k] = j[i][k];
}

void ref_mean_std_12(const double *rows, int n, double *mean, double *sd) { mean_std<12>(rows, n, mean, sd); }
void ref_mean_std_19(const double *rows, int n, double *mean, double *sd) { mean_std<19>(rows, n, mean, sd); }

/* the 3-sigma test as rpf.cpp:577-580 composes it from ops.h */
int ref_within_3std_12(const double *f, const double *mean, const double *sd) {
    std::array<double, 12> a, m, s;
    for (int i = 0; i < 12; ++i) { a[i] = f[i]; m[i] = mean[i]; s[i] = sd[i]; }
    return pbrt::allLessThan(pbrt::absArray(pbrt::subtractArrays(a, m)), pbrt::multiplyArray(s, 3)) ? 1 : 0;
}

/* SampleData::normalized as sd.h:229-232 composes it from ops.h */
void ref_normalize_19(const double *x, const double *mean, const double *sd, double *out) {
    std::array<double, 19> a, m, s;
    for (int i = 0; i < 19; ++i) { a[i] = x[i]; m[i] = mean[i]; s[i] = sd[i]; }
    std::array<double, 19> r = pbrt::divideArrays(pbrt::subtractArrays(a, m), s);
    for (int i = 0; i < 19; ++i) out[i] = r[i];
}

double ref_weighted_sqdist_2(const double *a, const double *b) { return wsq<2>(a, b, nullptr); }
double ref_weighted_sqdist_3(const double *a, const double *b, const double *w) { return wsq<3>(a, b, w); }
double ref_weighted_sqdist_12(const double *a, const double *b, const double *w) { return wsq<12>(a, b, w); }

} /* extern "C" */
