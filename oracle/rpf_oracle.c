/*
 * rpf_oracle.c -- CPU restatement (fp64) of the reference RPF filter pass.  TEST INFRASTRUCTURE ONLY
 * (see rpf_oracle.h).  Every function cites the reference lines it follows; statement order and
 * floating-point operation order are kept (no reassociation: build with -ffp-contract=off).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include "rpf_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Column groups, sd.h:62-94,149-214: [0,2) pFilm | [2,5) colour | [5,5+nR) random parameters | [5+nR,5+nR+nF)
 * features.  The reference has nR = 2 (pLens), nF = 12; the layout is a run-time parameter here so that the same
 * statements also check BASELINE configs[4]'s 27-dim buffers (nR = 4, nF = 18).  With nR = 2, nF = 12 every loop below
 * is the reference's. */
enum { C_P0 = 0, C_C0 = 2, C_R0 = 5 };
typedef struct { int nr, nf, nd, f0, npair; } dims_t;
static dims_t dims_of(int nr, int nf) {
    dims_t d;
    d.nr = nr > 0 ? nr : 2;
    d.nf = nf > 0 ? nf : 12;
    d.nd = 5 + d.nr + d.nf;
    d.f0 = C_R0 + d.nr;
    d.npair = d.nf * (d.nr + 2) + 3 * (d.nr + 2 + d.nf);
    return d;
}
static dims_t dims_desc(const rpf_oracle_desc *d) { return dims_of(d->n_random, d->n_feat); }

/* ------------------------------------------------------------------------------------------------
 * pair order = call order of MutualInformation in ComputeCFWeights (rpf.cpp:416-442)
 * ---------------------------------------------------------------------------------------------- */
void rpf_oracle_pair_table_ex(int32_t n_random, int32_t n_feat, int32_t *a, int32_t *b) {
    const dims_t D = dims_of(n_random, n_feat);
    int p = 0;
    for (int i = 0; i < D.nf; ++i) {
        for (int l = 0; l < D.nr; ++l) { a[p] = D.f0 + i; b[p] = C_R0 + l; ++p; } /* rpf.cpp:418-422 */
        for (int l = 0; l < 2; ++l) { a[p] = D.f0 + i; b[p] = C_P0 + l; ++p; }    /* rpf.cpp:424-426 */
    }
    for (int c = 0; c < 3; ++c) {
        for (int l = 0; l < D.nr; ++l) { a[p] = C_C0 + c; b[p] = C_R0 + l; ++p; } /* rpf.cpp:431-433 */
        for (int l = 0; l < 2; ++l) { a[p] = C_C0 + c; b[p] = C_P0 + l; ++p; }    /* rpf.cpp:435-437 */
        for (int j = 0; j < D.nf; ++j) { a[p] = C_C0 + c; b[p] = D.f0 + j; ++p; } /* rpf.cpp:439-441 */
    }
}
void rpf_oracle_pair_table(int32_t a[RPF_O_NPAIR], int32_t b[RPF_O_NPAIR]) { rpf_oracle_pair_table_ex(2, 12, a, b); }

/* x86-64 cvttsd2si semantics of static_cast<int>(double) for NaN / out-of-range (mi.cpp:14,29,35) */
static int to_int_x86(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

/* mi.cpp:14-16 / 29-31 / 35-37 */
static int bin_of(double v, double lo, double hi, int bins) {
    int bin = to_int_x86((v - lo) / (hi - lo) * bins);
    if (bins - 1 < bin) bin = bins - 1; /* std::min(bin, bins-1) */
    if (bin < 0) bin = 0;               /* std::max(bin, 0)      */
    return bin;
}

/* std::min_element / std::max_element (first extremum, operator< only) mi.cpp:47-50 */
static void min_max(const double *v, int n, double *lo, double *hi) {
    double mn = v[0], mx = v[0];
    for (int i = 1; i < n; ++i) {
        if (v[i] < mn) mn = v[i];
        if (mx < v[i]) mx = v[i];
    }
    *lo = mn;
    *hi = mx;
}

/* T[k] = round(k ln k * 2^44), k = 0..nmax: the table of the EPS residue contract below (the product computes the
 * same table on the host, csrc/rpf_api.hip ensure_tables) */
static int64_t *tfix_table(int nmax) {
    int64_t *t = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nmax + 1));
    t[0] = 0;
    for (int k = 1; k <= nmax; ++k) t[k] = (int64_t)llroundl(ldexpl((long double)k * logl((long double)k), 44));
    return t;
}

/* mi.cpp:45-90 with the default bins (= -1).
 * tfix == NULL: the reference, statement by statement (REF_ABORT policy, rpf_oracle_mi, every pin).
 * tfix != NULL (EPS policy only -- part of that DOCUMENTED DEVIATION, not of the reference): the EPS residue
 * contract.  N*MI = T[N] + sum T[J_ij] - sum T[hx_i] - sum T[hy_j] with T[k] = k ln k; for a table whose cells are
 * exactly independent that expression is 0, and what mi.cpp returns for it is pure rounding residue (+-1e-16, or an
 * exact 0 when N is a power of two).  Divided by (residue + eps) in rpf.cpp:464-470 such residue turns into an
 * arbitrary weight, so under EPS it is DEFINED away: when the integer sum of the 2^-44 fixed-point table lies
 * inside the table's own rounding band, (B*B + 2B + 1)/2 + 1 units, MI is exactly 0.  Both sides (this oracle and the
 * HIP kernel) apply the same integer rule, so the EPS weights agree to rounding. */
static double mi_scratch(const double *x, const double *y, int n, int *hx, int *hy, int *joint, const int64_t *tfix) {
    double minX, maxX, minY, maxY;
    min_max(x, n, &minX, &maxX);
    min_max(y, n, &minY, &maxY);
    int bins = (int)sqrt((double)n); /* mi.cpp:54,57 */
    if (bins < 1) bins = 1;
    memset(hx, 0, sizeof(int) * bins);
    memset(hy, 0, sizeof(int) * bins);
    memset(joint, 0, sizeof(int) * bins * bins);
    /* computeHistogram mi.cpp:5-20 */
    if (maxX == minX) hx[0] = n; else for (int i = 0; i < n; ++i) hx[bin_of(x[i], minX, maxX, bins)]++;
    if (maxY == minY) hy[0] = n; else for (int i = 0; i < n; ++i) hy[bin_of(y[i], minY, maxY, bins)]++;
    /* computeJointHistogram mi.cpp:23-42 */
    for (int i = 0; i < n; ++i) {
        int bx = 0, by = 0;
        if (maxX != minX) bx = bin_of(x[i], minX, maxX, bins);
        if (maxY != minY) by = bin_of(y[i], minY, maxY, bins);
        joint[bx * bins + by]++;
    }
    if (tfix) {
        int64_t f = tfix[n];
        for (int i = 0; i < bins; ++i) f -= tfix[hx[i]] + tfix[hy[i]];
        for (int i = 0; i < bins * bins; ++i) f += tfix[joint[i]];
        const int64_t band = ((int64_t)bins * bins + 2 * bins + 1) / 2 + 1;
        if (f <= band && f >= -band) return 0.0;
    }
    double total = (double)n; /* mi.cpp:66 */
    double mi = 0.0;
    for (int i = 0; i < bins; ++i) {
        double pX = hx[i] / total; /* mi.cpp:70-72 */
        for (int j = 0; j < bins; ++j) {
            double pY = hy[j] / total;
            double pXY = joint[i * bins + j] / total; /* mi.cpp:81 */
            double pp = pX * pY;
            if (pXY > 0 && pp != 0) mi += pXY * log(pXY / pp); /* mi.cpp:83-85 */
        }
    }
    return mi;
}

double rpf_oracle_mi(const double *x, const double *y, int32_t n) {
    int bins = (int)sqrt((double)n);
    if (bins < 1) bins = 1;
    int *buf = (int *)malloc(sizeof(int) * (size_t)(2 * bins + bins * bins));
    double r = mi_scratch(x, y, n, buf, buf + bins, buf + 2 * bins, NULL);
    free(buf);
    return r;
}

/* ops.h:111-144 : getMean then getStdDev, sequential sums */
void rpf_oracle_mean_std(const double *rows, int32_t n, int32_t ncols, double *mean, double *stddev) {
    for (int c = 0; c < ncols; ++c) { mean[c] = 0; stddev[c] = 0; }
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < ncols; ++c) mean[c] = mean[c] + rows[(size_t)i * ncols + c]; /* ops.h:121 */
    for (int c = 0; c < ncols; ++c) mean[c] = mean[c] / (double)n;                         /* ops.h:123 */
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < ncols; ++c) {
            double v = rows[(size_t)i * ncols + c];
            stddev[c] = stddev[c] + v * v; /* ops.h:138 */
        }
    for (int c = 0; c < ncols; ++c) stddev[c] = sqrt(stddev[c] / (double)n - mean[c] * mean[c]); /* ops.h:141 */
}

static size_t plane_stride(const rpf_oracle_desc *d) { return (size_t)d->H * d->W * d->S; }
static size_t sample_off(const rpf_oracle_desc *d, int y, int x, int s) {
    return ((size_t)y * d->W + x) * d->S + s;
}

static void clamp_var_eps(double *sd, int n, int policy) {
    /* EPS policy: a variance that rounds below zero is treated as zero (sqrt(-tiny)=NaN otherwise) */
    if (policy != RPF_O_DEGEN_EPS) return;
    for (int c = 0; c < n; ++c)
        if (isnan(sd[c])) sd[c] = 0.0;
}

/* rpf.cpp:338-347 for one pixel */
static void pixel_feature_stats(const rpf_oracle_desc *d, const float *planes, int y, int x, double *m12,
                                double *sd12) {
    const dims_t D = dims_desc(d);
    const size_t ps = plane_stride(d);
    double *r = (double *)malloc(sizeof(double) * (size_t)d->S * D.nf);
    for (int s = 0; s < d->S; ++s)
        for (int k = 0; k < D.nf; ++k) r[s * D.nf + k] = planes[(D.f0 + k) * ps + sample_off(d, y, x, s)];
    rpf_oracle_mean_std(r, d->S, D.nf, m12, sd12);
    clamp_var_eps(sd12, D.nf, d->degenerate_policy);
    free(r);
}

void rpf_oracle_pixel_stats(const rpf_oracle_desc *d, const float *planes, double *mean, double *stddev) {
    const int nf = dims_desc(d).nf;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < d->H; ++y)
        for (int x = 0; x < d->W; ++x) {
            size_t p = (size_t)y * d->W + x;
            pixel_feature_stats(d, planes, y, x, mean + p * nf, stddev + p * nf);
        }
}

/* rpf.cpp:356-488 on a normalised neighbourhood (z: n x 19, row major) */
typedef struct {
    double *col[RPF_O_MAXDIM]; /* the column vectors of length n (rpf.cpp:381-412) */
    int *hx, *hy, *joint;
    const int64_t *tfix;     /* EPS residue contract table (mi_scratch), NULL under REF_ABORT */
} cf_scratch;

static void cf_weights_core(const double *z, int n, dims_t D, int beta_map, int policy, double eps, cf_scratch *sc,
                            double alpha[3], double *beta, double *wrc, double *mi_out) {
    const int nd = D.nd, nf = D.nf, nr = D.nr, F0 = D.f0;
    for (int c = 0; c < nd; ++c)
        for (int i = 0; i < n; ++i) sc->col[c][i] = z[(size_t)i * nd + c];

    double D_r_fk[RPF_O_MAXDIM], D_p_fk[RPF_O_MAXDIM], D_r_ck[3], D_p_ck[3], D_f_ck[3]; /* rpf.cpp:363-377 */
    double D_cf_k[RPF_O_MAXDIM];                                    /* sum_c MI(c_c,f_k): PAPER numerator */
    for (int i = 0; i < nf; ++i) { D_r_fk[i] = 0; D_p_fk[i] = 0; D_cf_k[i] = 0; }
    for (int i = 0; i < 3; ++i) { D_r_ck[i] = 0; D_p_ck[i] = 0; D_f_ck[i] = 0; }
    const int64_t *tf = policy == RPF_O_DEGEN_EPS ? sc->tfix : NULL;

    int p = 0;
    double v;
    for (int i = 0; i < nf; ++i) { /* rpf.cpp:416-427 */
        for (int j = 0; j < nr; ++j) {
            v = mi_scratch(sc->col[F0 + i], sc->col[C_R0 + j], n, sc->hx, sc->hy, sc->joint, tf);
            D_r_fk[i] += v;
            if (mi_out) mi_out[p] = v;
            ++p;
        }
        for (int j = 0; j < 2; ++j) {
            v = mi_scratch(sc->col[F0 + i], sc->col[C_P0 + j], n, sc->hx, sc->hy, sc->joint, tf);
            D_p_fk[i] += v;
            if (mi_out) mi_out[p] = v;
            ++p;
        }
    }
    for (int i = 0; i < 3; ++i) { /* rpf.cpp:429-442 */
        for (int j = 0; j < nr; ++j) {
            v = mi_scratch(sc->col[C_C0 + i], sc->col[C_R0 + j], n, sc->hx, sc->hy, sc->joint, tf);
            D_r_ck[i] += v;
            if (mi_out) mi_out[p] = v;
            ++p;
        }
        for (int j = 0; j < 2; ++j) {
            v = mi_scratch(sc->col[C_C0 + i], sc->col[C_P0 + j], n, sc->hx, sc->hy, sc->joint, tf);
            D_p_ck[i] += v;
            if (mi_out) mi_out[p] = v;
            ++p;
        }
        for (int j = 0; j < nf; ++j) {
            v = mi_scratch(sc->col[C_C0 + i], sc->col[F0 + j], n, sc->hx, sc->hy, sc->joint, tf);
            D_f_ck[i] += v;
            D_cf_k[j] += v;
            if (mi_out) mi_out[p] = v;
            ++p;
        }
    }

    double D_f_c = 0, D_r_c = 0, D_p_c = 0; /* rpf.cpp:449-456 */
    for (int i = 0; i < 3; ++i) { D_f_c += D_f_ck[i]; D_r_c += D_r_ck[i]; D_p_c += D_p_ck[i]; }

    const double e = (policy == RPF_O_DEGEN_EPS) ? eps : 0.0;
    double num[RPF_O_MAXDIM]; /* what rpf.cpp:464 reads as D_f_ck[i], i < nF, on a 3-array; for nF != 12
                               * the presets keep the same stack rule (k < 3: D_f_ck, a gap of zeros, then D_r_fk) */
    for (int k = 0; k < nf; ++k) {
        switch (beta_map) {
        case RPF_O_BETA_REF_GCC11_O2: num[k] = k < 3 ? D_f_ck[k] : (k < 8 ? 0.0 : D_r_fk[k - 8]); break;
        case RPF_O_BETA_PAPER: num[k] = D_cf_k[k]; break;
        default: num[k] = k < 3 ? D_f_ck[k] : (k < 4 ? 0.0 : D_r_fk[k - 4]); break;
        }
    }
    double W_c_fk[RPF_O_MAXDIM], W_r_fk[RPF_O_MAXDIM], W_r_ck[3];
    for (int i = 0; i < nf; ++i) { /* rpf.cpp:463-466 */
        W_c_fk[i] = num[i] / (D_f_c + D_r_c + D_p_c + e);
        W_r_fk[i] = D_r_fk[i] / (D_r_fk[i] + D_p_fk[i] + e);
    }
    for (int i = 0; i < 3; ++i) W_r_ck[i] = D_r_ck[i] / (D_r_ck[i] + D_p_ck[i] + e); /* rpf.cpp:469-471 */
    for (int i = 0; i < 3; ++i) alpha[i] = 1 - W_r_ck[i];                            /* rpf.cpp:474-476 */
    for (int i = 0; i < nf; ++i) beta[i] = (1 - W_r_fk[i]) * W_c_fk[i];              /* rpf.cpp:478-480 */
    double w = 0;                                                                    /* rpf.cpp:483-487 */
    for (int i = 0; i < 3; ++i) w += W_r_ck[i];
    w /= 3;
    *wrc = w;
}

static void cf_scratch_alloc(cf_scratch *sc, int nmax) {
    int bins = (int)sqrt((double)nmax) + 1;
    for (int c = 0; c < RPF_O_MAXDIM; ++c) sc->col[c] = (double *)malloc(sizeof(double) * (size_t)nmax);
    sc->hx = (int *)malloc(sizeof(int) * (size_t)(2 * bins + bins * bins));
    sc->hy = sc->hx + bins;
    sc->joint = sc->hy + bins;
    sc->tfix = NULL;
}
static void cf_scratch_free(cf_scratch *sc) {
    for (int c = 0; c < RPF_O_MAXDIM; ++c) free(sc->col[c]);
    free(sc->hx);
}

void rpf_oracle_cf_weights(const double *z, int32_t n, int32_t beta_map, int32_t policy, double eps,
                           double alpha[3], double beta[12], double *wrc, double *mi96) {
    cf_scratch sc;
    cf_scratch_alloc(&sc, n);
    int64_t *tf = policy == RPF_O_DEGEN_EPS ? tfix_table(n) : NULL;
    sc.tfix = tf;
    cf_weights_core(z, n, dims_of(2, 12), beta_map, policy, eps, &sc, alpha, beta, wrc, mi96);
    cf_scratch_free(&sc);
    free(tf);
}

/* rpf.cpp:646-660: the three weighted squared distances of a pair of NORMALISED samples (19 columns each):
 * sumArray(squareArray(subtractArrays(P_i, P_j))), sumArray(multiplyArrays(squareArray(subtractArrays(C_i, C_j)), Alpha)),
 * sumArray(multiplyArrays(squareArray(subtractArrays(F_i, F_j)), Beta)) -- ops.h:17-97, sequential sums from 0 */
static void weighted_sqdist(const double *si, const double *sj, dims_t D, const double *alpha, const double *beta,
                            double *sp_, double *sc_, double *sf_) {
    double sp = 0, scol = 0, sf = 0;
    for (int k = 0; k < 2; ++k) { double t = si[C_P0 + k] - sj[C_P0 + k]; sp += t * t; }
    for (int k = 0; k < 3; ++k) { double t = si[C_C0 + k] - sj[C_C0 + k]; scol += (t * t) * alpha[k]; }
    for (int k = 0; k < D.nf; ++k) { double t = si[D.f0 + k] - sj[D.f0 + k]; sf += (t * t) * beta[k]; }
    *sp_ = sp; *sc_ = scol; *sf_ = sf;
}
void rpf_oracle_weighted_sqdist(const double *zi, const double *zj, const double alpha[3], const double beta[12],
                                double out3[3]) {
    weighted_sqdist(zi, zj, dims_of(2, 12), alpha, beta, &out3[0], &out3[1], &out3[2]);
}

static uint32_t fnv1a_u32(uint32_t h, uint32_t v) {
    for (int i = 0; i < 4; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 16777619u; }
    return h;
}
static uint32_t fnv1a_u16(uint32_t h, uint32_t v) {
    for (int i = 0; i < 2; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 16777619u; }
    return h;
}

void rpf_oracle_filter_pass(const rpf_oracle_desc *d, const float *planes, const double *colour_in,
                            double *colour_out, rpf_oracle_debug *dbg, rpf_oracle_result *res) {
    const int W = d->W, H = d->H, S = d->S, box = d->box;
    const dims_t D = dims_desc(d);
    const int ND = D.nd, NF = D.nf, F0 = D.f0;
    const int b = (box - 1) / 2; /* rpf.cpp:561 */
    const size_t ps = plane_stride(d);
    const int nmax = box * box * S;
    const double sigma_p = (double)(box / 4); /* rpf.cpp:531: integer division */
    const double seed = d->sigma_seed;        /* rpf.cpp:533 */

    /* rows outside the filtered range pass through */
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < ps; ++i)
            colour_out[c * ps + i] = colour_in ? colour_in[c * ps + i] : (double)planes[(C_C0 + c) * ps + i];

    /* stage 1a for every pixel of the buffer (rpf.cpp:519) */
    double *pmean = (double *)malloc(sizeof(double) * (size_t)H * W * NF);
    double *pstd = (double *)malloc(sizeof(double) * (size_t)H * W * NF);
    rpf_oracle_pixel_stats(d, planes, pmean, pstd);

    int32_t first_bad = INT_MAX, max_n = 0;
    int64_t n_bad = 0, sum_n = 0;
    int64_t *tfix = d->degenerate_policy == RPF_O_DEGEN_EPS ? tfix_table(nmax) : NULL;

#ifdef _OPENMP
    if (d->n_threads > 0) omp_set_num_threads(d->n_threads);
#endif
#pragma omp parallel
    {
        double *nb = (double *)malloc(sizeof(double) * (size_t)nmax * ND);  /* raw neighbourhood */
        double *z = (double *)malloc(sizeof(double) * (size_t)nmax * ND);   /* normalised */
        double *zo = (double *)malloc(sizeof(double) * (size_t)S * ND);     /* normalised own */
        double *wm = (double *)malloc(sizeof(double) * (size_t)S * nmax);           /* weights_mat */
        uint32_t *code = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nmax);
        cf_scratch sc;
        cf_scratch_alloc(&sc, nmax);
        sc.tfix = tfix;
        int32_t t_first = INT_MAX, t_max = 0;
        int64_t t_bad = 0, t_sum = 0;

#pragma omp for schedule(dynamic, 1)
        for (int y = d->row_begin; y < d->row_end; ++y) {
            for (int x = 0; x < W; ++x) {
                const size_t pix = (size_t)y * W + x;
                const double *m12 = pmean + pix * NF;
                const double *s12 = pstd + pix * NF;
                int n = 0;
                /* own samples first, unconditionally (rpf.cpp:558) */
                for (int s = 0; s < S; ++s) {
                    size_t o = sample_off(d, y, x, s);
                    for (int c = 0; c < ND; ++c) nb[(size_t)n * ND + c] = planes[c * ps + o];
                    if (colour_in)
                        for (int c = 0; c < 3; ++c) nb[(size_t)n * ND + C_C0 + c] = colour_in[c * ps + o];
                    code[n] = (uint32_t)((b * box + b) * S + s);
                    ++n;
                }
                /* rpf.cpp:562-586: xn outer ascending, yn inner ascending */
                for (int xn = x - b; xn <= x + b; ++xn) {
                    for (int yn = y - b; yn <= y + b; ++yn) {
                        if (xn == x && yn == y) continue;
                        if (xn < 0 || xn >= W || yn < 0 || yn >= H) continue;
                        for (int s = 0; s < S; ++s) {
                            size_t o = sample_off(d, yn, xn, s);
                            int within = 1;
                            for (int k = 0; k < NF; ++k) { /* ops.h:99-107 : fail iff a >= b */
                                double a = fabs((double)planes[(F0 + k) * ps + o] - m12[k]);
                                double lim = s12[k] * 3; /* rpf.cpp:579 */
                                if (a >= lim) { within = 0; break; }
                            }
                            if (!within) continue;
                            for (int c = 0; c < ND; ++c) nb[(size_t)n * ND + c] = planes[c * ps + o];
                            if (colour_in)
                                for (int c = 0; c < 3; ++c)
                                    nb[(size_t)n * ND + C_C0 + c] = colour_in[c * ps + o];
                            code[n] = (uint32_t)(((xn - x + b) * box + (yn - y + b)) * S + s);
                            ++n;
                        }
                    }
                }
                t_sum += n;
                if (n > t_max) t_max = n;

                /* stage 2: rpf.cpp:596-612 */
                double M[RPF_O_MAXDIM], SD[RPF_O_MAXDIM];
                rpf_oracle_mean_std(nb, n, ND, M, SD);
                clamp_var_eps(SD, ND, d->degenerate_policy);
                for (int i = 0; i < n; ++i)
                    for (int c = 0; c < ND; ++c) { /* sd.h:229-232, ops.h:48 */
                        double a = nb[(size_t)i * ND + c] - M[c];
                        z[(size_t)i * ND + c] = SD[c] == 0 ? 0 : a / SD[c];
                    }
                for (int i = 0; i < S; ++i) /* own samples are entries 0..S-1 of the neighbourhood */
                    for (int c = 0; c < ND; ++c) zo[i * ND + c] = z[(size_t)i * ND + c];

                /* stage 3: rpf.cpp:615-623 */
                double alpha[3], beta[RPF_O_MAXDIM], wrc;
                double *mi_out = (dbg && dbg->mi) ? dbg->mi + pix * D.npair : NULL;
                cf_weights_core(z, n, D, d->beta_map, d->degenerate_policy, d->eps, &sc, alpha, beta, &wrc, mi_out);

                if (dbg) {
                    if (dbg->nbhd_size) dbg->nbhd_size[pix] = n;
                    if (dbg->mean) memcpy(dbg->mean + pix * ND, M, sizeof(double) * ND);
                    if (dbg->stddev) memcpy(dbg->stddev + pix * ND, SD, sizeof(double) * ND);
                    if (dbg->alpha) memcpy(dbg->alpha + pix * 3, alpha, sizeof(alpha));
                    if (dbg->beta) memcpy(dbg->beta + pix * NF, beta, sizeof(double) * NF);
                    if (dbg->wrc) dbg->wrc[pix] = wrc;
                    if (dbg->member_hash) {
                        uint32_t h = 2166136261u;
                        for (int i = 0; i < n; ++i) h = fnv1a_u32(h, code[i]);
                        dbg->member_hash[pix] = h;
                    }
                    if (dbg->bin_hash) {
                        int bins = (int)sqrt((double)n);
                        if (bins < 1) bins = 1;
                        for (int c = 0; c < ND; ++c) {
                            double lo, hi;
                            min_max(sc.col[c], n, &lo, &hi);
                            uint32_t h = 2166136261u;
                            for (int i = 0; i < n; ++i)
                                h = fnv1a_u16(h, (uint32_t)(hi == lo ? 0 : bin_of(sc.col[c][i], lo, hi, bins)));
                            dbg->bin_hash[pix * ND + c] = h;
                        }
                    }
                }

                /* stage 4a: rpf.cpp:637-678 */
                double sigma_c_squared = seed * seed / (1 - wrc) / (1 - wrc); /* rpf.cpp:662 */
                double sigma_f_squared = sigma_c_squared;
                double sigma_p_squared = sigma_p * sigma_p;
                for (int i = 0; i < S; ++i) {
                    const double *si = zo + i * ND;
                    for (int j = 0; j < n; ++j) {
                        const double *sj = z + (size_t)j * ND;
                        double sp, scol, sf;
                        weighted_sqdist(si, sj, D, alpha, beta, &sp, &scol, &sf);
                        wm[(size_t)i * n + j] = exp(-sp / (2 * sigma_p_squared)) * exp(-scol / (2 * sigma_c_squared)) *
                                                exp(-sf / (2 * sigma_f_squared)); /* rpf.cpp:667-670 */
                    }
                }
                /* stage 4b: rpf.cpp:682-717 */
                int bad = 0;
                for (int i = 0; i < S; ++i) {
                    size_t o = sample_off(d, y, x, i);
                    for (int k = 0; k < 3; ++k) {
                        double sum_w = 0, sum_w_c = 0;
                        for (int j = 0; j < n; ++j) {
                            sum_w += wm[(size_t)i * n + j];
                            sum_w_c += wm[(size_t)i * n + j] * nb[(size_t)j * ND + C_C0 + k];
                        }
                        double prime = sum_w_c / sum_w;
                        if (isnan(prime)) { /* rpf.cpp:702-705: exit(1) in the reference */
                            bad = 1;
                            if (d->degenerate_policy == RPF_O_DEGEN_EPS) prime = nb[(size_t)i * ND + C_C0 + k];
                        }
                        colour_out[k * ps + o] = prime;
                    }
                }
                if (bad) {
                    ++t_bad;
                    if ((int32_t)pix < t_first) t_first = (int32_t)pix;
                }
            }
        }
#pragma omp critical
        {
            if (t_first < first_bad) first_bad = t_first;
            if (t_max > max_n) max_n = t_max;
            n_bad += t_bad;
            sum_n += t_sum;
        }
        cf_scratch_free(&sc);
        free(nb); free(z); free(zo); free(wm); free(code);
    }
    free(pmean);
    free(pstd);
    free(tfix);
    if (res) {
        res->nonfinite_pixels = n_bad;
        res->first_bad_pixel = n_bad ? first_bad : -1;
        res->status = (n_bad && d->degenerate_policy == RPF_O_DEGEN_REF_ABORT) ? 1 : 0;
        res->sum_nbhd = sum_n;
        res->max_nbhd = max_n;
        res->reserved = 0;
    }
}

/* rpf.cpp:783-794 with the default box reconstruction filter (radius 0.5, film.h:121-161): every sample
 * lands in its own pixel with filter weight 1, so the pixel value is sum(L*rayWeight)/sum(1). */
void rpf_oracle_pixel_mean(const rpf_oracle_desc *d, const double *colour, const float *ray_weight,
                           double *pixel_rgb) {
    const size_t ps = plane_stride(d);
    for (int y = d->row_begin; y < d->row_end; ++y)
        for (int x = 0; x < d->W; ++x)
            for (int c = 0; c < 3; ++c) {
                double acc = 0;
                for (int s = 0; s < d->S; ++s) {
                    size_t o = sample_off(d, y, x, s);
                    acc += colour[c * ps + o] * (ray_weight ? (double)ray_weight[o] : 1.0);
                }
                pixel_rgb[((size_t)y * d->W + x) * 3 + c] = acc / (double)d->S;
            }
}

/* rpf.cpp:37-101 and vis.cpp:34-51 */
void rpf_oracle_feature_images(const rpf_oracle_desc *d, const float *planes, double *out) {
    static const int first_col[6] = {7, 13, 10, 16, 0, 5}; /* n0, n1, p0, p1, pFilm, pLens */
    static const int ncol[6] = {3, 3, 3, 3, 2, 2};
    const size_t ps = plane_stride(d), HW = (size_t)d->H * d->W;
    for (int im = 0; im < 6; ++im) {
        double *img = out + (size_t)im * HW * 3;
        double mx[3] = {0, 0, 0}; /* vis.cpp:38 */
        for (size_t pix = 0; pix < HW; ++pix)
            for (int c = 0; c < 3; ++c) {
                double acc = 0; /* BasicRGB sums, rpf.cpp:71-76 */
                if (c < ncol[im])
                    for (int s = 0; s < d->S; ++s) acc = acc + (double)planes[(size_t)(first_col[im] + c) * ps + pix * d->S + s];
                acc = acc / (double)d->S; /* rpf.cpp:80-85 */
                img[pix * 3 + c] = acc;
                if (mx[c] < acc) mx[c] = acc; /* std::max, vis.cpp:41-43 */
            }
        for (size_t pix = 0; pix < HW; ++pix)
            for (int c = 0; c < 3; ++c) img[pix * 3 + c] = mx[c] == 0 ? 0 : img[pix * 3 + c] / mx[c]; /* vis.h:32-37 */
    }
}
