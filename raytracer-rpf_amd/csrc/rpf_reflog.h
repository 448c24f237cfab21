// rpf_reflog.h -- log(q) for q next to 1.0, as the host's libm evaluates it.
//
// Why it exists: for an EXACTLY independent joint histogram (J_ij * N == hx_i * hy_j on every occupied cell) mi.cpp:79-86
// sums pXY * log(pXY / (pX * pY)) over quotients that are 1.0 up to the rounding of three divisions and a product.  With N
// a power of two every quotient is exactly 1 and the sum is an exact 0; otherwise some quotients are 1 +- a few ulp and
// the sum is pure rounding residue (+-1e-16), which rpf.cpp:465/470 then divide by each other.  Under RPF_DEGEN_REF_ABORT
// the device reproduces that residue term by term (filter_pixel_big_kernel, "reference expression"), which needs
// log(1 +- k ulp) to come out as glibc's does on the host.  glibc (>= 2.28, sysdeps/ieee754/dbl-64/e_log.c) handles
// 1 - 2^-4 < x < 1 + 0x1.09p-4 with one polynomial in r = x - 1 whose leading terms r - r*r/2 are formed with an exact
// hi/lo split; the statement sequence below restates that published algorithm in plain IEEE operations (no contraction:
// the kernel TUs are built with -ffp-contract=off).  For |r| < 2^-40 -- the only range the residue path ever sees -- every
// product involved is exact, so the result does not depend on whether libm's own build used fused multiply-adds, and it
// is the correctly rounded value of r - r^2/2 + r^3/3.  tests/test_oracle.py::test_reflog_matches_libm_next_to_one compiles
// this header for the host and compares it with libm's log() bit for bit on q = 1 +- k ulp, k <= 4096 (and reports the
// agreement on wider r, where it is best effort).
#pragma once
#ifndef RPF_HD
#define RPF_HD __host__ __device__ __forceinline__
#endif

namespace rpf {

// true when q is inside the interval this routine covers
RPF_HD bool reflog_in_range(double q) { return q > 1.0 - 0x1p-4 && q < 1.0 + 0x1.09p-4; }

RPF_HD double reflog_near_one(double x) {
    if (x == 1.0) return 0.0;
    const double B0 = -0x1p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3,
                 B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de3p-3, B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4,
                 B8 = -0x1.999eb43b068ffp-4, B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
    const double r = x - 1.0;
    const double r2 = r * r;
    const double r3 = r * r2;
    double y = r3 * (B1 + r * B2 + r2 * B3 + r3 * (B4 + r * B5 + r2 * B6 + r3 * (B7 + r * B8 + r2 * B9 + r3 * B10)));
    double w = r * 0x1p27;
    const double rhi = r + w - w;
    const double rlo = r - rhi;
    w = rhi * rhi * B0;
    const double hi = r + w;
    double lo = r - hi + w;
    lo += B0 * rlo * (rhi + r);
    y += lo;
    y += hi;
    return y;
}

} // namespace rpf
