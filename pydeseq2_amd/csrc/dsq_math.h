// dsq_math.h — scalar fp64 special functions shared by every per-gene routine.
//
// Everything here is written once and compiled twice: by hipcc for gfx950
// (device code of the product) and by g++ for tests/hostsim (a test-only host
// instantiation of the same templates; it is never linked into the package).
//
// Reference arithmetic being matched: scipy.special.gammaln / polygamma(0,.) /
// scipy.stats.norm.sf as used by pydeseq2/utils.py:163-270, 718-811.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DSQ_HD __host__ __device__ __forceinline__
#define DSQ_D __device__ __forceinline__
#else
#define DSQ_HD inline
#define DSQ_D inline
#endif

namespace dsq {

constexpr double kHalfLog2Pi = 0.91893853320467274178032973640562;
constexpr double kEps = 2.220446049250313e-16;

// Stirling tail  sum_k B_2k / (2k (2k-1) z^(2k-1)),  z >= 10  (|err| < 4e-17)
DSQ_HD double stirling_tail(double rz) {
    const double r2 = rz * rz;
    double s = 6.4102564102564102564e-3;                 // 1/156
    s = s * r2 - 1.9175269175269175269e-3;               // -691/360360
    s = s * r2 + 8.4175084175084175084e-4;               // 1/1188
    s = s * r2 - 5.9523809523809523810e-4;               // -1/1680
    s = s * r2 + 7.9365079365079365079e-4;               // 1/1260
    s = s * r2 - 2.7777777777777777778e-3;               // -1/360
    s = s * r2 + 8.3333333333333333333e-2;               // 1/12
    return s * rz;
}

// asymptotic  psi(z) - log z,  z >= 10  (|err| < 5e-17)
DSQ_HD double digamma_tail(double rz) {
    const double r2 = rz * rz;
    double s = -8.3333333333333333333e-2;                // -1/12  (z^-14)
    s = s * r2 + 2.1092796092796092796e-2;               // 691/32760
    s = s * r2 - 7.5757575757575757576e-3;               // -1/132
    s = s * r2 + 4.1666666666666666667e-3;               // 1/240
    s = s * r2 - 3.9682539682539682540e-3;               // -1/252
    s = s * r2 + 8.3333333333333333333e-3;               // 1/120
    s = s * r2 - 8.3333333333333333333e-2;               // -1/12
    return s * r2 - 0.5 * rz;
}

// lgamma(x) and digamma(x) for x > 0, sharing the upward shift to z >= 10.
// want_dg == false skips the digamma arithmetic.
template <bool WANT_DG>
DSQ_HD void lgamma_digamma(double x, double& lg, double& dg) {
    double z = x, prod = 1.0, num = 0.0;
    bool shifted = false;
    while (z < 10.0) {
        if (WANT_DG) num = num * z + prod;   // num/prod accumulates sum 1/(x+i)
        prod *= z;
        z += 1.0;
        shifted = true;
    }
    const double rz = 1.0 / z;
    const double lz = log(z);
    lg = (z - 0.5) * lz - z + kHalfLog2Pi + stirling_tail(rz);
    if (shifted) lg -= log(prod);
    if (WANT_DG) {
        dg = lz + digamma_tail(rz);
        if (shifted) dg -= num / prod;
    }
}

DSQ_HD double lgamma_pos(double x) {
    double lg, dg;
    lgamma_digamma<false>(x, lg, dg);
    return lg;
}

DSQ_HD double digamma_pos(double x) {
    double lg, dg;
    lgamma_digamma<true>(x, lg, dg);
    return dg;
}

// standard normal survival function  sf(z) = 0.5 erfc(z / sqrt 2)   (scipy.stats.norm.sf)
// scipy evaluates it through cephes ndtr/erfc, which flushes to exactly 0 once
// (z/sqrt2)^2 > MAXLOG = 709.78...; mirrored so that p-values agree in the far tail.
DSQ_HD double norm_sf(double z) {
    const double a = z * 0.70710678118654752440;
    if (a > 0.0 && a * a > 7.09782712893383996843e2) return 0.0;
    return 0.5 * erfc(a);
}

DSQ_HD double dmax(double a, double b) { return a > b ? a : b; }
DSQ_HD double dmin(double a, double b) { return a < b ? a : b; }
// numpy fmax / fmin semantics (ignore NaN when the other operand is a number)
DSQ_HD double np_fmax(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
DSQ_HD double np_fmin(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
DSQ_HD double dsign(double a) { return a > 0.0 ? 1.0 : (a < 0.0 ? -1.0 : 0.0); }

}  // namespace dsq
