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
#include <cstring>

#include "dsq_exp_table.h"
#include "dsq_log_table.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DSQ_HD __host__ __device__ __forceinline__
#define DSQ_D __device__ __forceinline__
#else
#define DSQ_HD inline
#define DSQ_D inline
#endif

// Developer aid (tools/phase_probe): per-wave cycle accounting of the phases of a per-gene routine.
// Compiled out (empty macro) in the product.
#if defined(DSQ_PHASE_TIMING) && defined(__HIPCC__)
namespace dsq {
constexpr int kPhases = 12;
__shared__ long long g_ph_acc[4][kPhases];
__shared__ long long g_ph_last[4];
__shared__ int g_ph_cur[4];
__device__ __forceinline__ void phase_mark(int k) {
    __builtin_amdgcn_sched_barrier(0);  // keep the compiler from moving arithmetic across the mark
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        const long long t = clock64();
        g_ph_acc[w][g_ph_cur[w]] += t - g_ph_last[w];
        g_ph_last[w] = t;
        g_ph_cur[w] = k;
    }
    __builtin_amdgcn_sched_barrier(0);
}
}  // namespace dsq
#define DSQ_PHASE(k) ::dsq::phase_mark(k)
#else
#define DSQ_PHASE(k) ((void)0)
#endif

// a rarely taken branch that must not be hoisted above its test (an empty volatile asm is a side effect)
#if defined(__HIP_DEVICE_COMPILE__)
#define DSQ_NO_SPECULATE asm volatile("")
#else
#define DSQ_NO_SPECULATE ((void)0)
#endif

namespace dsq {

DSQ_HD double flog(double x);
DSQ_HD double flog_t(double x);
DSQ_HD double flog1p(double u);
DSQ_HD double frcp(double x);
DSQ_HD double frcp_g(double x);
DSQ_HD double frsq(double x);

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

// the same two series truncated for z >= 256 (dropped terms < 1e-15 absolute)
DSQ_HD double stirling_tail_big(double rz) {
    const double r2 = rz * rz;
    return (8.3333333333333333333e-2 - 2.7777777777777777778e-3 * r2) * rz;
}
DSQ_HD double digamma_tail_big(double rz) {
    const double r2 = rz * rz;
    return (8.3333333333333333333e-3 * r2 - 8.3333333333333333333e-2) * r2 - 0.5 * rz;
}

// lgamma(x) and digamma(x) for x > 0, sharing the upward shift to z >= 10.
// want_dg == false skips the digamma arithmetic.
// TAB: the two logarithms through the LDS table (flog_t: only in kernels that filled it)
template <bool WANT_DG, bool TAB = false>
DSQ_HD void lgamma_digamma(double x, double& lg, double& dg) {
    double z = x, prod = 1.0, num = 0.0;
    bool shifted = false;
    while (z < 10.0) {
        if (WANT_DG) num = num * z + prod;   // num/prod accumulates sum 1/(x+i)
        prod *= z;
        z += 1.0;
        shifted = true;
    }
    const double rz = frcp(z);
    const double lz = TAB ? flog_t(z) : flog(z);
    lg = (z - 0.5) * lz - z + kHalfLog2Pi + stirling_tail(rz);
    if (shifted) lg -= TAB ? flog_t(prod) : flog(prod);
    if (WANT_DG) {
        dg = lz + digamma_tail(rz);
        if (shifted) dg -= num * frcp(prod);
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

// ---------------------------------------------------------------------------------------------
// Lean fp64 log / log1p / reciprocal for the per-sample hot loops.  The library (ocml) versions
// cost 117 / 152 / 16 instructions on gfx950 (double-double internals, full IEEE edge handling);
// these use the classic argument reduction x = 2^k (1+f), s = f/(2+f) with a degree-7 minimax
// polynomial in s^2 (the published fdlibm scheme, |error| < 1 ulp) in ~40 / ~50 / 5 instructions.
// Domain: positive, finite, normal arguments (counts + 1/alpha, mu >= 0, 1 + mu*alpha) - which is
// all the hot loops feed them; anything else goes through the library functions.
DSQ_HD double frcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// 1/sqrt(x): v_rsq_f64 + two Newton steps (<= 1 ulp), positive normal x; NaN for negative x as sqrt gives
DSQ_HD double frsq(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(x);
    r = fma(r * fma(-x * r, r, 1.0), 0.5, r);
    r = fma(r * fma(-x * r, r, 1.0), 0.5, r);
    return r;
#else
    return 1.0 / sqrt(x);
#endif
}

// 1/x for the per-sample epilogues: as frcp, but 1/0 = inf and 1/inf = 0 survive the Newton steps (which would
// turn them into NaN), as an IEEE division gives them
DSQ_HD double frcp_g(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r0 = __builtin_amdgcn_rcp(x);
    double r = fma(fma(-x, r0, 1.0), r0, r0);
    r = fma(fma(-x, r, 1.0), r, r);
    return (r != r) ? r0 : r;
#else
    return 1.0 / x;
#endif
}

// a / b to <= 1 ulp without the IEEE division's ~30 dependent instructions (v_rcp_f64 + Newton + one residual
// correction): for scalar code whose latency matters (the optimisers between two evaluations), where b is a normal
// non-zero number in every regular case; b = 0 or inf give what the division gives (through frcp_g).  Host: a / b.
DSQ_HD double fdiv(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = frcp_g(b);
    const double q = a * r;
    const double q2 = fma(fma(-b, q, a), r, q);
    return (q2 != q2) ? q : q2;
#else
    return a / b;
#endif
}

namespace detail {
constexpr double kLn2Hi = 6.93147180369123816490e-01, kLn2Lo = 1.90821492927058770002e-10;
constexpr double kLg1 = 6.666666666666735130e-01, kLg2 = 3.999999999940941908e-01,
                 kLg3 = 2.857142874366239149e-01, kLg4 = 2.222219843214978396e-01,
                 kLg5 = 1.818357216161805012e-01, kLg6 = 1.531383769920937332e-01,
                 kLg7 = 1.479819860511658591e-01;
// log(1+f) core for 1+f in [sqrt(1/2), sqrt(2)): returns R such that log(1+f) = f - (hfsq - s*(hfsq+R))
DSQ_HD double log_poly(double f, double& s_out, double& hfsq) {
    const double s = f * frcp(2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (kLg2 + w * (kLg4 + w * kLg6));
    const double t2 = z * (kLg1 + w * (kLg3 + w * (kLg5 + w * kLg7)));
    s_out = s;
    hfsq = 0.5 * f * f;
    return t2 + t1;
}
}  // namespace detail

// log(x), x > 0 finite normal
DSQ_HD double flog(double x) {
    int k;
    double m = frexp(x, &k);  // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }
    const double f = m - 1.0;
    double s, hfsq;
    const double R = detail::log_poly(f, s, hfsq);
    const double dk = (double)k;
    return dk * detail::kLn2Hi - ((hfsq - (s * (hfsq + R) + dk * detail::kLn2Lo)) - f);
}

// log(1 + u), u >= 0 finite (accurate for tiny u: the rounding of 1+u is corrected)
DSQ_HD double flog1p(double u) {
    // no special case for tiny u: then w - 1 is u rounded to the 2^-52 grid, the polynomial returns
    // log(w) = f - f^2/2 + ... and c restores exactly what the rounding of 1 + u dropped
    const double w = 1.0 + u;
    int k;
    double m = frexp(w, &k);
    if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }
    // correction for the rounding error of w: c = (u - (w - 1)) / w   (w >= 1 here)
    const double c = (u - (w - 1.0)) * frcp(w);
    const double f = m - 1.0;
    double s, hfsq;
    const double R = detail::log_poly(f, s, hfsq);
    const double dk = (double)k;
    return dk * detail::kLn2Hi - ((hfsq - (s * (hfsq + R) + (dk * detail::kLn2Lo + c))) - f);
}

// ---------------------------------------------------------------------------------------------------------------
// Table-driven logarithm for the per-sample loops of the dispersion and IRLS kernels (the two logs there were a third
// of the loops' instructions).  x = 2^k m, m in [1, 2): entry j = top 7 mantissa bits holds rc = double(1 / c_j),
// c_j = 1 + j/128, and T = -log(rc);  r = fma(m, rc, -1) is exact to 2^-60 and lies in [0, 2^-7), so
//     log(x) = k ln2 + T + log1p(r),   log1p(r) = r + r^2 q(r)   (degree 8, truncation < 2^-59 relative).
// Entry 0 is (1, 0): arguments just above 1 (log1p of a tiny u) keep full RELATIVE accuracy; for x in [0.5, 1) the
// cancellation against -ln2 leaves an ABSOLUTE error of ~1e-16 (nothing here takes the log of such a number and then
// relies on its relative accuracy).  Measured against binary128: <= 0.93 ulp (the polynomial logarithm above: < 1 ulp),
// no division, no reciprocal: 26 instead of 39 instructions, one quarter-rate instruction fewer.
// On the device the table lives in LDS (one ds_read_b128 per logarithm): a kernel that reaches flog_t / flog1p_t calls
// log_tab_fill() with all of its threads and synchronises before the first use.
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ double g_log_tab[2 * kLogTabN];
#endif
DSQ_D void log_tab_fill() {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int i = threadIdx.x; i < 2 * kLogTabN; i += blockDim.x) g_log_tab[i] = kLogTab[i];
#endif
}
namespace detail {
DSQ_HD void log_split(double w, int& k, double& rc, double& T, double& m) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int hi = __double2hiint(w);
    k = (hi >> 20) - 1023;
    const int j = (hi >> 13) & 127;
    m = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, __double2loint(w));
    typedef __attribute__((address_space(3))) const double2 lds_d2;
    const double2 e = ((lds_d2*)g_log_tab)[j];
    rc = e.x;
    T = e.y;
#else
    uint64_t b;
    std::memcpy(&b, &w, 8);
    const uint32_t hi = (uint32_t)(b >> 32);
    k = (int)(hi >> 20) - 1023;
    const int j = (hi >> 13) & 127;
    b = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    std::memcpy(&m, &b, 8);
    rc = kLogTab[2 * j];
    T = kLogTab[2 * j + 1];
#endif
}
// log1p(r) - r,  0 <= r < 2^-7
DSQ_HD double log1p_tail(double r) {
    const double r2 = r * r, r4 = r2 * r2;
    const double a = fma(r, 1.0 / 3.0, -0.5), b = fma(r, 0.2, -0.25), c = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    const double t = fma(r2, -0.125, c);
    const double u = fma(r2, b, a);
    return r2 * fma(r4, t, u);
}
}  // namespace detail

// log(x), x > 0 finite normal
DSQ_HD double flog_t(double x) {
#if defined(DSQ_NO_LOG_TABLE)  // A/B builds (make variant): the polynomial logarithm
    return flog(x);
#endif
    int k;
    double rc, T, m;
    detail::log_split(x, k, rc, T, m);
    const double r = fma(m, rc, -1.0);
    const double p = detail::log1p_tail(r);
    const double dk = (double)k;
    return fma(dk, detail::kLn2Hi, T + (r + (p + dk * detail::kLn2Lo)));
}

// ---------------------------------------------------------------------------------------------------------------
// Table-driven exponential for the per-sample loops (mu = sf exp(x . beta): one per sample and IRLS sweep).  The library
// exp costs ~40 instructions; here x = k ln2/128 + r with |r| <= ln2/256, exp(x) = 2^(k >> 7) * 2^((k & 127)/128) * e^r:
// the middle factor from a 128-entry table (correctly rounded), e^r - 1 by a degree-5 polynomial (truncation < 6e-19
// relative), the power of two applied as two exact scale factors so that overflow gives inf and underflow is gradual.
// Measured against math.exp over [-745, 710]: <= 1 ulp (tests/test_hostsim.py).  ~22 instructions, one LDS read.
// On the device the table lives in LDS: a kernel that reaches fexp_t calls exp_tab_fill() with all of its threads and
// synchronises before the first use.
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ double g_exp_tab[kExpTabN];
#endif
DSQ_D void exp_tab_fill() {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int i = threadIdx.x; i < kExpTabN; i += blockDim.x) g_exp_tab[i] = kExpTab[i];
#endif
}
DSQ_HD double fexp_t(double x) {
    // beyond +-800 the result is inf / 0 anyway: clamping keeps the integer arithmetic in range (NaN passes through)
    const double xc = x > 800.0 ? 800.0 : (x < -800.0 ? -800.0 : x);
    const double kd = rint(xc * kExpInvStep);
    double r = fma(-kd, kExpStepHi, xc);
    r = fma(-kd, kExpStepLo, r);
    const int ki = (xc == xc) ? (int)kd : 0;
    const int j = ki & (kExpTabN - 1), m = ki >> 7;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const double lds_d;
    const double t = ((lds_d*)g_exp_tab)[j];
#else
    const double t = kExpTab[j];
#endif
    double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = fma(r, q, 1.0 / 6.0);
    q = fma(r, q, 0.5);
    const double p = fma(r * r, q, r);
    const double e = fma(t, p, t);
    const int m1 = m >> 1, m2 = m - m1;  // |m| <= 1155: both scale factors are normal numbers
#if defined(__HIP_DEVICE_COMPILE__)
    const double s1 = __hiloint2double((1023 + m1) << 20, 0), s2 = __hiloint2double((1023 + m2) << 20, 0);
#else
    const uint64_t b1 = (uint64_t)(1023 + m1) << 52, b2 = (uint64_t)(1023 + m2) << 52;
    double s1, s2;
    std::memcpy(&s1, &b1, 8);
    std::memcpy(&s2, &b2, 8);
#endif
    return (e * s1) * s2;
}

// log(1 + u), u >= 0 finite; rw = 1 / (1 + u) (the callers have it)
DSQ_HD double flog1p_t(double u, double rw) {
#if defined(DSQ_NO_LOG_TABLE)
    return flog1p(u);
#endif
    const double w = 1.0 + u;
    const double c = (u - (w - 1.0)) * rw;  // restores what the rounding of 1 + u dropped
    int k;
    double rc, T, m;
    detail::log_split(w, k, rc, T, m);
    const double r = fma(m, rc, -1.0);
    const double p = detail::log1p_tail(r);
    const double dk = (double)k;
    return fma(dk, detail::kLn2Hi, T + (r + (p + fma(dk, detail::kLn2Lo, c))));
}

}  // namespace dsq
