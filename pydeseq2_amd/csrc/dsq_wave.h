// dsq_wave.h — "one gene per wavefront" execution policy.
//
// Every per-gene routine is a template over a Wave policy:
//   * DeviceWave : a 64-lane CDNA4 wavefront; lane l owns samples l, l+64, ...
//                  (gene-major rows => each strided access is one coalesced
//                  256 B / 512 B request); cross-lane sums are butterfly
//                  all-reduces, so every lane ends up with the bit-identical
//                  total and all control flow stays wave-uniform.
//   * HostWave   : a single "lane" that walks all samples; used only by
//                  tests/hostsim to unit-test the same code on a CPU.
#pragma once
#include "dsq_math.h"

namespace dsq {

// Neumaier-compensated accumulator.  The alpha-dependent NLL terms are O(count) each and
// cancel against an alpha-independent constant, so plain accumulation would leave
// ulp(sum of counts) noise in a loss whose line search resolves ~1e-12 differences.
struct KSum {
    double s = 0.0, c = 0.0;
    DSQ_HD void add(double x) {  // Knuth TwoSum: e = exact rounding error of s + x, no compare/select
        const double t = s + x;
        const double bp = t - s;
        c += (s - (t - bp)) + (x - bp);
        s = t;
    }
    DSQ_HD void merge(double os, double oc) {  // symmetric in (this, other)
        const double t = s + os;
        const double e = (fabs(s) >= fabs(os)) ? ((s - t) + os) : ((os - t) + s);
        c = (c + oc) + e;
        s = t;
    }
    DSQ_HD double value() const { return s + c; }
};

// Pointers whose address space the compiler cannot see (they reach an out-of-line function through a
// struct): naming it turns flat loads with 64-bit address arithmetic into ds_read / global_load with
// immediate offsets.  Identity on the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define DSQ_AS_LDS(T, p) ((const __attribute__((address_space(3))) T*)(p))
#define DSQ_AS_GLOBAL(T, p) ((const __attribute__((address_space(1))) T*)(p))
#define DSQ_LDS_STRUCT(T) __attribute__((address_space(3))) T  // a struct that lives in LDS (writable)
#else
#define DSQ_AS_LDS(T, p) (p)
#define DSQ_AS_GLOBAL(T, p) (p)
#define DSQ_LDS_STRUCT(T) T
#endif

#if defined(__HIPCC__)
// The xor butterfly of the wave reductions without the LDS crossbar (ds_bpermute): gfx950's
// v_permlane32_swap / v_permlane16_swap exchange the wave halves / neighbouring 16-lane rows, DPP
// row_ror:8, row_ror:4 and quad permutations do the steps inside a row.  Same partners, same order
// (32, 16, 8, 4, 2, 1), same operand values as `v op= shfl_xor(v, m)`: bit-identical results.
// (row_ror:4 reaches lane i +- 4 rather than i ^ 4; after the xor-8 step the values have period 8
// inside a row, so that is the same value.)  All 64 lanes must be active, as for the shuffles.
namespace detail {
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>((int)(b & 0xffffffffll)), hi = dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (a, c): a = the value of the lower half (row) replicated, c = the value of the upper half (row)
template <bool HALF>
__device__ __forceinline__ void swap_i(int v, int& a, int& c) {
    if constexpr (HALF) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        a = (int)r[0]; c = (int)r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        a = (int)r[0]; c = (int)r[1];
    }
}
template <bool HALF>
__device__ __forceinline__ void swap_d(double v, double& a, double& c) {
    const long long b = __double_as_longlong(v);
    int alo, clo, ahi, chi;
    swap_i<HALF>((int)(b & 0xffffffffll), alo, clo);
    swap_i<HALF>((int)(b >> 32), ahi, chi);
    a = __longlong_as_double(((long long)ahi << 32) | (unsigned int)alo);
    c = __longlong_as_double(((long long)chi << 32) | (unsigned int)clo);
}
// two different operands: a = [x.lower | y.lower], c = [x.upper | y.upper] (halves for HALF, else 16-lane rows: rows 0, 2
// of the result take x, rows 1, 3 take y)
template <bool HALF>
__device__ __forceinline__ void swap_xy_i(int x, int y, int& a, int& c) {
    if constexpr (HALF) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)y, false, false);
        a = (int)r[0]; c = (int)r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)y, false, false);
        a = (int)r[0]; c = (int)r[1];
    }
}
template <bool HALF>
__device__ __forceinline__ void swap_xy_d(double x, double y, double& a, double& c) {
    const long long bx = __double_as_longlong(x), by = __double_as_longlong(y);
    int alo, clo, ahi, chi;
    swap_xy_i<HALF>((int)(bx & 0xffffffffll), (int)(by & 0xffffffffll), alo, clo);
    swap_xy_i<HALF>((int)(bx >> 32), (int)(by >> 32), ahi, chi);
    a = __longlong_as_double(((long long)ahi << 32) | (unsigned int)alo);
    c = __longlong_as_double(((long long)chi << 32) | (unsigned int)clo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kRor8 = 0x128, kRor4 = 0x124, kXor2 = 0x4E, kXor1 = 0xB1;
}  // namespace detail

struct DeviceWave {
    static constexpr int W = 64;
    static __device__ __forceinline__ int lane() { return threadIdx.x & 63; }
    static __device__ __forceinline__ double sum(double v) {
        double a, c;
        detail::swap_d<true>(v, a, c); v = a + c;
        detail::swap_d<false>(v, a, c); v = a + c;
        v += detail::dpp_d<detail::kRor8>(v);
        v += detail::dpp_d<detail::kRor4>(v);
        v += detail::dpp_d<detail::kXor2>(v);
        v += detail::dpp_d<detail::kXor1>(v);
        return v;
    }
    // (shuffles: with a NaN operand `v > o ? v : o` is not symmetric between the two partners, which the
    // row_ror:4 step of the DPP butterfly relies on; max is not on any hot path)
    static __device__ __forceinline__ double max(double v) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double o = __shfl_xor(v, m, 64);
            v = v > o ? v : o;
        }
        return v;
    }
    static __device__ __forceinline__ int sumi(int v) {
        int a, c;
        detail::swap_i<true>(v, a, c); v = a + c;
        detail::swap_i<false>(v, a, c); v = a + c;
        v += detail::dpp_i<detail::kRor8>(v);
        v += detail::dpp_i<detail::kRor4>(v);
        v += detail::dpp_i<detail::kXor2>(v);
        v += detail::dpp_i<detail::kXor1>(v);
        return v;
    }
    static __device__ __forceinline__ int maxi(int v) {
        int a, c;
        detail::swap_i<true>(v, a, c); v = a > c ? a : c;
        detail::swap_i<false>(v, a, c); v = a > c ? a : c;
        { const int o = detail::dpp_i<detail::kRor8>(v); v = v > o ? v : o; }
        { const int o = detail::dpp_i<detail::kRor4>(v); v = v > o ? v : o; }
        { const int o = detail::dpp_i<detail::kXor2>(v); v = v > o ? v : o; }
        { const int o = detail::dpp_i<detail::kXor1>(v); v = v > o ? v : o; }
        return v;
    }
    // exclusive prefix sum over the lanes
    static __device__ __forceinline__ int excl_scan_i(int v) {
        int s = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(s, d, 64);
            if ((int)(threadIdx.x & 63) >= d) s += o;
        }
        return s - v;
    }
    // wave-private LDS histogram cell += 1; the segment is touched by this wave only
    static __device__ __forceinline__ void hist_add(unsigned int* cell) { atomicAdd(cell, 1u); }
    // ... returning the previous value (an append position; lanes of one instruction get consecutive values in lane
    // order, so a single wave fills its list deterministically)
    static __device__ __forceinline__ unsigned int slot_add(unsigned int* cell) { return atomicAdd(cell, 1u); }
    // wave-private LDS accumulator cell += v (several lanes may name the same cell: ds_add_f64; the segment is
    // touched by this wave only and the LDS serialises same-address lanes in a fixed order => deterministic)
    static __device__ __forceinline__ void cell_add(double* cell, double v) {
        typedef __attribute__((address_space(3))) double lds_double;  // always an LDS address: ds_add_f64, not flat
        __hip_atomic_fetch_add((lds_double*)cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    static __device__ __forceinline__ bool any(bool p) { return __any(p); }
    // value held by lane `src` (all lanes must be active); src in [0, 64)
    static __device__ __forceinline__ double from_lane(double v, int src) { return __shfl(v, src, 64); }
    // a value known to be identical in every lane -> scalar registers (frees VGPRs)
    static __device__ __forceinline__ double uniform(double v) {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
        const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    static __device__ __forceinline__ double sum_comp(KSum k) {
        double a, c;
        const bool up32 = (threadIdx.x & 32) != 0, up16 = (threadIdx.x & 16) != 0;
        {
            detail::swap_d<true>(k.s, a, c); const double os = up32 ? a : c;
            detail::swap_d<true>(k.c, a, c); const double oc = up32 ? a : c;
            k.merge(os, oc);
        }
        {
            detail::swap_d<false>(k.s, a, c); const double os = up16 ? a : c;
            detail::swap_d<false>(k.c, a, c); const double oc = up16 ? a : c;
            k.merge(os, oc);
        }
        k.merge(detail::dpp_d<detail::kRor8>(k.s), detail::dpp_d<detail::kRor8>(k.c));
        k.merge(detail::dpp_d<detail::kRor4>(k.s), detail::dpp_d<detail::kRor4>(k.c));
        k.merge(detail::dpp_d<detail::kXor2>(k.s), detail::dpp_d<detail::kXor2>(k.c));
        k.merge(detail::dpp_d<detail::kXor1>(k.s), detail::dpp_d<detail::kXor1>(k.c));
        return k.value();
    }
    // K sums at once.  The butterfly above, applied per value, moves every value through all six stages in all 64
    // lanes (18 instructions each).  Here the two exchange stages pair the values up - after the half-wave swap a
    // lane keeps one of two values, after the row swap one of four - so the four in-row stages run on K/4 registers,
    // and the totals (one value per 16-lane row) are broadcast with v_readlane.  Same partners and the same order of
    // additions for every value: bit-identical to sum(), at about 40 % of its instructions for K >= 8.
    template <int K>
    static __device__ __forceinline__ void sum_n(double (&v)[K]) {
        if constexpr (K < 3) {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = sum(v[k]);
        } else {
            constexpr int Q = (K + 3) / 4;
            double u[2 * Q], w[Q];
#pragma unroll
            for (int i = 0; i < 2 * Q; ++i) {
                const double x = i < K ? v[i] : 0.0;
                const double y = i + 2 * Q < K ? v[i + 2 * Q] : 0.0;
                double a, c;
                detail::swap_xy_d<true>(x, y, a, c);
                u[i] = a + c;
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                double a, c;
                detail::swap_xy_d<false>(u[i], u[i + Q], a, c);
                double t = a + c;
                t += detail::dpp_d<detail::kRor8>(t);
                t += detail::dpp_d<detail::kRor4>(t);
                t += detail::dpp_d<detail::kXor2>(t);
                t += detail::dpp_d<detail::kXor1>(t);
                w[i] = t;
            }
            // row r of w[i] holds the total of value i + r * Q
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = detail::readlane_d(w[k % Q], 16 * (k / Q));
        }
    }
};

// Sixteen lanes per gene, four genes per wavefront.  For designs whose kernels spend their time in the p x p algebra
// between short sample loops (p >= 5 with design cells: dsq_irls.h, irls_sweep_cell) the wave-redundant algebra of
// DeviceWave repeats ONE gene's Cholesky / solve in 64 lanes; here the four 16-lane rows of a wavefront work on four
// genes, so the same instructions serve four of them.  Reductions stay inside a row (DPP row_ror / quad permutations:
// the last four stages of the butterfly above), "uniform" values are uniform per row only (no scalar registers), and
// rows may diverge (different sweep counts): every cross-lane operation used by the per-gene templates is row-scoped,
// and a row is always active or inactive as a whole.
struct RowWave {
    static constexpr int W = 16;
    static __device__ __forceinline__ int lane() { return threadIdx.x & 15; }
    static __device__ __forceinline__ double sum(double v) {
        v += detail::dpp_d<detail::kRor8>(v);
        v += detail::dpp_d<detail::kRor4>(v);
        v += detail::dpp_d<detail::kXor2>(v);
        v += detail::dpp_d<detail::kXor1>(v);
        return v;
    }
    static __device__ __forceinline__ double max(double v) {
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            const double o = __shfl_xor(v, m, 16);
            v = v > o ? v : o;
        }
        return v;
    }
    static __device__ __forceinline__ int sumi(int v) {
        v += detail::dpp_i<detail::kRor8>(v);
        v += detail::dpp_i<detail::kRor4>(v);
        v += detail::dpp_i<detail::kXor2>(v);
        v += detail::dpp_i<detail::kXor1>(v);
        return v;
    }
    static __device__ __forceinline__ int maxi(int v) {
        { const int o = detail::dpp_i<detail::kRor8>(v); v = v > o ? v : o; }
        { const int o = detail::dpp_i<detail::kRor4>(v); v = v > o ? v : o; }
        { const int o = detail::dpp_i<detail::kXor2>(v); v = v > o ? v : o; }
        { const int o = detail::dpp_i<detail::kXor1>(v); v = v > o ? v : o; }
        return v;
    }
    static __device__ __forceinline__ int excl_scan_i(int v) {
        int s = v;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int o = __shfl_up(s, d, 16);
            if ((int)(threadIdx.x & 15) >= d) s += o;
        }
        return s - v;
    }
    static __device__ __forceinline__ void hist_add(unsigned int* cell) { atomicAdd(cell, 1u); }
    static __device__ __forceinline__ unsigned int slot_add(unsigned int* cell) { return atomicAdd(cell, 1u); }
    // the value lane L of the row holds, in every lane of the row (DPP row_newbcast: no LDS crossbar)
    template <int L>
    static __device__ __forceinline__ double row_bcast(double v) { return detail::dpp_d<0x150 + L>(v); }
    static __device__ __forceinline__ void cell_add(double* cell, double v) { DeviceWave::cell_add(cell, v); }
    static __device__ __forceinline__ void sync() { DeviceWave::sync(); }
    static __device__ __forceinline__ bool any(bool p) { return __any(p); }  // over the active rows: conservative
    static __device__ __forceinline__ double from_lane(double v, int src) { return __shfl(v, src, 16); }
    static __device__ __forceinline__ double uniform(double v) { return v; }
    static __device__ __forceinline__ double sum_comp(KSum k) {
        k.merge(detail::dpp_d<detail::kRor8>(k.s), detail::dpp_d<detail::kRor8>(k.c));
        k.merge(detail::dpp_d<detail::kRor4>(k.s), detail::dpp_d<detail::kRor4>(k.c));
        k.merge(detail::dpp_d<detail::kXor2>(k.s), detail::dpp_d<detail::kXor2>(k.c));
        k.merge(detail::dpp_d<detail::kXor1>(k.s), detail::dpp_d<detail::kXor1>(k.c));
        return k.value();
    }
    template <int K>
    static __device__ __forceinline__ void sum_n(double (&v)[K]) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = sum(v[k]);
    }
};
#endif

struct HostWave {
    static constexpr int W = 1;
    static inline int lane() { return 0; }
    static inline double sum(double v) { return v; }
    static inline double max(double v) { return v; }
    static inline int sumi(int v) { return v; }
    static inline int maxi(int v) { return v; }
    static inline int excl_scan_i(int) { return 0; }
    static inline void hist_add(unsigned int* cell) { *cell += 1u; }
    static inline unsigned int slot_add(unsigned int* cell) { return (*cell)++; }
    static inline void cell_add(double* cell, double v) { *cell += v; }
    static inline void sync() {}
    static inline bool any(bool p) { return p; }
    static inline double from_lane(double v, int) { return v; }
    static inline double uniform(double v) { return v; }
    static inline double sum_comp(KSum k) { return k.value(); }
    template <int K>
    static inline void sum_n(double (&)[K]) {}
};

}  // namespace dsq
