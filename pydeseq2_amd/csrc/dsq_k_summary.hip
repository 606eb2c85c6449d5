// dsq_k_summary.hip — adjusted p-values of DeseqStats.summary() (ds.py:486-542) on the device.
//
// The reference runs 50 Benjamini-Hochberg passes (one per candidate baseMean cut-off, each an
// argsort of up to G p-values) plus pandas bookkeeping.  Here the p-values are sorted ONCE (device
// radix sort of order-preserving 64-bit keys, rocPRIM: a plain library sort, the only library call
// of the engine); a gene belongs to the passes whose cut-off is <= its baseMean, i.e. to passes
// 0 .. bin-1 with bin = #{cut-offs <= baseMean}, so the rank of a p-value inside pass i is a prefix
// count over the globally sorted order.  One workgroup per pass counts its rejections; the chosen
// pass (lowess over the 50 counts, host, 50 points) is then adjusted by a reverse running minimum.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "dsq_launch.h"

namespace dsq {

constexpr int kPasses = 50;  // ds.py:503 np.linspace(lower, upper, 50)

__device__ __forceinline__ unsigned long long sum_key(double v) {  // order preserving, NaN last
    if (v != v) return ~0ull;
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_val(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__global__ void k_sum_keys(const double* __restrict__ base_mean, const double* __restrict__ pvalue, int n,
                           unsigned long long* __restrict__ key_bm, unsigned long long* __restrict__ key_p,
                           int* __restrict__ idx, int* __restrict__ counters) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < n;
    const double bm = live ? base_mean[g] : 1.0, p = live ? pvalue[g] : NAN;
    if (live) {
    key_bm[g] = sum_key(bm);
    key_p[g] = sum_key(p);
    idx[g] = g;
    }
    // lower_quantile = mean(base_mean == 0); number of genes with a p-value (one atomic per wave)
    const unsigned long long zb = __ballot(bm == 0.0), vb = __ballot(p == p);
    if ((threadIdx.x & 63) == 0) {
        if (zb) atomicAdd(&counters[0], __popcll(zb));
        if (vb) atomicAdd(&counters[1], __popcll(vb));
    }
}

// theta = linspace(lower, upper, 50); cutoffs = numpy.quantile(base_mean, theta) (method "linear":
// virtual index theta (n-1), numpy's _lerp incl. its t >= 0.5 branch).  out: theta[50] cutoffs[50]
__global__ void k_sum_quantiles(const unsigned long long* __restrict__ sorted_bm, int n,
                                const int* __restrict__ counters, double* __restrict__ out) {
    // numpy / scipy evaluate every product and sum separately: no fused multiply-adds here (a
    // contracted a + d*t moves a cut-off by an ulp and flips `base_mean >= cutoff` on ties)
#pragma clang fp contract(off)
    const int i = threadIdx.x;
    if (i >= kPasses) return;
    const double lower = (double)counters[0] / (double)n;
    const double upper = lower < 0.95 ? 0.95 : 1.0;
    const double step = (upper - lower) / (double)(kPasses - 1);
    const double theta = (i == kPasses - 1) ? upper : (double)i * step + lower;
    const double vi = theta * (double)(n - 1);
    double lo = floor(vi);
    int ilo = (int)lo, ihi = ilo + 1;
    if (ilo > n - 1) ilo = n - 1;
    if (ihi > n - 1) ihi = n - 1;
    if (ilo < 0) ilo = 0;
    const double t = vi - lo;
    const double a = key_val(sorted_bm[ilo]), b = key_val(sorted_bm[ihi]);
    const double d = b - a;
    double q = a + d * t;
    if (t >= 0.5) q = b - d * (1.0 - t);
    if (t == 0.0) q = a;  // also covers d = inf - inf when both ends are equal infinities
    out[i] = theta;
    out[kPasses + i] = q;
}

// bin[g] = #{ i : cutoffs[i] <= base_mean[g] }  (cut-offs ascend); NaN p-values belong to no pass
__global__ void k_sum_bins(const double* __restrict__ base_mean, const double* __restrict__ pvalue, int n,
                           const double* __restrict__ cut, unsigned char* __restrict__ bins) {
    __shared__ double c[kPasses];
    if (threadIdx.x < kPasses) c[threadIdx.x] = cut[kPasses + threadIdx.x];
    __syncthreads();
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const double bm = base_mean[g];
    int k = 0;
    for (int i = 0; i < kPasses; ++i) k += (bm >= c[i]) ? 1 : 0;
    bins[g] = (pvalue[g] == pvalue[g]) ? (unsigned char)k : (unsigned char)0;
}

// block-wide inclusive scan of one int per thread (1024 threads), returns this thread's inclusive
// prefix; total in *tot (valid for all threads after the call)
__device__ int block_scan_incl(int v, int* sh /*[17]*/, int* tot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(s, d, 64);
        if (lane >= d) s += o;
    }
    if (lane == 63) sh[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) { const int t = sh[q]; sh[q] = acc; acc += t; }
        sh[16] = acc;
    }
    __syncthreads();
    const int r = s + sh[w];
    *tot = sh[16];
    __syncthreads();
    return r;
}

// one workgroup per pass i: m_i = #{genes with bin > i}; rank s of a gene inside the pass = prefix
// count over the sorted order; rejections = max s with p_(s) * (m_i / s) < alpha
// (scipy.stats.false_discovery_control: ps *= m / arange(1, m+1), running minimum from the right,
// so the count of adjusted values below alpha is the largest such s).  out: num_rej[50], m[50]
__global__ __launch_bounds__(1024) void k_sum_numrej(const unsigned long long* __restrict__ sorted_p,
                                                     const int* __restrict__ sorted_idx,
                                                     const unsigned char* __restrict__ bins, int n_valid,
                                                     double alpha, double* __restrict__ out) {
    // numpy / scipy evaluate every product and sum separately: no fused multiply-adds here (a
    // contracted a + d*t moves a cut-off by an ulp and flips `base_mean >= cutoff` on ties)
#pragma clang fp contract(off)
    __shared__ int sh[17];
    __shared__ int best_sh;
    const int i = blockIdx.x;
    int cnt = 0;
    for (int k = threadIdx.x; k < n_valid; k += blockDim.x) cnt += (bins[sorted_idx[k]] > i) ? 1 : 0;
    int m = 0;
    block_scan_incl(cnt, sh, &m);
    if (threadIdx.x == 0) best_sh = 0;
    __syncthreads();
    int base = 0, best = 0;
    for (int k0 = 0; k0 < n_valid; k0 += blockDim.x) {
        const int k = k0 + threadIdx.x;
        const bool in = k < n_valid && bins[sorted_idx[k]] > i;
        int tot;
        const int s = base + block_scan_incl(in ? 1 : 0, sh, &tot);
        if (in) {
            const double adj = key_val(sorted_p[k]) * ((double)m / (double)s);
            if (adj < alpha && s > best) best = s;
        }
        base += tot;
    }
    atomicMax(&best_sh, best);
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * kPasses + i] = (double)best_sh; out[3 * kPasses + i] = (double)m; }
}

// BH-adjusted p-values of pass j (j < 0: all genes with a p-value), one workgroup:
// forward ranks, then a reverse running minimum of p_(s) m / s, clipped to [0, 1]
__global__ __launch_bounds__(1024) void k_sum_padj(const unsigned long long* __restrict__ sorted_p,
                                                   const int* __restrict__ sorted_idx,
                                                   const unsigned char* __restrict__ bins, int n, int n_valid,
                                                   int j, int* __restrict__ rank_tmp,
                                                   double* __restrict__ padj) {
    // numpy / scipy evaluate every product and sum separately: no fused multiply-adds here (a
    // contracted a + d*t moves a cut-off by an ulp and flips `base_mean >= cutoff` on ties)
#pragma clang fp contract(off)
    __shared__ int sh[17];
    __shared__ double mn[16];
    __shared__ double carry_sh;
    for (int g = threadIdx.x; g < n; g += blockDim.x) padj[g] = NAN;
    int base = 0;
    for (int k0 = 0; k0 < n_valid; k0 += blockDim.x) {
        const int k = k0 + threadIdx.x;
        const bool in = k < n_valid && (int)bins[sorted_idx[k]] > j;
        int tot;
        const int s = base + block_scan_incl(in ? 1 : 0, sh, &tot);
        if (k < n_valid) rank_tmp[k] = in ? s : 0;
        base += tot;
    }
    const int m = base;
    if (threadIdx.x == 0) carry_sh = INFINITY;
    __syncthreads();
    const int nchunk = (n_valid + (int)blockDim.x - 1) / (int)blockDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int k = c * blockDim.x + threadIdx.x;
        const int s = (k < n_valid) ? rank_tmp[k] : 0;
        double v = (s > 0) ? key_val(sorted_p[k]) * ((double)m / (double)s) : INFINITY;
        // suffix minimum inside the chunk: wave suffix scan, then across waves
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double o = __shfl_down(v, d, 64);
            if (lane + d < 64) v = o < v ? o : v;
        }
        if (lane == 0) mn[w] = v;
        __syncthreads();
        double tail = carry_sh;  // minimum over everything after this wave
        for (int q = w + 1; q < nw; ++q) tail = mn[q] < tail ? mn[q] : tail;
        const double r = tail < v ? tail : v;
        if (s > 0) padj[sorted_idx[k]] = r > 1.0 ? 1.0 : (r < 0.0 ? 0.0 : r);
        __syncthreads();
        if (threadIdx.x == 0) {
            double cm = carry_sh;
            for (int q = 0; q < nw; ++q) cm = mn[q] < cm ? mn[q] : cm;
            carry_sh = cm;
        }
        __syncthreads();
    }
}

size_t summary_sort_temp_bytes(int n) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_keys(nullptr, a, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                   (unsigned)n, 0, 64, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                    (const int*)nullptr, (int*)nullptr, (unsigned)n, 0, 64, (hipStream_t)0);
    return a > b ? a : b;
}

// work: 4 n u64 + 2 n i32 + 4 ints (layout below); out200: theta[50] cutoffs[50] num_rej[50] m[50]
hipError_t launch_padj_prepare(hipStream_t st, const double* base_mean, const double* pvalue, int n, double alpha,
                               void* sort_tmp, size_t sort_tmp_bytes, void* work, unsigned long long* sorted_p,
                               int* sorted_idx, unsigned char* bins, double* out200, int* counters) {
    unsigned long long* key_bm = (unsigned long long*)work;
    unsigned long long* key_bm_s = key_bm + n;
    unsigned long long* key_p = key_bm_s + n;
    int* idx = (int*)(key_p + n);
    hipError_t e = hipMemsetAsync(counters, 0, 4 * sizeof(int), st);
    if (e != hipSuccess) return e;
    const dim3 g((n + 255) / 256), b(256);
    hipLaunchKernelGGL(k_sum_keys, g, b, 0, st, base_mean, pvalue, n, key_bm, key_p, idx, counters);
    e = rocprim::radix_sort_keys(sort_tmp, sort_tmp_bytes, key_bm, key_bm_s, (unsigned)n, 0, 64, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sum_quantiles, dim3(1), dim3(64), 0, st, key_bm_s, n, counters, out200);
    hipLaunchKernelGGL(k_sum_bins, g, b, 0, st, base_mean, pvalue, n, out200, bins);
    e = rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, key_p, sorted_p, idx, sorted_idx, (unsigned)n, 0, 64,
                                  st);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// second half of prepare: needs the host to know n_valid (counters[1]) to size the passes
hipError_t launch_padj_numrej(hipStream_t st, const unsigned long long* sorted_p, const int* sorted_idx,
                              const unsigned char* bins, int n_valid, double alpha, double* out200) {
    hipLaunchKernelGGL(k_sum_numrej, dim3(kPasses), dim3(1024), 0, st, sorted_p, sorted_idx, bins, n_valid, alpha,
                       out200);
    return hipGetLastError();
}

hipError_t launch_padj_finish(hipStream_t st, const unsigned long long* sorted_p, const int* sorted_idx,
                              const unsigned char* bins, int n, int n_valid, int j, int* rank_tmp, double* padj) {
    hipLaunchKernelGGL(k_sum_padj, dim3(1), dim3(1024), 0, st, sorted_p, sorted_idx, bins, n, n_valid, j, rank_tmp,
                       padj);
    return hipGetLastError();
}

}  // namespace dsq
