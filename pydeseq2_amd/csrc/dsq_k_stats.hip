// dsq_k_stats.hip — layout transforms, size factors, method-of-moments, linear mu,
// Cook's distances, outlier replacement, Wald test, small reductions (gfx950).
//
// HBM-bound stages: every per-gene kernel reads a gene-major row with unit stride
// (64 lanes x 4 B / 8 B = one 256 B / 512 B request per wave instruction) and writes its
// API-visible outputs once.  The only cross-gene step that needs the other orientation is
// the per-sample median of log-ratios (size factors), which streams the sample-major
// matrix as uploaded and radix-selects each row in LDS-histogram passes.
#include <cfloat>
#include <hip/hip_cooperative_groups.h>

#include <cstdio>

#include "dsq_dispatch.h"
#include "dsq_launch.h"
#include "dsq_stats.h"
#include "dsq_trend.h"

namespace dsq {

// ------------------------------------------------------------------ wave-private LDS sort
// Bitonic sort of n doubles (padded to L = next pow2 with +inf) held in a wave-private LDS
// segment; lanes stride over compare-exchange pairs.  Only this wave touches the segment,
// LDS operations of one wave execute in order, so a wave-level fence is sufficient.
struct LdsSorter {
    __device__ __forceinline__ static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ int operator()(double* buf, int n) const {
        int L = 1;
        while (L < n) L <<= 1;
        const int lane = threadIdx.x & 63;
        for (int k = n + lane; k < L; k += 64) buf[k] = INFINITY;
        wave_sync();
        for (int k = 2; k <= L; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = lane; i < (L >> 1); i += 64) {
                    const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    const int hi = lo | j;
                    const bool up = ((lo & k) == 0);
                    const double a = buf[lo], b = buf[hi];
                    // NaNs sort last (numpy.sort semantics)
                    const bool gt = (a > b) || (a != a && b == b);
                    if (gt == up) { buf[lo] = b; buf[hi] = a; }
                }
                wave_sync();
            }
        }
        return L;
    }
    // buf[0..n) bitonic (here: decreasing then increasing), buf[n..L) = +inf from the preceding sort:
    // the final merge phase of the network alone leaves it ascending
    __device__ __forceinline__ void merge(double* buf, int n) const {
        int L = 1;
        while (L < n) L <<= 1;
        const int lane = threadIdx.x & 63;
        wave_sync();
        for (int j = L >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < (L >> 1); i += 64) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const double a = buf[lo], b = buf[hi];
                const bool gt = (a > b) || (a != a && b == b);
                if (gt) { buf[lo] = b; buf[hi] = a; }
            }
            wave_sync();
        }
    }
};

// ------------------------------------------------------------------ transposes
// sample-major [N][G] (SrcT) -> gene-major [G][ldn] (DstT) through a 64x65 LDS tile
template <class SrcT, class DstT, bool CHECK>
__global__ __launch_bounds__(256) void k_transpose(const SrcT* __restrict__ src, int N, int G,
                                                   DstT* __restrict__ dst, int ldn, int* bad) {
    __shared__ DstT tile[64][65];
    const int g0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    int isbad = 0;
    for (int r = ty; r < 64; r += 4) {
        const int n = n0 + r, g = g0 + tx;
        if (n < N && g < G) {
            const SrcT v = src[(size_t)n * G + g];
            if (CHECK) {
                if ((double)v < 0.0 || (double)v > 2147483647.0) isbad = 1;
            }
            tile[r][tx] = (DstT)v;
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int g = g0 + r, n = n0 + tx;
        if (n < N && g < G) dst[(size_t)g * ldn + n] = tile[tx][r];
    }
    if (CHECK && isbad) atomicOr(bad, 1);
}

// gene-major [G][N] (SrcT) -> gene-major [G][ldn] (DstT) (pitch / type change only)
template <class SrcT, class DstT, bool CHECK>
__global__ __launch_bounds__(256) void k_repitch(const SrcT* __restrict__ src, int N, int G,
                                                 DstT* __restrict__ dst, int ldn, int* bad) {
    const int g = blockIdx.x;  // genes on grid.x: grid.y is limited to 65535
    int isbad = 0;
    for (int n = blockIdx.y * 256 + threadIdx.x; n < N; n += gridDim.y * 256) {
        const SrcT v = src[(size_t)g * N + n];
        if (CHECK) {
            if ((double)v < 0.0 || (double)v > 2147483647.0) isbad = 1;
        }
        dst[(size_t)g * ldn + n] = (DstT)v;
    }
    if (CHECK && isbad) atomicOr(bad, 1);
}

// upload chunks that travel as uint16 (dsq_upload_counts_i32): widened into their place in the int32 matrix
__global__ __launch_bounds__(256) void k_widen_u16(const uint16_t* __restrict__ src, int32_t* __restrict__ dst, size_t n) {
    const size_t n8 = n / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 v = ((const uint4*)src)[i];  // eight counts
        int4 a, b;
        a.x = (int)(v.x & 0xffffu); a.y = (int)(v.x >> 16); a.z = (int)(v.y & 0xffffu); a.w = (int)(v.y >> 16);
        b.x = (int)(v.z & 0xffffu); b.y = (int)(v.z >> 16); b.z = (int)(v.w & 0xffffu); b.w = (int)(v.w >> 16);
        ((int4*)dst)[2 * i] = a;
        ((int4*)dst)[2 * i + 1] = b;
    }
    for (size_t i = n8 * 8 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (int)src[i];
}
hipError_t launch_widen_u16(hipStream_t st, const uint16_t* src, int32_t* dst, size_t n) {
    if (n == 0) return hipSuccess;
    // (dst = matrix base + a multiple of the chunk size: 16-byte aligned whenever the matrix is)
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_widen_u16, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_transpose_counts(hipStream_t st, const void* src, int count_type, int layout, int N,
                                   int G, int32_t* dst, int ldn, int* bad_flag) {
    if (N <= 0 || G <= 0) return hipSuccess;
    if (layout == 0) {
        const dim3 grid((G + 63) / 64, (N + 63) / 64), block(256);
        if (count_type == 1)
            hipLaunchKernelGGL((k_transpose<int64_t, int32_t, true>), grid, block, 0, st,
                               (const int64_t*)src, N, G, dst, ldn, bad_flag);
        else
            hipLaunchKernelGGL((k_transpose<int32_t, int32_t, true>), grid, block, 0, st,
                               (const int32_t*)src, N, G, dst, ldn, bad_flag);
    } else {
        const dim3 grid(G, (N + 255) / 256 > 64 ? 64 : (N + 255) / 256), block(256);
        if (count_type == 1)
            hipLaunchKernelGGL((k_repitch<int64_t, int32_t, true>), grid, block, 0, st,
                               (const int64_t*)src, N, G, dst, ldn, bad_flag);
        else
            hipLaunchKernelGGL((k_repitch<int32_t, int32_t, true>), grid, block, 0, st,
                               (const int32_t*)src, N, G, dst, ldn, bad_flag);
    }
    return hipGetLastError();
}

hipError_t launch_transpose_f64(hipStream_t st, const double* src, int layout, int N, int G,
                                double* dst, int ldn) {
    if (N <= 0 || G <= 0) return hipSuccess;
    if (layout == 0) {
        const dim3 grid((G + 63) / 64, (N + 63) / 64), block(256);
        hipLaunchKernelGGL((k_transpose<double, double, false>), grid, block, 0, st, src, N, G, dst, ldn,
                           (int*)nullptr);
    } else {
        const dim3 grid(G, (N + 255) / 256 > 64 ? 64 : (N + 255) / 256), block(256);
        hipLaunchKernelGGL((k_repitch<double, double, false>), grid, block, 0, st, src, N, G, dst, ldn,
                           (int*)nullptr);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ logmeans
__global__ __launch_bounds__(kBlock) void k_logmeans(const int32_t* __restrict__ y, int ldn, int N,
                                                     int G, double* __restrict__ logmeans,
                                                     uint8_t* __restrict__ nonzero) {
    __shared__ double s_logint[256];
    static_assert(kBlock == 256, "one table entry per thread");
    s_logint[threadIdx.x] = kLogInt[threadIdx.x];
    __syncthreads();
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    double lm;
    int nz;
    gene_logmean<DeviceWave>(y + (size_t)g * ldn, N, lm, nz, s_logint);
    if ((threadIdx.x & 63) == 0) {
        logmeans[g] = lm;
        nonzero[g] = (uint8_t)nz;
    }
}

// "poscounts" log geometric means (dds.py:655-662): mean over ALL samples of log(count), zero counts
// contributing 0; usable = finite and > 0
__global__ __launch_bounds__(kBlock) void k_logmeans_pos(const int32_t* __restrict__ y, int ldn, int N, int G,
                                                         double* __restrict__ logmeans,
                                                         uint8_t* __restrict__ usable) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    const int32_t* yr = y + (size_t)g * ldn;
    double s = 0.0;
    for (int n = threadIdx.x & 63; n < N; n += 64) {
        const int v = yr[n];
        if (v != 0) s += log_count(v);
    }
    s = DeviceWave::sum(s) / (double)N;
    if ((threadIdx.x & 63) == 0) {
        logmeans[g] = s;
        usable[g] = (uint8_t)((s > 0.0 && s != INFINITY) ? 1 : 0);
    }
}

hipError_t launch_logmeans_pos(hipStream_t st, const int32_t* y, int ldn, int N, int G, double* logmeans,
                               uint8_t* usable) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_logmeans_pos, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, ldn, N, G, logmeans, usable);
    return hipGetLastError();
}

hipError_t launch_logmeans(hipStream_t st, const int32_t* y, int ldn, int N, int G, double* logmeans,
                           uint8_t* nonzero) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_logmeans, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, ldn, N, G, logmeans,
                       nonzero);
    return hipGetLastError();
}

// ------------------------------------------------------------------ size factors
// order-preserving map double -> uint64
__device__ __forceinline__ unsigned long long f64_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// pass 1: keys[n][g] = key(log(count) - logmeans[g]) for usable genes, ~0 (sorts last) otherwise
template <class SrcT>
__global__ __launch_bounds__(256) void k_ratio_keys(const SrcT* __restrict__ counts, int N, int G,
                                                    const double* __restrict__ logmeans,
                                                    const uint8_t* __restrict__ mask,
                                                    unsigned long long* __restrict__ keys) {
    const int n = blockIdx.y;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) {
        const double lm = logmeans[g];
        const bool use = (lm != -INFINITY) && (lm == lm) && (mask == nullptr || mask[g] != 0);
        unsigned long long k = ~0ull;
        // a zero count never occurs among the genes of the default mode (their logmean is -inf); in
        // "poscounts" mode the sample's zero entries are left out of its median (dds.py:668-671)
        const double c = (double)counts[(size_t)n * G + g];
        if (use && c > 0.0) k = f64_key((c < 256.0 ? kLogInt[(int)c] : flog(c)) - lm);
        keys[(size_t)n * G + g] = k;
    }
}

// ---- workgroup-wide exact median by radix select --------------------------------------------
// 64-bit order-preserving keys are resolved 11 bits at a time (6 passes: 11,11,11,11,11,9 bits);
// both middle order statistics (ranks (M-1)/2 and M/2) are tracked in the same sweeps, and the
// first sweep's histogram also yields the number M of valid keys.  key(i) may recompute its value
// (no key array needed).  All 1024 threads of the block must call it.
struct MedianShared {
    unsigned int hist[2][2048];
    unsigned long long prefix[2];
    unsigned int rank[2];
    unsigned int M;
};

// one wave: digit d with cumsum(h[0..d-1]) <= r < cumsum(h[0..d]); r becomes the rank inside bin d
__device__ __forceinline__ int pick_digit(const unsigned int* h, int nbins, unsigned int& r) {
    const int lane = threadIdx.x & 63;
    const int per = nbins / 64;  // 32 (2048 bins) or 8 (512 bins)
    unsigned int mine = 0;
    for (int k = 0; k < per; ++k) mine += h[lane * per + k];
    unsigned int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const unsigned int excl = incl - mine;
    const bool here = (excl <= r) && (r < incl);
    const unsigned long long ball = __ballot(here);
    const int src = ball ? (__ffsll((long long)ball) - 1) : 63;
    int d = nbins - 1;
    unsigned int rr = 0;
    if (lane == src) {
        unsigned int acc = excl;
        int k = 0;
        for (; k < per - 1; ++k) {
            const unsigned int c = h[lane * per + k];
            if (acc + c > r) break;
            acc += c;
        }
        d = lane * per + k;
        rr = r - acc;
    }
    d = __shfl(d, src, 64);
    r = __shfl(rr, src, 64);
    return d;
}

// each(fn): calls fn(key) for every key this thread contributes (the same keys in every pass)
template <class Each>
__device__ void block_median_each(Each each, MedianShared& S, double& median, unsigned int& M_out) {
    const int shifts[6] = {53, 42, 31, 20, 9, 0};
    const int tid = threadIdx.x, NT = blockDim.x;
    for (int pass = 0; pass < 6; ++pass) {
        const int shift = shifts[pass];
        const int nbins = pass == 5 ? 512 : 2048;
        for (int i = tid; i < 2 * 2048; i += NT) (&S.hist[0][0])[i] = 0;
        __syncthreads();
        const unsigned long long p0 = S.prefix[0], p1 = S.prefix[1];
        const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shifts[pass - 1]));
        each([&](unsigned long long k) {
            if (k == ~0ull) return;
            const unsigned int d = (unsigned int)(k >> shift) & (unsigned int)(nbins - 1);
            if (pass == 0) {
                atomicAdd(&S.hist[0][d], 1u);
            } else {
                if ((k & himask) == p0) atomicAdd(&S.hist[0][d], 1u);
                if ((k & himask) == p1) atomicAdd(&S.hist[1][d], 1u);
            }
        });
        __syncthreads();
        if (pass == 0) {
            if (tid < 64) {  // total count and the two target ranks
                unsigned int c = 0;
                for (int k = tid; k < 2048; k += 64) c += S.hist[0][k];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
                if (tid == 0) {
                    S.M = c;
                    S.rank[0] = c ? (c - 1) / 2 : 0;
                    S.rank[1] = c / 2;
                    S.prefix[0] = 0ull;
                    S.prefix[1] = 0ull;
                }
            }
            __syncthreads();
        }
        const int w = tid >> 6;
        if (w < 2) {
            unsigned int r = S.rank[w];
            const int d = pick_digit(S.hist[pass == 0 ? 0 : w], nbins, r);
            if ((tid & 63) == 0) {
                S.rank[w] = r;
                S.prefix[w] |= ((unsigned long long)d << shift);
            }
        }
        __syncthreads();
    }
    const unsigned int M = S.M;
    M_out = M;
    if (M == 0) { median = NAN; return; }
    const double v0 = key_f64(S.prefix[0]), v1 = key_f64(S.prefix[1]);
    median = ((M - 1) / 2 == M / 2) ? v0 : (v0 + v1) / 2.0;
}

template <class KeyFn>
__device__ void block_median(KeyFn key, int n, MedianShared& S, double& median, unsigned int& M_out) {
    block_median_each(
        [&](auto fn) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) fn(key(i));
        },
        S, median, M_out);
}

// per-sample median of the log-ratio keys -> size factor (preprocessing.py:96-100)
__global__ __launch_bounds__(1024) void k_row_median(const unsigned long long* __restrict__ keys, int N,
                                                     int G, double* __restrict__ sf) {
    __shared__ MedianShared S;
    const int n = blockIdx.x;
    const unsigned long long* row = keys + (size_t)n * G;
    double med;
    unsigned int M;
    block_median([&](int g) { return row[g]; }, G, S, med, M);
    if (threadIdx.x == 0) sf[n] = (M == 0) ? NAN : exp(med);
}

// ---- workgroup-wide exact median by narrowing in VALUE space -------------------------------------
// The bitwise radix select above resolves the keys' bits from the top.  The values it is used on here - a sample's
// log ratios, the genes' log residuals - are tight clusters: their leading 11 or 22 bits are all the same, so the first
// passes narrow nothing and send every LDS histogram increment of the workgroup to the same two or three words.  This
// select bins by VALUE: a 2048-bin histogram over [min, max] of the current candidates spreads a cluster over the bins
// wherever it lies; the bin(s) that hold the two middle ranks are the new candidates; at <= 1024 candidates they are
// ranked against each other.  Binning is monotone in the value and membership is decided by the same expression in every
// pass, so the result is the exact order statistic, ties included.  Usually: one pass for min / max / count, one for the
// histogram, one to collect the candidates.
// each(fn): calls fn(x) for every value of this thread (the same values every time); NaN = not part of the median,
// -inf / +inf count at the low / high end (numpy's median: an infinite middle value gives an infinite median, opposite
// infinities NaN) and never enter a histogram.  All threads of the workgroup must call it.
constexpr int kSelCand = 1024;

struct ValueSelectShared {
    unsigned int hist[2048];
    double cand[kSelCand];
    double dred[16][2];
    unsigned int ured[16][2];
    unsigned int wtot[16];
    unsigned int bin[2], pre[2], ncand;
    double a[2];
};

// min / max of (lo, hi) and sums of (c0, c1) over the workgroup's threads; every thread gets the results
__device__ __forceinline__ void vs_block_reduce(ValueSelectShared& S, double& lo, double& hi, unsigned int& c0,
                                                unsigned int& c1) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = dmin(lo, __shfl_xor(lo, o, 64));
        hi = dmax(hi, __shfl_xor(hi, o, 64));
        c0 += __shfl_xor(c0, o, 64);
        c1 += __shfl_xor(c1, o, 64);
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();  // (the arrays may still be read from the previous reduction)
    if ((threadIdx.x & 63) == 0) { S.dred[w][0] = lo; S.dred[w][1] = hi; S.ured[w][0] = c0; S.ured[w][1] = c1; }
    __syncthreads();
    lo = S.dred[0][0]; hi = S.dred[0][1]; c0 = S.ured[0][0]; c1 = S.ured[0][1];
    for (int q = 1; q < (int)(blockDim.x >> 6); ++q) {
        lo = dmin(lo, S.dred[q][0]); hi = dmax(hi, S.dred[q][1]);
        c0 += S.ured[q][0]; c1 += S.ured[q][1];
    }
}

template <int NT, class Each>
__device__ void block_median_values(Each each, ValueSelectShared& S, double& median, unsigned int& M_out) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double lo = INFINITY, hi = -INFINITY;
    unsigned int n_fin = 0, n_low = 0, n_high = 0, none = 0;
    each([&](double x) {
        if (x != x) return;
        if (x == -INFINITY) { n_low += 1; return; }
        if (x == INFINITY) { n_high += 1; return; }
        lo = dmin(lo, x); hi = dmax(hi, x);
        n_fin += 1;
    });
    vs_block_reduce(S, lo, hi, n_fin, n_low);
    {
        double d0 = 0.0, d1 = 0.0;
        vs_block_reduce(S, d0, d1, n_high, none);
    }
    const unsigned int M = n_fin + n_low + n_high;
    M_out = M;
    if (M == 0) { median = NAN; return; }
    const unsigned int r0 = (M - 1) / 2, r1 = M / 2;
    if (r1 < n_low) { median = -INFINITY; return; }
    if (r1 >= n_low + n_fin) { median = (r0 < n_low) ? NAN : INFINITY; return; }  // (-inf + inf) / 2 = NaN, as numpy
    // ranks among the finite values (r0 may still be a -inf: then only r1 is selected)
    const bool low0 = r0 < n_low;
    unsigned int q0 = (low0 ? r1 : r0) - n_low, q1 = r1 - n_low;
    unsigned int cnt = n_fin;
    double a0 = NAN, a1 = NAN;
    for (int level = 0; level < 64; ++level) {
        if (lo == hi) { a0 = lo; a1 = lo; break; }
        const double scale = 2048.0 / (hi - lo);
        auto bin_of = [&](double x) {
            const int bb = (int)((x - lo) * scale);
            return bb > 2047 ? 2047 : bb;
        };
        const bool direct = cnt <= (unsigned int)kSelCand;  // every value in [lo, hi] is a candidate
        int b0 = 0, b1 = 2047;
        if (!direct) {
            for (int i = tid; i < 2048; i += NT) S.hist[i] = 0;
            __syncthreads();
            each([&](double x) {
                if (x >= lo && x <= hi) atomicAdd(&S.hist[bin_of(x)], 1u);
            });
            __syncthreads();
            // the bins of the two ranks: exclusive prefix over the 2048 bins, 2048 / NT consecutive bins per thread
            {
                constexpr int PER = 2048 / NT;
                unsigned int h[PER], all = 0;
#pragma unroll
                for (int q = 0; q < PER; ++q) { h[q] = S.hist[PER * tid + q]; all += h[q]; }
                unsigned int incl = all;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned int t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                if (lane == 63) S.wtot[w] = incl;
                __syncthreads();
                unsigned int base = 0;
                for (int q = 0; q < w; ++q) base += S.wtot[q];
                unsigned int excl = base + incl - all;
#pragma unroll
                for (int q = 0; q < PER; ++q) {
#pragma unroll
                    for (int which = 0; which < 2; ++which) {
                        const unsigned int t = which ? q1 : q0;
                        if (excl <= t && t < excl + h[q]) { S.bin[which] = PER * tid + q; S.pre[which] = excl; }
                    }
                    excl += h[q];
                }
                __syncthreads();
            }
            b0 = (int)S.bin[0]; b1 = (int)S.bin[1];
            const unsigned int pre0 = S.pre[0];
            const unsigned int c0 = S.hist[b0], csel = c0 + (b1 != b0 ? S.hist[b1] : 0u);
            q0 -= pre0; q1 -= pre0;
            if (csel > (unsigned int)kSelCand) {
                // still too many: the ranks are neighbours, so either both lie in bin b0 - the next level's range - or
                // they are the LAST value of bin b0 and the FIRST of bin b1
                double nlo = INFINITY, nhi = -INFINITY, lo1 = INFINITY;
                unsigned int d0 = 0, d1 = 0;
                each([&](double x) {
                    if (x >= lo && x <= hi) {
                        const int bb = bin_of(x);
                        if (bb == b0) { nlo = dmin(nlo, x); nhi = dmax(nhi, x); }
                        if (bb == b1) lo1 = dmin(lo1, x);
                    }
                });
                if (b0 != b1) {
                    vs_block_reduce(S, lo1, nhi, d0, d1);
                    a0 = nhi; a1 = lo1;
                    break;
                }
                vs_block_reduce(S, nlo, nhi, d0, d1);
                lo = nlo; hi = nhi; cnt = c0;
                continue;
            }
        }
        // rank the candidates against each other
        if (tid == 0) { S.ncand = 0; S.a[0] = NAN; S.a[1] = NAN; }
        __syncthreads();
        each([&](double x) {
            if (x >= lo && x <= hi) {
                bool take = direct;
                if (!direct) {
                    const int bb = bin_of(x);
                    take = bb == b0 || bb == b1;
                }
                if (take) S.cand[atomicAdd(&S.ncand, 1u)] = x;
            }
        });
        __syncthreads();
        const unsigned int m = S.ncand;
        for (int i = tid; i < (int)m; i += NT) {
            const double c = S.cand[i];
            unsigned int less = 0, eq = 0;
            for (unsigned int j = 0; j < m; ++j) {
                const double cj = S.cand[j];
                less += cj < c;
                eq += cj == c;
            }
            if (less <= q0 && q0 < less + eq) S.a[0] = c;
            if (less <= q1 && q1 < less + eq) S.a[1] = c;
        }
        __syncthreads();
        a0 = S.a[0]; a1 = S.a[1];
        break;
    }
    const double v0 = low0 ? -INFINITY : a0;
    median = (r0 == r1) ? v0 : (v0 + a1) / 2.0;
    __syncthreads();  // (S may be reused by the caller's next select)
}

// dispersion prior (dds.py:866-884, utils.py:1210-1227): MAD^2 of log(genewise) - log(fitted) over the genes with
// genewise >= 100 * min_disp.  The residuals come from a wide kernel (two logarithms per gene: one workgroup's ALUs
// would need longer for them than for everything else), the two medians - of the residuals, then of their absolute
// deviations from the first - from one workgroup that reads them from the L2 (three passes each, usually).
// out[0] = squared_logres, out[1] = number of genes that entered.
__global__ void k_prior_res(const double* __restrict__ gw_raw, const double* __restrict__ fitted, int n,
                            double min_disp, double max_disp, double* __restrict__ res) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double g = dmin(dmax(gw_raw[i], min_disp), max_disp);
    res[i] = (g >= 100.0 * min_disp) ? log(g) - log(fitted[i]) : NAN;
}

__global__ __launch_bounds__(1024) void k_prior_mad(const double* __restrict__ res, int n, double* __restrict__ out) {
    __shared__ ValueSelectShared S;
    const int tid = threadIdx.x;
    double center, mad;
    unsigned int M, M2;
    // (eight loads in flight per thread: the values are independent, a one-value loop body waits for each of them -
    // at 60 000 genes a pass is 59 L2 round trips per thread then, and the passes are nothing else)
    auto walk = [&](auto&& f) {
        int i = tid;
        for (; i + 7 * 1024 < n; i += 8 * 1024) {
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = res[i + q * 1024];
#pragma unroll
            for (int q = 0; q < 8; ++q) f(x[q]);
        }
        for (; i < n; i += 1024) f(res[i]);
    };
    block_median_values<1024>([&](auto fn) { walk(fn); }, S, center, M);
    block_median_values<1024>([&](auto fn) { walk([&](double r) { fn(fabs(r - center)); }); }, S, mad, M2);
    if (tid == 0) {
        const double m = mad / 0.67448975019608171;  // norm.ppf(0.75)
        out[0] = m * m;
        out[1] = (double)M;
    }
}

// res_scratch: n doubles
size_t prior_mad_work_doubles(int n) { return (size_t)n + 8; }

hipError_t launch_prior_mad(hipStream_t st, const double* gw_raw, const double* fitted, int n, double min_disp,
                            double max_disp, double* res_scratch, double* out2) {
    if (n > 0)
        hipLaunchKernelGGL(k_prior_res, dim3((n + 255) / 256), dim3(256), 0, st, gw_raw, fitted, n, min_disp, max_disp,
                           res_scratch);
    hipLaunchKernelGGL(k_prior_mad, dim3(1), dim3(1024), 0, st, (const double*)res_scratch, n, out2);
    return hipGetLastError();
}

// ---- compacted variant: only the usable genes (finite logmean, mask) get keys.  With ~1000 samples
// most genes contain a zero somewhere (75 % in the benchmark data), so the key matrix and the six radix
// passes over it shrink accordingly.  idx[j] = j-th usable gene, *count = their number.
__global__ __launch_bounds__(256) void k_sf_compact(const double* __restrict__ logmeans,
                                                    const uint8_t* __restrict__ mask, int G,
                                                    int* __restrict__ idx, int* __restrict__ count) {
    // one atomic per wave reserves a slot range; the order of idx is irrelevant (only medians over the
    // usable genes are taken), so no scan is needed.  *count must be zero at launch.
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    bool use = false;
    if (g < G) {
        const double lm = logmeans[g];
        use = (lm != -INFINITY) && (lm == lm) && (mask == nullptr || mask[g] != 0);
    }
    const unsigned long long b = __ballot(use);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && b) base = atomicAdd(count, __popcll(b));
    base = __shfl(base, 0, 64);
    if (use) idx[base + __popcll(b & ((1ull << lane) - 1ull))] = g;
}

// ---- the usual case - at most kSfRegGenes = 32 768 usable genes: ONE kernel, one workgroup per sample, the
// sample's log ratios in REGISTERS, and a selection that suits them.
// The key-matrix path below is a bitwise radix select: it writes N x Gu keys, reads them six times (r03: 4.4 x the bytes
// of the counts), and - log ratios of one sample are a tight cluster, so their leading 11 or 22 bits are all the same -
// its first passes send every LDS histogram increment of a workgroup to the same two or three words.  Here a thread
// gathers its <= 32 counts once and the median is narrowed in VALUE space: a 2048-bin histogram over [min, max] of the
// current candidates spreads the cluster over the bins whatever its location; the bin(s) holding the two middle ranks
// become the new candidate set; at <= 1024 candidates they are ranked against each other exactly.  Binning is monotone
// in the value and membership is decided by the same expression in every pass, so the result is the exact order
// statistic (ties included); usually one histogram level is enough.  More usable genes than the registers hold: this
// kernel leaves at once and the two kernels below (which leave at once in the usual case) do the work.
constexpr int kSfRegGenes = 32768;  // 1024 threads x 32 genes each

template <class SrcT, int NT, int kSfRegKeys>
__global__ __launch_bounds__(NT) void k_sf_row(const SrcT* __restrict__ counts, int N, int G,
                                                 const double* __restrict__ logmeans, const int* __restrict__ idx,
                                                 const int* __restrict__ count, double* __restrict__ sf,
                                                 int zeros_low) {
    __shared__ ValueSelectShared S;
    const int tid = threadIdx.x;
    const int n = blockIdx.x, Gu = *count;
    if (Gu > NT * kSfRegKeys) return;
    // this thread's log ratios: NaN = not part of the median (a zero count in the training data: see k_ratio_keys_c),
    // -inf = a zero count of a NEW sample (counts at the low end, as numpy's median has it)
    double v[kSfRegKeys];
#pragma unroll
    for (int k = 0; k < kSfRegKeys; ++k) v[k] = NAN;
    // (every loop over a thread's slots stops at the sample's last occupied slot - a uniform branch: the slots are sized
    // for 32 768 usable genes, a sample of the benchmark has 15 000, one of a gene shard 2 000)
    const int kmax = (Gu + NT - 1) / NT;
    {
        // straight-line gathers, eight genes at a time: gene index -> count and log mean -> table logarithm are three
        // dependent memory round trips, and with a branch per gene they ran one gene after the other (2 x 15 HBM
        // latencies per thread: 316 us for 1000 samples where the bytes need 60)
#pragma unroll
        for (int k0 = 0; k0 < kSfRegKeys; k0 += 8) {
            if (k0 >= kmax) continue;
            __builtin_amdgcn_sched_barrier(0);
            int g[8];
            double c[8], lm[8], tl[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int j = tid + (k0 + kk) * NT;
                g[kk] = idx[j < Gu ? j : Gu - 1];
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                c[kk] = (double)counts[(size_t)n * G + g[kk]];
                lm[kk] = logmeans[g[kk]];
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) tl[kk] = kLogInt[c[kk] < 256.0 ? (c[kk] > 0.0 ? (int)c[kk] : 1) : 255];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bool in = tid + (k0 + kk) * NT < Gu, pos = c[kk] > 0.0;
                const double x = (c[kk] < 256.0 ? tl[kk] : flog(pos ? c[kk] : 1.0)) - lm[kk];
                const bool fin = in && pos, low = in && !pos && zeros_low != 0;
                v[k0 + kk] = fin ? x : (low ? -INFINITY : NAN);
            }
        }
    }
    double med;
    unsigned int M;
    block_median_values<NT>(
        [&](auto fn) {
#pragma unroll
            for (int k = 0; k < kSfRegKeys; ++k)
                if (k < kmax) fn(v[k]);
        },
        S, med, M);
    if (tid == 0) sf[n] = (M == 0) ? NAN : exp(med);
}

template <class SrcT>
__global__ __launch_bounds__(256) void k_ratio_keys_c(const SrcT* __restrict__ counts, int N, int G,
                                                      const double* __restrict__ logmeans,
                                                      const int* __restrict__ idx, const int* __restrict__ count,
                                                      unsigned long long* __restrict__ keys, int zeros_low,
                                                      int after_sf_row) {
    const int n = blockIdx.y, Gu = *count;
    if (after_sf_row && Gu <= kSfRegGenes) return;  // k_sf_row's case
    // a zero count: never among the usable genes of the training data in "ratio" mode; left out of the sample's
    // median in "poscounts" mode (dds.py:668-671); for NEW samples transformed with the training log means
    // (zeros_low) it is log(0) - logmean = -inf and counts at the low end of the median, as numpy does
    // (preprocessing.py:59-102)
    const unsigned long long kz = zeros_low ? f64_key(-INFINITY) : ~0ull;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < Gu; j += gridDim.x * 256) {
        const int g = idx[j];
        const double c = (double)counts[(size_t)n * G + g];
        keys[(size_t)n * Gu + j] = (c > 0.0) ? f64_key((c < 256.0 ? kLogInt[(int)c] : flog(c)) - logmeans[g]) : kz;
    }
}

__global__ __launch_bounds__(1024) void k_row_median_c(const unsigned long long* __restrict__ keys, int N,
                                                       const int* __restrict__ count, double* __restrict__ sf) {
    __shared__ MedianShared S;
    const int n = blockIdx.x, Gu = *count;
    if (Gu <= kSfRegGenes) return;  // k_sf_row has written this sample's size factor
    const unsigned long long* row = keys + (size_t)n * Gu;
    double med;
    unsigned int M;
    block_median([&](int g) { return row[g]; }, Gu, S, med, M);
    if (threadIdx.x == 0) sf[n] = (M == 0) ? NAN : exp(med);
}

// work: N*G u64 keys (worst case) followed by G + 2 ints (dsq_size_factors_work_doubles)
size_t size_factors_work_doubles(int N, int G) { return (size_t)N * G + (size_t)G / 2 + 8; }

hipError_t launch_size_factors(hipStream_t st, const void* counts_sm, int count_type, int N, int G,
                               const double* logmeans, const uint8_t* gene_mask, double* work,
                               double* sf, int zeros_low) {
    if (N <= 0 || G <= 0) return hipSuccess;
    unsigned long long* keys = (unsigned long long*)work;
    int* idx = (int*)(work + (size_t)N * G);
    int* count = idx + G;
    hipError_t e0 = hipMemsetAsync(count, 0, sizeof(int), st);
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL(k_sf_compact, dim3((G + 255) / 256), dim3(256), 0, st, logmeans, gene_mask, G, idx, count);
    // (samples with few usable genes - a gene shard, a small panel - in 256-thread workgroups, four of them per compute
    // unit: the per-sample passes are latency, not work, and a 5000-sample shard has 20 samples per compute unit)
    if (G <= 256 * 32) {
        if (count_type == 1)
            hipLaunchKernelGGL((k_sf_row<int64_t, 256, 32>), dim3(N), dim3(256), 0, st, (const int64_t*)counts_sm, N, G,
                               logmeans, (const int*)idx, (const int*)count, sf, zeros_low);
        else
            hipLaunchKernelGGL((k_sf_row<int32_t, 256, 32>), dim3(N), dim3(256), 0, st, (const int32_t*)counts_sm, N, G,
                               logmeans, (const int*)idx, (const int*)count, sf, zeros_low);
        return hipGetLastError();
    }
    if (count_type == 1)
        hipLaunchKernelGGL((k_sf_row<int64_t, 1024, 32>), dim3(N), dim3(1024), 0, st, (const int64_t*)counts_sm, N, G,
                           logmeans, (const int*)idx, (const int*)count, sf, zeros_low);
    else
        hipLaunchKernelGGL((k_sf_row<int32_t, 1024, 32>), dim3(N), dim3(1024), 0, st, (const int32_t*)counts_sm, N, G,
                           logmeans, (const int*)idx, (const int*)count, sf, zeros_low);
    if (G <= kSfRegGenes) return hipGetLastError();  // (the number of usable genes is known on the device only)
    // (few, grid-striding workgroups: in the usual case they all leave at once, and a launch of N x 235 empty workgroups
    // costs more than k_sf_row)
    int gx = (G + 255) / 256 > 256 ? 256 : (G + 255) / 256;
    if (gx > 4096 / N + 1) gx = 4096 / N + 1;
    if (count_type == 1)
        hipLaunchKernelGGL((k_ratio_keys_c<int64_t>), dim3(gx, N), dim3(256), 0, st, (const int64_t*)counts_sm, N, G,
                           logmeans, (const int*)idx, (const int*)count, keys, zeros_low, 1);
    else
        hipLaunchKernelGGL((k_ratio_keys_c<int32_t>), dim3(gx, N), dim3(256), 0, st, (const int32_t*)counts_sm, N, G,
                           logmeans, (const int*)idx, (const int*)count, keys, zeros_low, 1);
    hipLaunchKernelGGL(k_row_median_c, dim3(N), dim3(1024), 0, st, (const unsigned long long*)keys, N,
                       (const int*)count, sf);
    return hipGetLastError();
}

// ------------------------------------------------------------------ MoM / linear mu
__global__ void k_mean_inv(const double* __restrict__ sf, int N, double* out) {
    // single block reduction of mean(1/sf)
    __shared__ double part[256];
    double s = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) s += 1.0 / sf[n];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] / (double)N;
}

constexpr int kMomGenes = 4;  // genes per wavefront of the method-of-moments kernels
static inline int mom_blocks(int G) { return (G + kWavesPerBlock * kMomGenes - 1) / (kWavesPerBlock * kMomGenes); }

template <int P>
__global__ __launch_bounds__(kBlock) void k_mom(const int32_t* __restrict__ y, int ldn,
                                                const double* __restrict__ sf,
                                                const double* __restrict__ Xt,
                                                const double* __restrict__ pinvXt, int ldx, int N, int G,
                                                const double* __restrict__ s_mean_inv, double min_disp,
                                                double max_disp, double* __restrict__ normed_mean,
                                                double* __restrict__ rough, double* __restrict__ moments,
                                                double* __restrict__ mom) {
    // four genes per wavefront: the shared vectors are read once for four count rows (dsq_stats.h, mom_lin_mu_block)
    const int g0 = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * kMomGenes;
    if (g0 >= G) return;
    const int nv = G - g0 < kMomGenes ? G - g0 : kMomGenes;
    MomOut o[kMomGenes];
    mom_lin_mu_block<DeviceWave, P, kMomGenes>(y + (size_t)g0 * ldn, ldn, nv, sf, Xt, pinvXt, ldx, N, s_mean_inv[0],
                                               min_disp, max_disp, 0.0, nullptr, nullptr, o);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kMomGenes; ++k) {
            if (k < nv) {
                normed_mean[g0 + k] = o[k].normed_mean;
                if (rough) rough[g0 + k] = o[k].rough;
                if (moments) moments[g0 + k] = o[k].moments;
                mom[g0 + k] = o[k].mom;
            }
        }
    }
}

hipError_t launch_mom(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                      const double* pinvXt, int ldx, int N, int G, int P_, double min_disp,
                      double max_disp, double* normed_mean, double* rough, double* moments,
                      double* mom, double* d_scalar, const double* sf_moments) {
    if (G <= 0) return hipSuccess;
    // sf_moments: the size factors whose mean reciprocal enters the moments estimate when they differ
    // from the ones the counts are normalised with (iterative size factors, dds.py:1149-1156 there)
    hipLaunchKernelGGL(k_mean_inv, dim3(1), dim3(256), 0, st, sf_moments ? sf_moments : sf, N, d_scalar);
    if (P_ > DSQ_REG_MAX_P)
        return launch_wide_mom(st, y, ldn, sf, Xt, pinvXt, ldx, N, G, P_, min_disp, max_disp, 0.5, normed_mean, rough,
                               moments, mom, nullptr, nullptr, d_scalar);
    const dim3 grid(mom_blocks(G)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_mom<P>, grid, block, 0, st, y, ldn, sf, Xt, pinvXt, ldx, N, G,
                                          (const double*)d_scalar, min_disp, max_disp, normed_mean, rough,
                                          moments, mom))
    return hipGetLastError();
}

template <int P>
__global__ __launch_bounds__(kBlock) void k_mom_lin_mu(const int32_t* __restrict__ y, int ldn,
                                                       const double* __restrict__ sf,
                                                       const double* __restrict__ Xt,
                                                       const double* __restrict__ pinvXt, int ldx, int N, int G,
                                                       const double* __restrict__ s_mean_inv, double min_disp,
                                                       double max_disp, double min_mu,
                                                       double* __restrict__ normed_mean, double* __restrict__ mom,
                                                       double* __restrict__ mu, double* __restrict__ coef) {
    const int g0 = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * kMomGenes;
    if (g0 >= G) return;
    const int nv = G - g0 < kMomGenes ? G - g0 : kMomGenes;
    MomOut o[kMomGenes];
    mom_lin_mu_block<DeviceWave, P, kMomGenes>(y + (size_t)g0 * ldn, ldn, nv, sf, Xt, pinvXt, ldx, N, s_mean_inv[0],
                                               min_disp, max_disp, min_mu, mu ? mu + (size_t)g0 * ldn : nullptr,
                                               coef ? coef + (size_t)g0 * P : nullptr, o);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kMomGenes; ++k) {
            if (k < nv) {
                normed_mean[g0 + k] = o[k].normed_mean;
                mom[g0 + k] = o[k].mom;
            }
        }
    }
}

hipError_t launch_mom_lin_mu(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                             const double* pinvXt, int ldx, int N, int G, int P_, double min_disp, double max_disp,
                             double min_mu, double* normed_mean, double* mom, double* mu, double* d_scalar,
                             double* coef) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mean_inv, dim3(1), dim3(256), 0, st, sf, N, d_scalar);
    if (P_ > DSQ_REG_MAX_P)
        return launch_wide_mom(st, y, ldn, sf, Xt, pinvXt, ldx, N, G, P_, min_disp, max_disp, min_mu, normed_mean,
                               nullptr, nullptr, mom, mu, coef, d_scalar);
    const dim3 grid(mom_blocks(G)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_mom_lin_mu<P>, grid, block, 0, st, y, ldn, sf, Xt, pinvXt, ldx, N, G,
                                          (const double*)d_scalar, min_disp, max_disp, min_mu, normed_mean, mom, mu,
                                          coef))
    return hipGetLastError();
}

template <int P>
__global__ __launch_bounds__(kBlock) void k_lin_mu(const int32_t* __restrict__ y, int ldn,
                                                   const double* __restrict__ sf,
                                                   const double* __restrict__ Xt,
                                                   const double* __restrict__ pinvXt, int ldx, int N,
                                                   int G, double min_mu, double* __restrict__ mu) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    lin_mu_gene<DeviceWave, P>(y + (size_t)g * ldn, sf, Xt, pinvXt, ldx, N, min_mu, mu + (size_t)g * ldn);
}

hipError_t launch_lin_mu(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                         const double* pinvXt, int ldx, int N, int G, int P_, double min_mu,
                         double* mu) {
    if (G <= 0) return hipSuccess;
    if (P_ > DSQ_REG_MAX_P) {  // s_mean_inv is not used for mu_hat: any finite scalar (sf[0]) serves
        return launch_wide_mom(st, y, ldn, sf, Xt, pinvXt, ldx, N, G, P_, 1e-8, 1.0, min_mu, nullptr, nullptr, nullptr,
                               nullptr, mu, nullptr, sf);
    }
    const dim3 grid(genes_to_blocks(G)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_lin_mu<P>, grid, block, 0, st, y, ldn, sf, Xt, pinvXt, ldx, N,
                                          G, min_mu, mu))
    return hipGetLastError();
}

// mu_hat rows of a few listed genes from their OLS coefficients (the grid-search fallback of the dispersion fit
// when no N x G mu_hat was materialised): dst[k][:] = max(sf * (X coef[list[k]]), min_mu), idx_out[k] = k
// EXP: the IRLS route's mu_hat instead, sf * exp(X beta) UNclamped (dds.py:757-771, utils.py:435-437)
template <int P, bool EXP = false>
__global__ __launch_bounds__(kBlock) void k_mu_from_coef(const double* __restrict__ coef, const double* __restrict__ sf,
                                                         const double* __restrict__ Xt, int ldx, int N, double min_mu,
                                                         const int32_t* __restrict__ list, int n_list,
                                                         double* __restrict__ dst, int ldn, int32_t* __restrict__ idx_out,
                                                         const int32_t* __restrict__ n_dev) {
    const int k = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (n_dev != nullptr) n_list = min(n_list, *n_dev);  // launched for a capacity, the count lives on the device
    if (k >= n_list) return;
    const int g = list[k];
    double b[P];
#pragma unroll
    for (int j = 0; j < P; ++j) b[j] = coef[(size_t)g * P + j];
    // (blockIdx.y: a slice of the row - the list is a handful of genes, and one wavefront alone walking 5000 samples
    // took 61 us of load latency)
    for (int n = blockIdx.y * 64 + DeviceWave::lane(); n < N; n += 64 * gridDim.y) {
        double yh = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) yh += Xt[j * ldx + n] * b[j];
        dst[(size_t)k * ldn + n] = EXP ? sf[n] * exp(yh) : dmax(sf[n] * yh, min_mu);
    }
    if ((threadIdx.x & 63) == 0 && blockIdx.y == 0) idx_out[k] = k;
}
static inline unsigned mu_row_slices(int N) {
    const int s = (N + 255) / 256;
    return (unsigned)(s < 1 ? 1 : (s > 16 ? 16 : s));
}

hipError_t launch_mu_from_coef(hipStream_t st, const double* coef, const double* sf, const double* Xt, int ldx, int N,
                               int P_, double min_mu, const int32_t* list, int n_list, double* dst, int ldn,
                               int32_t* idx_out, const int32_t* n_dev) {
    if (n_list <= 0) return hipSuccess;
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_mu_from_coef<P>, dim3(genes_to_blocks(n_list), mu_row_slices(N)), dim3(kBlock), 0, st, coef,
                                          sf, Xt, ldx, N, min_mu, list, n_list, dst, ldn, idx_out, n_dev))
    return hipGetLastError();
}

hipError_t launch_mu_from_beta(hipStream_t st, const double* beta, const double* sf, const double* Xt, int ldx, int N,
                               int P_, const int32_t* list, int n_list, double* dst, int ldn, int32_t* idx_out,
                               const int32_t* n_dev) {
    if (n_list <= 0) return hipSuccess;
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL((k_mu_from_coef<P, true>), dim3(genes_to_blocks(n_list), mu_row_slices(N)), dim3(kBlock), 0, st,
                                          beta, sf, Xt, ldx, N, 0.0, list, n_list, dst, ldn, idx_out, n_dev))
    return hipGetLastError();
}

// rough / moments dispersions from already-normalised counts (Inference-level entry points)
template <int P>
__global__ __launch_bounds__(kBlock) void k_rough_normed(const double* __restrict__ normed, int ldn,
                                                         const double* __restrict__ Xt,
                                                         const double* __restrict__ pinvXt, int ldx,
                                                         int N, int G, double* __restrict__ out) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    const double* v = normed + (size_t)g * ldn;
    double b[P];
#pragma unroll
    for (int j = 0; j < P; ++j) b[j] = 0.0;
    for (int n = DeviceWave::lane(); n < N; n += 64) {
#pragma unroll
        for (int j = 0; j < P; ++j) b[j] += pinvXt[j * ldx + n] * v[n];
    }
    DeviceWave::sum_n<P>(b);
    double rr = 0.0;
    const double dof = (double)(N - P);
    for (int n = DeviceWave::lane(); n < N; n += 64) {
        double yh = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) yh += Xt[j * ldx + n] * b[j];
        yh = dmax(yh, 1.0);
        rr += ((v[n] - yh) * (v[n] - yh) - yh) / (dof * yh * yh);
    }
    rr = DeviceWave::sum(rr);
    if ((threadIdx.x & 63) == 0) out[g] = dmax(rr, 0.0);
}

hipError_t launch_rough_from_normed(hipStream_t st, const double* normed, int ldn, const double* Xt,
                                    const double* pinvXt, int ldx, int N, int G, int P_, double* out) {
    if (G <= 0) return hipSuccess;
    if (P_ > DSQ_REG_MAX_P) return launch_wide_rough_normed(st, normed, ldn, Xt, pinvXt, ldx, N, G, P_, out);
    const dim3 grid(genes_to_blocks(G)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_rough_normed<P>, grid, block, 0, st, normed, ldn, Xt, pinvXt,
                                          ldx, N, G, out))
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_moments_normed(const double* __restrict__ normed, int ldn,
                                                           int N, int G, double s_mean_inv,
                                                           double* __restrict__ out) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    const double* v = normed + (size_t)g * ldn;
    double s = 0.0;
    for (int n = DeviceWave::lane(); n < N; n += 64) s += v[n];
    const double mean = DeviceWave::sum(s) / (double)N;
    double ss = 0.0;
    for (int n = DeviceWave::lane(); n < N; n += 64) ss += (v[n] - mean) * (v[n] - mean);
    const double var = DeviceWave::sum(ss) / (double)(N - 1);
    double m = (var - s_mean_inv * mean) / (mean * mean);
    if (m != m) m = 0.0;
    else if (m == INFINITY) m = DBL_MAX;
    else if (m == -INFINITY) m = -DBL_MAX;
    if ((threadIdx.x & 63) == 0) out[g] = m;
}

hipError_t launch_moments_from_normed(hipStream_t st, const double* normed, int ldn, int N, int G,
                                      double s_mean_inv, double* out) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_moments_normed, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, normed, ldn, N, G,
                       s_mean_inv, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ Wald
template <int P>
__global__ __launch_bounds__(kBlock) void k_wald(const double* __restrict__ mu, int ldn,
                                                 const double* __restrict__ sf,
                                                 const double* __restrict__ Xt, int ldx, int N, int G,
                                                 const double* __restrict__ disp,
                                                 const double* __restrict__ beta,
                                                 const double* __restrict__ ridge,
                                                 const double* __restrict__ contrast, double lfc_null,
                                                 int alt, double* __restrict__ pvals,
                                                 double* __restrict__ stats, double* __restrict__ se) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    double b[P];
#pragma unroll
    for (int j = 0; j < P; ++j) b[j] = beta[(size_t)g * P + j];
    const WaldOut o = wald_gene<DeviceWave, P>(mu ? mu + (size_t)g * ldn : nullptr, sf, Xt, ldx, N,
                                               disp[g], b, ridge, contrast, lfc_null, alt);
    if ((threadIdx.x & 63) == 0) {
        pvals[g] = o.p;
        stats[g] = o.stat;
        se[g] = o.se;
    }
}

hipError_t launch_wald(hipStream_t st, const double* mu, int ldn, const double* sf, const double* Xt,
                       int ldx, int N, int G, int P_, const double* disp, const double* beta,
                       const double* d_ridge, const double* d_contrast, double lfc_null, int alt,
                       double* pvals, double* stats, double* se) {
    if (G <= 0) return hipSuccess;
    if (P_ > DSQ_REG_MAX_P)
        return launch_wide_wald(st, mu, ldn, sf, Xt, ldx, N, G, P_, disp, beta, d_ridge, d_contrast, lfc_null, alt,
                                pvals, stats, se);
    const dim3 grid(genes_to_blocks(G)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_wald<P>, grid, block, 0, st, mu, ldn, sf, Xt, ldx, N, G, disp,
                                          beta, d_ridge, d_contrast, lfc_null, alt, pvals, stats, se))
    return hipGetLastError();
}

// ------------------------------------------------------------------ Cook's
// WPB waves per block share the dynamic LDS: each wave owns `cap` doubles.
template <int WPB, bool BIG>
__global__ __launch_bounds__(64 * WPB) void k_cooks(const int32_t* __restrict__ y, int ldn,
                                                    const double* __restrict__ sf,
                                                    const double* __restrict__ mu,
                                                    const double* __restrict__ hat,
                                                    const int32_t* __restrict__ cell_offsets,
                                                    const int32_t* __restrict__ cell_index, int n_cells,
                                                    int whole, int cap, int stride,
                                                    const uint8_t* __restrict__ flags,
                                                    int N, int G, int P, double cutoff,
                                                    double* __restrict__ cooks,
                                                    double* __restrict__ robust_disp,
                                                    uint8_t* __restrict__ any_all,
                                                    uint8_t* __restrict__ any_use,
                                                    uint8_t* __restrict__ any_use_nr,
                                                    uint8_t* __restrict__ few_above) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int w = threadIdx.x >> 6;
    const int g = blockIdx.x * WPB + w;
    if (g >= G) return;
    // per wave: cap doubles of values (+ a BucketWork when a cell is large enough for the bucket path;
    // `stride` is the per-wave segment in doubles)
    double* scratch = lds + (size_t)w * stride;
    unsigned int* hist = (unsigned int*)(scratch + cap);
    CellPlan C{cell_offsets, cell_index, n_cells, whole};
    const CooksOut o = cooks_gene<DeviceWave, BIG>(y + (size_t)g * ldn, sf, mu + (size_t)g * ldn,
                                              hat + (size_t)g * ldn, C, flags, N, P, cutoff, scratch,
                                              hist, LdsSorter(), cooks ? cooks + (size_t)g * ldn : nullptr);
    if ((threadIdx.x & 63) == 0) {
        robust_disp[g] = o.robust_disp;
        any_all[g] = (uint8_t)o.any_gt_all;
        any_use[g] = (uint8_t)o.any_gt_use;
        any_use_nr[g] = (uint8_t)o.any_gt_use_nr;
        few_above[g] = (uint8_t)o.few_above;
    }
}

static int next_pow2(int n) {
    int L = 1;
    while (L < n) L <<= 1;
    return L;
}
// per-wave LDS of the trimmed statistics (dsq_stats.h, robust_disp_gene): room for the values of the largest cell
// (a power of two for cells that are sorted) and, with a cell of kTrimBucketMin samples or more, a BucketWork behind it
static int trim_cap(int biggest) {
    return biggest < kTrimBucketMin ? next_pow2(biggest) : ((biggest + 15) & ~15);  // (>= 128: smaller cells of the same design)
}
static int trim_work_doubles(int biggest) {
    return biggest < kTrimBucketMin ? 0 : (int)((sizeof(BucketWork) + 7) / 8);
}

hipError_t launch_cooks(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* mu,
                        const double* hat, const int32_t* cell_offsets, const int32_t* cell_index,
                        int n_cells, int whole, int max_cell, const uint8_t* flags, int N, int G,
                        int P, double cutoff, double* cooks, double* robust_disp, uint8_t* any_all,
                        uint8_t* any_use, uint8_t* any_use_nr, uint8_t* few_above) {
    if (G <= 0) return hipSuccess;
    const int biggest = whole ? N : max_cell;  // sorted cells need power-of-two room, selected ones do not
    const int cap = trim_cap(biggest);
    const int stride = cap + trim_work_doubles(biggest);
    const size_t per_wave = (size_t)stride * sizeof(double);
    const bool big = biggest >= kTrimBucketMin;
    if (per_wave > 160 * 1024) return hipErrorInvalidValue;  // > ~20000 samples in one cell
#define DSQ_COOKS_LAUNCH(WPB) \
    do { if (big) DSQ_COOKS_LAUNCH_(WPB, true); else DSQ_COOKS_LAUNCH_(WPB, false); } while (0)
#define DSQ_COOKS_LAUNCH_(WPB, BIG)                                                                    \
    do {                                                                                               \
        if (per_wave * WPB > 48 * 1024) {                                                              \
            (void)hipFuncSetAttribute((const void*)k_cooks<WPB, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(per_wave * WPB));                                          \
            (void)hipGetLastError();                                                                   \
        }                                                                                              \
        hipLaunchKernelGGL((k_cooks<WPB, BIG>), dim3((G + WPB - 1) / WPB), dim3(64 * WPB), per_wave * WPB, st, \
                           y, ldn, sf, mu, hat, cell_offsets, cell_index, n_cells, whole, cap, stride, flags, \
                           N, G, P, cutoff, cooks, robust_disp, any_all, any_use, any_use_nr,         \
                           few_above);                                                                \
    } while (0)
    if (per_wave * 4 <= 64 * 1024) DSQ_COOKS_LAUNCH(4);
    else if (per_wave * 2 <= 160 * 1024) DSQ_COOKS_LAUNCH(2);
    else DSQ_COOKS_LAUNCH(1);
#undef DSQ_COOKS_LAUNCH
#undef DSQ_COOKS_LAUNCH_
    return hipGetLastError();
}

// The design-only half of the Cook's stage on its own (robust_disp_gene): what the fused LFC epilogue
// (dsq_irls.h, LfcEpilogue) needs beforehand.  Independent of every fit, so the pipeline runs it on a side
// stream underneath the latency-bound dispersion-trend / prior kernels.
template <int WPB, bool BIG>
__global__ __launch_bounds__(64 * WPB) void k_robust_disp(const int32_t* __restrict__ y, int ldn,
                                                          const double* __restrict__ sf,
                                                          const int32_t* __restrict__ cell_offsets,
                                                          const int32_t* __restrict__ cell_index, int n_cells,
                                                          int whole, int cap, int stride, int seg_len, int N, int G,
                                                          double* __restrict__ robust_disp,
                                                          const int32_t* __restrict__ list,
                                                          const int32_t* __restrict__ n_dev) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int w = threadIdx.x >> 6;
    const int k = blockIdx.x * WPB + w;
    if (n_dev != nullptr) G = min(G, *n_dev);  // second pass of the lean kernel: launched for a capacity
    if (k >= G) return;
    const int g = list != nullptr ? list[k] : k;
    double* scratch = lds + (size_t)w * stride;
    unsigned int* hist = (unsigned int*)(scratch + cap);
    CellPlan C{cell_offsets, cell_index, n_cells, whole};
    const double ar = robust_disp_gene<DeviceWave, BIG>(y + (size_t)g * ldn, sf, C, N, scratch, hist, LdsSorter(), seg_len);
    if ((threadIdx.x & 63) == 0) robust_disp[g] = ar;
}

// Designs whose cells all have at least kTrimBucketMin samples (or no cells at all: one trimmed variance over every
// sample - designs with continuous covariates): the bucket sums recompute the normalised counts from the gene's row on
// every pass instead of keeping them in LDS (robust_disp_gene_lean).  At N = 5000 the buffered kernel held 70 KB of LDS
// per wavefront - two wavefronts per CU, 13.3 ms for 60 000 genes; this one holds the 8 KB bucket table: 5.5 ms, 2.6 ms with the
// batched fetch of NormedValues (dsq_stats.h, fetch_batch).  A gene on
// which a bucket pass gives up (a boundary bucket with more than kBucketGather values, a non-finite value) is listed
// and redone by the buffered kernel.
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k_robust_disp_lean(const int32_t* __restrict__ y, int ldn,
                                                               const double* __restrict__ sf,
                                                               const int32_t* __restrict__ cell_offsets,
                                                               const int32_t* __restrict__ cell_index, int n_cells,
                                                               int whole, int N, int G,
                                                               double* __restrict__ robust_disp,
                                                               int32_t* __restrict__ redo_count,
                                                               int32_t* __restrict__ redo_list, int can_redo) {
    __shared__ BucketWork W[WPB];
    const int w = threadIdx.x >> 6;
    const int g = blockIdx.x * WPB + w;
    if (g >= G) return;
    CellPlan C{cell_offsets, cell_index, n_cells, whole};
    bool failed;
    const double ar = robust_disp_gene_lean<DeviceWave>(y + (size_t)g * ldn, sf, C, N, W[w], failed);
    if ((threadIdx.x & 63) == 0) {
        // failed: a normalised count that is not finite (a size factor of 0 / inf / NaN).  The buffered kernel orders such
        // values as numpy.sort does; where a cell is too long for its LDS the gene's robust dispersion is NaN (can_redo == 0)
        if (failed && can_redo) redo_list[atomicAdd(redo_count, 1)] = g;
        else robust_disp[g] = failed ? NAN : ar;
    }
}

// ... where the buffer costs occupancy: from 2048 samples in the largest cell on (16 KB + 8 KB of LDS per wavefront).
// Measured (before the batched fetch): c5 (one "cell" of 5000 samples) 13.3 -> 5.5 ms per 60 000 genes; c3 (two cells of 500: 12 KB per wavefront
// either way) 0.84 -> 1.14 ms - the recomputation costs more than the buffer there, so c3 stays on the buffered kernel.
bool robust_disp_lean_eligible(int min_cell, int max_cell, int whole, int N) {
    const bool off = getenv("DSQ_NO_ROBUST_LEAN") != nullptr;  // A/B switch (read per launch: the tests flip it)
    const char* mn = getenv("DSQ_ROBUST_LEAN_MIN");
    const int min_big = mn != nullptr ? atoi(mn) : 2048;
    return !off && (whole ? N : min_cell) >= kTrimBucketMin && (whole ? N : max_cell) >= min_big;
}

// redo: G + 1 int32 of device scratch (lean path only, may be null otherwise): [0] the number of genes the lean kernel
// handed back, [1 ..] their indices
hipError_t launch_robust_disp(hipStream_t st, const int32_t* y, int ldn, const double* sf,
                              const int32_t* cell_offsets, const int32_t* cell_index, int n_cells, int whole,
                              int max_cell, int N, int G, double* robust_disp, int min_cell, int32_t* redo) {
    if (G <= 0) return hipSuccess;
    const int biggest = whole ? N : max_cell;
    // designs whose cells all have at most kSegMaxCell samples: several cells per sorting pass (seg_trimmed_variances)
    static const bool seg_off = getenv("DSQ_NO_SEG_CELLS") != nullptr;  // A/B switch
    const int seg_len = (!whole && !seg_off && biggest <= kSegMaxCell) ? next_pow2(biggest < 2 ? 2 : biggest) : 0;
    const int cap = seg_len > 0 ? kSegBatch + kSegBatch / 2 : trim_cap(biggest);
    const int stride = cap + trim_work_doubles(biggest);
    const size_t per_wave = (size_t)stride * sizeof(double);
    const bool big = biggest >= kTrimBucketMin;
    // a cell of more than ~13 000 samples does not fit a wavefront's LDS: the buffer-less kernel takes the design whatever
    // its other cells look like (it selects where it cannot bucket), and there is no buffered second pass
    const bool buffered_fits = per_wave <= 160 * 1024;
    if (!buffered_fits && redo == nullptr) return hipErrorInvalidValue;
    const bool lean = redo != nullptr && (!buffered_fits || robust_disp_lean_eligible(min_cell, max_cell, whole, N));
    if (lean) {
        hipError_t e = hipMemsetAsync(redo, 0, sizeof(int32_t), st);
        if (e != hipSuccess) return e;
        constexpr int WPB = 4;
        hipLaunchKernelGGL((k_robust_disp_lean<WPB>), dim3((G + WPB - 1) / WPB), dim3(64 * WPB), 0, st, y, ldn, sf,
                           cell_offsets, cell_index, n_cells, whole, N, G, robust_disp, redo, redo + 1,
                           buffered_fits ? 1 : 0);
        e = hipGetLastError();
        if (e != hipSuccess || !buffered_fits) return e;
    }
    const int32_t* list = lean ? redo + 1 : nullptr;
    const int32_t* n_dev = lean ? redo : nullptr;
#define DSQ_RD_LAUNCH(WPB) \
    do { if (big) DSQ_RD_LAUNCH_(WPB, true); else DSQ_RD_LAUNCH_(WPB, false); } while (0)
#define DSQ_RD_LAUNCH_(WPB, BIG)                                                                             \
    do {                                                                                                     \
        if (per_wave * WPB > 48 * 1024) {                                                                    \
            (void)hipFuncSetAttribute((const void*)k_robust_disp<WPB, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(per_wave * WPB));                                                \
            (void)hipGetLastError();                                                                         \
        }                                                                                                    \
        hipLaunchKernelGGL((k_robust_disp<WPB, BIG>), dim3((G + WPB - 1) / WPB), dim3(64 * WPB), per_wave * WPB, st, y, \
                           ldn, sf, cell_offsets, cell_index, n_cells, whole, cap, stride, seg_len, N, G, robust_disp, \
                           list, n_dev);                                                                     \
    } while (0)
    if (per_wave * 4 <= 64 * 1024) DSQ_RD_LAUNCH(4);
    else if (per_wave * 2 <= 160 * 1024) DSQ_RD_LAUNCH(2);
    else DSQ_RD_LAUNCH(1);
#undef DSQ_RD_LAUNCH
#undef DSQ_RD_LAUNCH_
    return hipGetLastError();
}

// ------------------------------------------------------------------ outlier replacement
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k_replace(const int32_t* __restrict__ y,
                                                      const double* __restrict__ cooks, int ldn,
                                                      const double* __restrict__ sf,
                                                      const uint8_t* __restrict__ flags,
                                                      const int32_t* __restrict__ gene_idx, int n_sel,
                                                      int N, int cap, int stride, double cutoff,
                                                      int32_t* __restrict__ y_out,
                                                      uint8_t* __restrict__ all_zero, int cooks_ld,
                                                      const int32_t* __restrict__ slot_of) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int w = threadIdx.x >> 6;
    const int k = blockIdx.x * WPB + w;
    if (k >= n_sel) return;
    const int g = gene_idx[k];
    const int32_t* yr = y + (size_t)g * ldn;
    // (mixed designs: the Cook's layer is in slot order with its own pitch, see k_irls_mix)
    const double* ck = cooks + (size_t)g * (slot_of != nullptr ? (size_t)cooks_ld : (size_t)ldn);
    double* scratch = lds + (size_t)w * stride;
    const double tbm = trimmed_base_mean<DeviceWave>(yr, sf, N, 0.2, scratch, LdsSorter(),
                                                     stride > cap ? (BucketWork*)(scratch + cap) : nullptr);
    int nonzero = 0;
    for (int n = DeviceWave::lane(); n < N; n += 64) {
        int v = yr[n];
        if ((flags[n] & 2) && ck[slot_of != nullptr ? slot_of[n] : n] > cutoff) v = (int)(tbm * sf[n]);  // truncation (astype(int))
        y_out[(size_t)k * ldn + n] = v;
        nonzero |= (v != 0);
    }
    nonzero = DeviceWave::sumi(nonzero);
    if ((threadIdx.x & 63) == 0) all_zero[k] = (uint8_t)(nonzero == 0);
    // A gene whose counts all became zero leaves the refit (dds.py:1368-1383: LFC = 0, its row is dropped from the
    // sub-dataset).  The caller enqueues the refit stages for every row of the batch without waiting for these flags,
    // so such a row keeps its ORIGINAL counts - an ordinary gene for the kernels downstream; its results are discarded
    // by the flag.
    if (nonzero == 0)
        for (int n = DeviceWave::lane(); n < N; n += 64) y_out[(size_t)k * ldn + n] = yr[n];
}

// The same for rows too long for a wavefront's LDS (more than ~8 000 samples: next_pow2(N) doubles + the bucket table):
// the trimmed mean from the buffer-less routine (trimmed_base_mean_lean) - any number of samples.
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k_replace_lean(const int32_t* __restrict__ y,
                                                           const double* __restrict__ cooks, int ldn,
                                                           const double* __restrict__ sf,
                                                           const uint8_t* __restrict__ flags,
                                                           const int32_t* __restrict__ gene_idx, int n_sel, int N,
                                                           double cutoff, int32_t* __restrict__ y_out,
                                                           uint8_t* __restrict__ all_zero, int cooks_ld,
                                                           const int32_t* __restrict__ slot_of) {
    __shared__ BucketWork W[WPB];
    const int w = threadIdx.x >> 6;
    const int k = blockIdx.x * WPB + w;
    if (k >= n_sel) return;
    const int g = gene_idx[k];
    const int32_t* yr = y + (size_t)g * ldn;
    const double* ck = cooks + (size_t)g * (slot_of != nullptr ? (size_t)cooks_ld : (size_t)ldn);
    bool failed;
    const double tbm = trimmed_base_mean_lean<DeviceWave>(yr, sf, N, 0.2, W[w], failed);
    int nonzero = 0;
    for (int n = DeviceWave::lane(); n < N; n += 64) {
        int v = yr[n];
        if ((flags[n] & 2) && ck[slot_of != nullptr ? slot_of[n] : n] > cutoff) v = (int)(tbm * sf[n]);  // truncation (astype(int))
        y_out[(size_t)k * ldn + n] = v;
        nonzero |= (v != 0);
    }
    nonzero = DeviceWave::sumi(nonzero);
    if ((threadIdx.x & 63) == 0) all_zero[k] = (uint8_t)(nonzero == 0);
    if (nonzero == 0)  // (see k_replace)
        for (int n = DeviceWave::lane(); n < N; n += 64) y_out[(size_t)k * ldn + n] = yr[n];
}

hipError_t launch_replace(hipStream_t st, const int32_t* y, const double* cooks, int ldn,
                          const double* sf, const uint8_t* flags, const int32_t* gene_idx, int n_sel,
                          int N, double cutoff, int32_t* y_out, uint8_t* all_zero, int cooks_ld,
                          const int32_t* slot_of) {
    if (n_sel <= 0) return hipSuccess;
    const int cap = next_pow2(N);  // (the sort is the fallback of the bucket path: power-of-two room either way)
    const int stride = cap + trim_work_doubles(N);
    const size_t per_wave = (size_t)stride * sizeof(double);
    static const bool force_lean = getenv("DSQ_REPLACE_LEAN") != nullptr;  // A/B switch (tests: the buffer-less kernel on short rows)
    if (per_wave > 160 * 1024 || force_lean) {
        constexpr int WPB = 4;
        hipLaunchKernelGGL(k_replace_lean<WPB>, dim3((n_sel + WPB - 1) / WPB), dim3(64 * WPB), 0, st, y, cooks, ldn, sf,
                           flags, gene_idx, n_sel, N, cutoff, y_out, all_zero, cooks_ld, slot_of);
        return hipGetLastError();
    }
#define DSQ_REPL_LAUNCH(WPB)                                                                         \
    do {                                                                                             \
        if (per_wave * WPB > 48 * 1024) {                                                            \
            (void)hipFuncSetAttribute((const void*)k_replace<WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(per_wave * WPB));                                        \
            (void)hipGetLastError();                                                                 \
        }                                                                                            \
        hipLaunchKernelGGL(k_replace<WPB>, dim3((n_sel + WPB - 1) / WPB), dim3(64 * WPB),            \
                           per_wave * WPB, st, y, cooks, ldn, sf, flags, gene_idx, n_sel, N, cap,    \
                           stride, cutoff, y_out, all_zero, cooks_ld, slot_of);                      \
    } while (0)
    if (per_wave * 4 <= 64 * 1024) DSQ_REPL_LAUNCH(4);
    else if (per_wave * 2 <= 160 * 1024) DSQ_REPL_LAUNCH(2);
    else DSQ_REPL_LAUNCH(1);
#undef DSQ_REPL_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------ misc
template <class T>
__global__ __launch_bounds__(256) void k_gather_rows(const T* __restrict__ src, int ld,
                                                     const int32_t* __restrict__ idx, int n_idx,
                                                     int ncols, T* __restrict__ dst,
                                                     const int32_t* __restrict__ n_dev = nullptr) {
    const int k = blockIdx.x;  // rows (genes) on grid.x: grid.y is limited to 65535
    if (n_dev != nullptr && k >= *n_dev) return;
    const int g = idx[k];
    for (int c = blockIdx.y * 256 + threadIdx.x; c < ncols; c += gridDim.y * 256)
        dst[(size_t)k * ld + c] = src[(size_t)g * ld + c];
}

hipError_t launch_gather_rows_f64(hipStream_t st, const double* src, int ld, const int32_t* idx,
                                  int n_idx, int ncols, double* dst) {
    if (n_idx <= 0) return hipSuccess;
    const int gx = (ncols + 255) / 256 > 64 ? 64 : (ncols + 255) / 256;
    hipLaunchKernelGGL(k_gather_rows<double>, dim3(n_idx, gx), dim3(256), 0, st, src, ld, idx, n_idx, ncols,
                       dst);
    return hipGetLastError();
}

hipError_t launch_gather_rows_i32(hipStream_t st, const int32_t* src, int ld, const int32_t* idx,
                                  int n_idx, int ncols, int32_t* dst, const int32_t* n_dev) {
    if (n_idx <= 0) return hipSuccess;
    const int gx = (ncols + 255) / 256 > 64 ? 64 : (ncols + 255) / 256;
    hipLaunchKernelGGL(k_gather_rows<int32_t>, dim3(n_idx, gx), dim3(256), 0, st, src, ld, idx, n_idx,
                       ncols, dst, n_dev);
    return hipGetLastError();
}

// ---- O(G) glue between the per-gene stages (keeps the dispersion vectors on the device)
// fitted trend  a0 + a1 / normed_mean  (dds.py:826-833; a1 = 0 gives the mean trend, dds.py:1277-1299)
__global__ void k_trend_eval(const double* __restrict__ nm, int n, double a0, double a1,
                             double* __restrict__ fitted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fitted[i] = (a1 == 0.0) ? a0 : a0 + a1 / nm[i];
}

// the same with the coefficients where the trend-fit kernel left them on the device (coef[0], coef[1]): the fitted
// values and the prior can then be enqueued without the host seeing the coefficients first
__global__ void k_trend_eval_dev(const double* __restrict__ nm, int n, const double* __restrict__ coef,
                                 double* __restrict__ fitted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double a0 = coef[0], a1 = coef[1];
    if (i < n) fitted[i] = (a1 == 0.0) ? a0 : a0 + a1 / nm[i];
}

// final dispersions (dds.py:912-935): MAP value, except for dispersion outliers
// log(genewise) > log(fitted) + 2 sqrt(squared_logres), which keep the (clipped) genewise value
// the genewise and MAP dispersions are clipped IN PLACE (dds.py:792-794, 905-907: the reference stores them clipped):
// the host then takes the vectors as they come
__global__ void k_select_disp(double* __restrict__ gw_raw, double* __restrict__ map_raw,
                              const double* __restrict__ fitted, int n, double min_disp, double max_disp,
                              double two_sd, double* __restrict__ disp, uint8_t* __restrict__ outlier) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double gw = fmin(fmax(gw_raw[i], min_disp), max_disp);
    const double mp = fmin(fmax(map_raw[i], min_disp), max_disp);
    const bool out = log(gw) > log(fitted[i]) + two_sd;
    disp[i] = out ? gw : mp;
    outlier[i] = out ? 1 : 0;
    if (gw_raw[i] == gw_raw[i]) gw_raw[i] = gw;  // a NaN stays a NaN (numpy.clip), fmax / fmin would replace it
    if (map_raw[i] == map_raw[i]) map_raw[i] = mp;
}

// The same selection for one part of the genes (an LFC fit in two launches, dsq_dev_select_dispersions_part):
// mode 1: the genes whose MAP fit has finished AND converged in the stage's full-size launch (map_conv[i] == 1: the vector
//         was filled with 0xFF before the stage and only that launch writes it - the later launches of the stage write
//         conv_late, dsq_alpha_set_late_flags -, so what this kernel reads does not depend on how far those have come) -
//         part[i] = 1 for them, 0 for the rest;
// mode 0: the rest (part[i] == 0), once the continuation launch and the grid-search pass have run; their late flags move
//         into map_conv.
__global__ void k_select_disp_part(double* __restrict__ gw_raw, double* __restrict__ map_raw,
                                   const double* __restrict__ fitted, int n, double min_disp, double max_disp,
                                   double two_sd, double* __restrict__ disp, uint8_t* __restrict__ outlier,
                                   uint8_t* __restrict__ map_conv, const uint8_t* __restrict__ conv_late,
                                   uint8_t* __restrict__ part, int mode, int ready_limit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mode == 1) {
        // (genes from ready_limit on wait for the second launch whatever their fit: their robust dispersions - read by the
        // LFC fit's epilogue - are computed beside the first one)
        const bool ready = map_conv[i] == 1 && i < ready_limit;
        part[i] = ready ? 1 : 0;
        if (!ready) return;
    } else {
        if (part[i] != 0) return;
        if (conv_late != nullptr && conv_late[i] != 0xFF) map_conv[i] = conv_late[i];
    }
    const double gw = fmin(fmax(gw_raw[i], min_disp), max_disp);
    const double mp = fmin(fmax(map_raw[i], min_disp), max_disp);
    const bool out = log(gw) > log(fitted[i]) + two_sd;
    disp[i] = out ? gw : mp;
    outlier[i] = out ? 1 : 0;
    if (gw_raw[i] == gw_raw[i]) gw_raw[i] = gw;
    if (map_raw[i] == map_raw[i]) map_raw[i] = mp;
}

// dst[idx[k]][0..width) = src[k][0..width)   (results of the outlier refit back into the full vectors)
__global__ void k_scatter_rows(const double* __restrict__ src, const int32_t* __restrict__ idx, int n_idx,
                               int width, double* __restrict__ dst, const int32_t* __restrict__ n_dev) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev != nullptr) n_idx = min(n_idx, *n_dev);
    if (t >= n_idx * width) return;
    const int k = t / width, c = t % width;
    dst[(size_t)idx[k] * width + c] = src[t];
}

// variance stabilising transformation of the normalised counts (dds.py:486-514), sample-major N x G:
// mode 0: log2((1 + a1 + 2 a0 x + 2 sqrt(a0 x (1 + a1 + a0 x))) / (4 a0)),  x = count / size factor
// mode 1: (2 asinh(sqrt(a0 x)) - log a0 - log 4) / log 2                    (a0 = mean dispersion)
template <class T>
__global__ void k_vst(const T* __restrict__ counts, int N, int G, const double* __restrict__ sf, int mode,
                      double a0, double a1, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * G) return;
    const double x = (double)counts[i] / sf[i / G];
    double v;
    if (mode == 0) v = log2((1.0 + a1 + 2.0 * a0 * x + 2.0 * sqrt(a0 * x * (1.0 + a1 + a0 * x))) / (4.0 * a0));
    else v = (2.0 * asinh(sqrt(a0 * x)) - log(a0) - log(4.0)) / log(2.0);
    out[i] = v;
}

// ---- iterative size factors (dds.py:1460-1548): per-gene NLL under rescaled size factors
// cst[g] = N a log(alpha) - sum_n [ lgamma(y+a) - lgamma(y+1) - lgamma(a) ]   (alpha-only part, utils.py:216-226)
__global__ __launch_bounds__(kBlock) void k_nll_const(const int32_t* __restrict__ y, int ldn, int N, int G,
                                                      const double* __restrict__ disp, double* __restrict__ cst) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    const double alpha = disp[g], a = 1.0 / alpha;
    const int32_t* yr = y + (size_t)g * ldn;
    double s = 0.0;
    for (int n = threadIdx.x & 63; n < N; n += 64) {
        const double yv = (double)yr[n];
        s += lgamma_pos(yv + a) - lgamma_pos(yv + 1.0);
    }
    s = DeviceWave::sum(s);
    if ((threadIdx.x & 63) == 0) cst[g] = (double)N * a * log(alpha) - (s - (double)N * lgamma_pos(a));
}

// nll[g] = cst[g] + sum_n (y + a) log(mu scale_n + a) - y log(mu scale_n)
__global__ __launch_bounds__(kBlock) void k_nll_scaled(const int32_t* __restrict__ y, const double* __restrict__ mu,
                                                       int ldn, int N, int G, const double* __restrict__ disp,
                                                       const double* __restrict__ scale,
                                                       const double* __restrict__ cst, double* __restrict__ nll) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    const double a = 1.0 / disp[g];
    const int32_t* yr = y + (size_t)g * ldn;
    const double* mr = mu + (size_t)g * ldn;
    double s = 0.0;
    for (int n = threadIdx.x & 63; n < N; n += 64) {
        const double yv = (double)yr[n], m = mr[n] * scale[n];
        s += (yv + a) * log(m + a) - yv * log(m);
    }
    s = DeviceWave::sum(s);
    if ((threadIdx.x & 63) == 0) nll[g] = cst[g] + s;
}

hipError_t launch_nll_const(hipStream_t st, const int32_t* y, int ldn, int N, int G, const double* disp, double* cst) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_nll_const, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, ldn, N, G, disp, cst);
    return hipGetLastError();
}
hipError_t launch_nll_scaled(hipStream_t st, const int32_t* y, const double* mu, int ldn, int N, int G,
                             const double* disp, const double* scale, const double* cst, double* nll) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_nll_scaled, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, mu, ldn, N, G, disp, scale, cst,
                       nll);
    return hipGetLastError();
}

hipError_t launch_vst(hipStream_t st, const void* counts_sm, int count_type, int N, int G, const double* sf, int mode,
                      double a0, double a1, double* out) {
    if (N <= 0 || G <= 0) return hipSuccess;
    const size_t total = (size_t)N * G;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (count_type == 1)
        hipLaunchKernelGGL(k_vst<int64_t>, grid, block, 0, st, (const int64_t*)counts_sm, N, G, sf, mode, a0, a1, out);
    else
        hipLaunchKernelGGL(k_vst<int32_t>, grid, block, 0, st, (const int32_t*)counts_sm, N, G, sf, mode, a0, a1, out);
    return hipGetLastError();
}

hipError_t launch_trend_eval(hipStream_t st, const double* nm, int n, double a0, double a1, double* fitted) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_trend_eval, dim3((n + 255) / 256), dim3(256), 0, st, nm, n, a0, a1, fitted);
    return hipGetLastError();
}
hipError_t launch_trend_eval_dev(hipStream_t st, const double* nm, int n, const double* coef, double* fitted) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_trend_eval_dev, dim3((n + 255) / 256), dim3(256), 0, st, nm, n, coef, fitted);
    return hipGetLastError();
}
hipError_t launch_select_disp(hipStream_t st, double* gw_raw, double* map_raw, const double* fitted,
                              int n, double min_disp, double max_disp, double two_sd, double* disp,
                              uint8_t* outlier) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_select_disp, dim3((n + 255) / 256), dim3(256), 0, st, gw_raw, map_raw, fitted, n,
                       min_disp, max_disp, two_sd, disp, outlier);
    return hipGetLastError();
}
hipError_t launch_select_disp_part(hipStream_t st, double* gw_raw, double* map_raw, const double* fitted, int n,
                                   double min_disp, double max_disp, double two_sd, double* disp, uint8_t* outlier,
                                   uint8_t* map_conv, const uint8_t* conv_late, uint8_t* part, int mode,
                                   int ready_limit) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_select_disp_part, dim3((n + 255) / 256), dim3(256), 0, st, gw_raw, map_raw, fitted, n, min_disp,
                       max_disp, two_sd, disp, outlier, map_conv, conv_late, part, mode, ready_limit);
    return hipGetLastError();
}
hipError_t launch_scatter_rows(hipStream_t st, const double* src, const int32_t* idx, int n_idx, int width,
                               double* dst, const int32_t* n_dev) {
    if (n_idx <= 0 || width <= 0) return hipSuccess;
    const int total = n_idx * width;
    hipLaunchKernelGGL(k_scatter_rows, dim3((total + 255) / 256), dim3(256), 0, st, src, idx, n_idx, width, dst, n_dev);
    return hipGetLastError();
}

// gamma-GLM trend loss/gradient partial sums, one row of 4 per block (summed by the host in a
// fixed order => run-to-run deterministic): {sum(t/m + log m), sum g0, sum g1, count}
constexpr int kTrendBlocks = 256;
__global__ __launch_bounds__(256) void k_trend(const double* __restrict__ cov,
                                               const double* __restrict__ targets,
                                               const uint8_t* __restrict__ keep, int n, double a0,
                                               double a1, double* __restrict__ partials) {
    __shared__ double red[4][4];
    double s = 0.0, g0 = 0.0, g1 = 0.0, cnt = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (keep != nullptr && keep[i] == 0) continue;
        const double c = cov[i], t = targets[i];
        const double m = a0 + a1 * c;
        const double v = t / m + log(m);
        if (v != v) continue;  // np.nanmean skips NaN terms
        s += v;
        const double r = (t / m - 1.0) / m;
        g0 += r;
        g1 += r * c;
        cnt += 1.0;
    }
    s = DeviceWave::sum(s); g0 = DeviceWave::sum(g0); g1 = DeviceWave::sum(g1); cnt = DeviceWave::sum(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = s; red[w][1] = g0; red[w][2] = g1; red[w][3] = cnt; }
    __syncthreads();
    if (threadIdx.x < 4) {
        partials[blockIdx.x * 4 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}

// partials: [kTrendBlocks][4] device doubles
hipError_t launch_trend_loss_grad(hipStream_t st, const double* cov, const double* targets,
                                  const uint8_t* keep, int n, double a0, double a1, double* partials) {
    hipLaunchKernelGGL(k_trend, dim3(kTrendBlocks), dim3(256), 0, st, cov, targets, keep, n, a0, a1,
                       partials);
    return hipGetLastError();
}

// ------------------------------------------------------------------ trend fit
// The fit is a CHAIN of dependent data passes - one per L-BFGS-B evaluation (about 20 per gamma-GLM fit), one per
// outlier filter - over two doubles per gene: its duration is (number of passes) x (latency of a pass), nearly
// independent of the number of genes (r03: 0.39 ms at 7 500 genes, 0.47 ms at 60 000).  Everything here serves that
// latency:
//   - the genes live in REGISTERS: kTrendGridBlocks x 64 kTrendWaves threads hold kTrendRegGenes genes each (clipped
//     dispersion, 1 / mean, keep bit), loaded once by the first pass; a pass reads no memory (genes beyond that
//     capacity - more than 65 536 - go through the arrays, as before);
//   - one cross-lane butterfly per reduction level for the six sums of a pass (loss, two gradient sums and their three
//     NaN-free counts, carried as doubles): lanes -> wave, the waves of a workgroup (through LDS), the workgroups;
//   - EVERY workgroup's wave 0 runs the same deterministic optimiser on the same totals (bit-identical: fixed butterfly
//     order), so nothing is broadcast, and the workgroups exchange their partial sums through SELF-VALIDATING words:
//     every 64-bit word carries 32 bits of payload and the pass number.  A reader polls the words of all workgroups
//     until every tag is the current pass: one store and one (polled) load per pass and workgroup - no arrival counter,
//     no fence, no second round trip for the data (the arrival-counter barrier this replaces cost three dependent
//     memory round trips per pass).  Words are double-buffered by pass parity: a workgroup can be at most one pass
//     ahead of the slowest reader (it needs that reader's words of pass p to leave pass p).
// Launched cooperatively (all workgroups resident); a bounded poll sets a flag instead of hanging the GPU.
#ifndef DSQ_TREND_WAVES
#define DSQ_TREND_WAVES 8
#endif
constexpr int kTrendWaves = DSQ_TREND_WAVES;
#ifndef DSQ_TREND_GRID_BLOCKS
#define DSQ_TREND_GRID_BLOCKS 32
#endif
constexpr int kTrendGridBlocks = DSQ_TREND_GRID_BLOCKS;  // <= 64: one lane of the leader wave per workgroup
constexpr int kTrendRegGenes = 4;
constexpr int kTrendWords = 12;  // six doubles, two tagged words each

struct TrendGridMem {  // device global memory, zeroed before every launch (pass numbers start at 1)
    unsigned int timeout;
    unsigned int pad[31];
    unsigned long long word[2][64][16];  // [pass parity][workgroup][word]
};

struct TrendShared {
    TrendWork W;
    double a0, a1;  // mailbox: leader wave -> helper waves of this workgroup
    int cmd;        // 1 eval, 2 filter, 3 load + initial mask, 0 done
    double part[kTrendWaves][6];
};

// loss / gradient terms of one gene (trend_eval_partial's arithmetic, dsq_trend.h)
__device__ __forceinline__ void trend_eval_one(double cov, double t, double a0, double a1, TrendPartial& P) {
    const double mu = a0 + a1 * cov;
    const double rmu = frcp(mu);
    const double tm = t * rmu;
    const double v = tm + flog(mu);
    if (v == v) { P.s.add(v); P.cf += 1; }
    const double r = tm - 1.0;
    const double v0 = r * rmu, v1 = (r * cov) * rmu;
    if (v0 == v0) { P.g0.add(v0); P.c0 += 1; }
    if (v1 == v1) { P.g1.add(v1); P.c1 += 1; }
}

struct GridTrendOps {
    TrendData D;
    TrendGridMem* Gm;
    TrendShared* S;  // LDS of this workgroup
    unsigned int seq = 0;
    // this thread's genes (i = tid + k * NT)
    double cov_e[kTrendRegGenes], cov_f[kTrendRegGenes], tg[kTrendRegGenes];
    unsigned int kbits = 0;
#ifdef DSQ_TREND_PHASES
    long long t_last = 0, c_opt = 0, c_pass = 0;
    int n_eval = 0;
#endif

    // every thread of every workgroup, once per pass: this thread's share -> wave sums -> LDS
    __device__ void work(int cmd, double a0, double a1) {
        const int w = threadIdx.x >> 6;
        const int tid = blockIdx.x * (64 * kTrendWaves) + threadIdx.x, NT = gridDim.x * 64 * kTrendWaves;
        double v[6];
        if (cmd == 1) {
            TrendPartial P;
#pragma unroll
            for (int k = 0; k < kTrendRegGenes; ++k)
                if (kbits & (1u << k)) trend_eval_one(cov_e[k], tg[k], a0, a1, P);
            trend_eval_partial(D, tid + kTrendRegGenes * NT, NT, a0, a1, P);
            v[0] = P.s.value(); v[1] = P.g0.value(); v[2] = P.g1.value();
            v[3] = (double)P.cf; v[4] = (double)P.c0; v[5] = (double)P.c1;
        } else {
            int kept = 0;
            if (cmd == 3) {
#pragma unroll
                for (int k = 0; k < kTrendRegGenes; ++k) {
                    const int i = tid + k * NT;
                    cov_e[k] = 0.0; cov_f[k] = 0.0; tg[k] = 0.0;
                    if (i < D.n) {
                        const double m = D.means[i], d = D.disp[i];
                        const double c = D.raw ? 0.0 : 1.0 / m;
                        const bool bad = (c != c) || (c == INFINITY) || (c == -INFINITY);  // dds.py:1225-1231
                        cov_e[k] = D.raw ? m : frcp(m);
                        cov_f[k] = c;
                        tg[k] = D.raw ? d : dmin(dmax(d, D.min_disp), D.max_disp);
                        if (!bad) { kbits |= 1u << k; kept += 1; }
                    }
                }
                kept += trend_init_keep(D, tid + kTrendRegGenes * NT, NT);
            } else {
#pragma unroll
                for (int k = 0; k < kTrendRegGenes; ++k)
                    if (kbits & (1u << k)) {
                        const double ratio = tg[k] / (a0 + a1 * cov_f[k]);
                        if (ratio < 1e-4 || ratio >= 15.0) kbits &= ~(1u << k);  // dds.py:1254-1264
                        else kept += 1;
                    }
                kept += trend_filter(D, tid + kTrendRegGenes * NT, NT, a0, a1);
            }
            v[0] = (double)kept; v[1] = 0.0; v[2] = 0.0; v[3] = 0.0; v[4] = 0.0; v[5] = 0.0;
        }
        DeviceWave::sum_n<6>(v);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) S->part[w][q] = v[q];
        }
    }

    // leader wave: the workgroup's sums -> its words of this pass; poll everybody's; combine.  tot[] is identical in
    // every lane of every workgroup's leader wave.
    __device__ void exchange(double (&tot)[6]) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int q = 0; q < 6; ++q) tot[q] = (lane < kTrendWaves) ? S->part[lane][q] : 0.0;
        DeviceWave::sum_n<6>(tot);
        if (gridDim.x == 1) return;
        seq += 1;
        unsigned long long(*words)[16] = Gm->word[seq & 1];
        if (lane < kTrendWords) {
            double mine = tot[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) mine = (lane >> 1) == q ? tot[q] : mine;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
            const unsigned long long half = (lane & 1) ? (bits >> 32) : (bits & 0xffffffffull);
            __hip_atomic_store(&words[blockIdx.x][lane], (half << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool on = lane < (int)gridDim.x;
        unsigned long long wd[kTrendWords];
        unsigned int spins = 0;
        for (;;) {
            bool ok = true;
            if (on) {
#pragma unroll
                for (int j = 0; j < kTrendWords; ++j)
                    wd[j] = __hip_atomic_load(&words[lane][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < kTrendWords; ++j) ok = ok && ((unsigned int)wd[j] == seq);
            }
            if (__all(ok)) break;
            if (++spins > 20000000u) {
                __hip_atomic_store(&Gm->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const unsigned long long bits = (wd[2 * q] >> 32) | (wd[2 * q + 1] & 0xffffffff00000000ull);
            tot[q] = on ? __longlong_as_double((long long)bits) : 0.0;
        }
        DeviceWave::sum_n<6>(tot);
    }

    __device__ void pass(int cmd, double a0, double a1, double (&tot)[6]) {  // wave 0 of every workgroup
        if ((threadIdx.x & 63) == 0) { S->a0 = a0; S->a1 = a1; S->cmd = cmd; }
        __syncthreads();  // A: the mailbox is visible to the helper waves
        work(cmd, a0, a1);
        __syncthreads();  // B: the waves' sums are in LDS
        exchange(tot);
    }
    __device__ int init_keep() {
        double tot[6];
        pass(3, 0.0, 0.0, tot);
        return (int)tot[0];
    }
    __device__ void eval(double a0, double a1, double& f, double* g) {
#ifdef DSQ_TREND_PHASES
        const long long t0 = clock64();
        if (t_last) c_opt += t0 - t_last;
#endif
        double tot[6];
        pass(1, a0, a1, tot);
        f = fdiv(tot[0], tot[3]);
        g[0] = -fdiv(tot[1], tot[4]);
        g[1] = -fdiv(tot[2], tot[5]);
#ifdef DSQ_TREND_PHASES
        t_last = clock64();
        c_pass += t_last - t0;
        n_eval += 1;
#endif
    }
    __device__ int filter(double a0, double a1) {
        double tot[6];
        pass(2, a0, a1, tot);
        return (int)tot[0];
    }
};

__global__ __launch_bounds__(64 * kTrendWaves) void k_trend_fit(const double* __restrict__ disp,
                                                                const double* __restrict__ means, int n,
                                                                double min_disp, double max_disp,
                                                                uint8_t* __restrict__ keep, TrendGridMem* Gm,
                                                                double* __restrict__ out5, int single) {
    __shared__ TrendShared S;
    GridTrendOps ops;
    ops.D = TrendData{disp, means, keep, n, min_disp, max_disp, single};
    ops.Gm = Gm;
    ops.S = &S;
    if ((threadIdx.x >> 6) == 0) {  // the leader wave of every workgroup runs the optimiser
        const TrendOut o = trend_fit_core(ops, S.W, single != 0);
        if (threadIdx.x == 0) {
            if (blockIdx.x == 0) {
                const bool timed_out =
                    gridDim.x > 1 && __hip_atomic_load(&Gm->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                out5[0] = o.a0; out5[1] = o.a1; out5[2] = timed_out ? -1.0 : (double)o.ok;
                out5[3] = (double)o.n_outer; out5[4] = (double)o.n_kept;
#ifdef DSQ_TREND_PHASES
                printf("trend phases: evals %d  optimiser %lld  pass %lld cycles | build_B %lld cauchy %lld subsm %lld "
                       "ls-setup %lld fg %lld after-ls+update %lld\n", ops.n_eval, ops.c_opt, ops.c_pass, g_lbd_phase[0],
                       g_lbd_phase[1], g_lbd_phase[2], g_lbd_phase[3], g_lbd_phase[4], g_lbd_phase[5]);
                for (int q = 0; q < 8; ++q) g_lbd_phase[q] = 0;
#endif
            }
            S.cmd = 0;
        }
        __syncthreads();  // release this workgroup's helper waves
    } else {
        for (;;) {
            __syncthreads();  // A: wait for the leader's next command
            const int cmd = S.cmd;
            if (cmd == 0) break;
            ops.work(cmd, S.a0, S.a1);
            __syncthreads();  // B
        }
    }
}

size_t trend_grid_mem_bytes() { return sizeof(TrendGridMem); }

// out5: {a0, a1, ok (1 converged, 0 not, -1 the workgroups' exchange timed out), gamma-GLM fits, genes in the last fit}
hipError_t launch_trend_fit(hipStream_t st, const double* disp, const double* means, int n, double min_disp,
                            double max_disp, uint8_t* keep, double* out5, void* grid_mem, int force_grid) {
    TrendGridMem* gm = (TrendGridMem*)grid_mem;
    int single = 0;
    if (gm != nullptr && force_grid >= 0 && (force_grid > 0 || n >= 3072)) {  // (measured crossover of one workgroup against the grid: ~3000 genes)
        hipError_t e0 = hipMemsetAsync(gm, 0, sizeof(TrendGridMem), st);
        if (e0 != hipSuccess) return e0;
        void* args[] = {(void*)&disp, (void*)&means, (void*)&n,    (void*)&min_disp, (void*)&max_disp,
                        (void*)&keep, (void*)&gm,    (void*)&out5, (void*)&single};
        return hipLaunchCooperativeKernel((const void*)k_trend_fit, dim3(kTrendGridBlocks), dim3(64 * kTrendWaves), args,
                                          0, st);
    }
    hipLaunchKernelGGL(k_trend_fit, dim3(1), dim3(64 * kTrendWaves), 0, st, disp, means, n, min_disp, max_disp, keep,
                       (TrendGridMem*)nullptr, out5, 0);
    return hipGetLastError();
}

// Inference.dispersion_trend_gamma_glm (inference.py:284-308): ONE gamma-GLM fit of targets ~ a0 + a1 * cov
hipError_t launch_trend_glm(hipStream_t st, const double* targets, const double* cov, int n, uint8_t* keep,
                            double* out5) {
    hipLaunchKernelGGL(k_trend_fit, dim3(1), dim3(64 * kTrendWaves), 0, st, targets, cov, n, 0.0, 0.0, keep,
                       (TrendGridMem*)nullptr, out5, 1);
    return hipGetLastError();
}

// ------------------------------------------------------------------ distributed size factors
// Multi-GPU layout: every rank owns a gene shard, but a sample's size factor is the median over
// ALL genes.  The radix select of k_row_median is therefore split into per-pass kernels whose
// per-sample 256-bin digit histograms are summed across ranks (RCCL all-reduce) between passes.
// state (per sample): prefix[2] (u64), rank[2] (u32) for the two middle order statistics.
struct SfState {
    unsigned long long* prefix;  // [2][N]
    unsigned int* rank;          // [2][N]
};

__global__ __launch_bounds__(256) void k_sf_count(const unsigned long long* __restrict__ keys, int N, int G,
                                                  unsigned int* __restrict__ counts) {
    __shared__ unsigned int s;
    const int n = blockIdx.x;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    unsigned int c = 0;
    for (int g = threadIdx.x; g < G; g += 256) c += (keys[(size_t)n * G + g] != ~0ull) ? 1u : 0u;
    atomicAdd(&s, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[n] = s;
}

__global__ void k_sf_init(const unsigned int* __restrict__ total, int N, unsigned long long* prefix,
                          unsigned int* rank) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const unsigned int M = total[n];
    prefix[n] = 0ull; prefix[N + n] = 0ull;
    rank[n] = M ? (M - 1) / 2 : 0;
    rank[N + n] = M / 2;
}

// hist[which][n][256]: digit (key >> shift) & 255 of the keys matching prefix[which][n] above it
__global__ __launch_bounds__(1024) void k_sf_hist(const unsigned long long* __restrict__ keys, int N, int G,
                                                  const unsigned long long* __restrict__ prefix, int shift,
                                                  unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[2][256];
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < 512; i += 1024) (&h[0][0])[i] = 0;
    __syncthreads();
    const unsigned long long p0 = prefix[n], p1 = prefix[N + n];
    const unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
    const unsigned long long* row = keys + (size_t)n * G;
    for (int g = threadIdx.x; g < G; g += 1024) {
        const unsigned long long k = row[g];
        if (k == ~0ull) continue;
        const unsigned int d = (unsigned int)(k >> shift) & 0xff;
        if ((k & himask) == p0) atomicAdd(&h[0][d], 1u);
        if ((k & himask) == p1) atomicAdd(&h[1][d], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 1024)
        hist[((size_t)(i >> 8) * N + n) * 256 + (i & 255)] = (&h[0][0])[i];
}

__global__ void k_sf_pick(const unsigned int* __restrict__ hist, int N, int shift, unsigned long long* prefix,
                          unsigned int* rank) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // which * N + n
    if (idx >= 2 * N) return;
    const unsigned int* h = hist + (size_t)idx * 256;
    unsigned int r = rank[idx], acc = 0;
    int d = 0;
    for (; d < 255; ++d) {
        if (acc + h[d] > r) break;
        acc += h[d];
    }
    rank[idx] = r - acc;
    prefix[idx] |= ((unsigned long long)d << shift);
}

__global__ void k_sf_finish(const unsigned long long* __restrict__ prefix, const unsigned int* __restrict__ total,
                            int N, double* __restrict__ sf) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const unsigned int M = total[n];
    if (M == 0) { sf[n] = NAN; return; }
    const double v0 = key_f64(prefix[n]), v1 = key_f64(prefix[N + n]);
    const double med = ((M - 1) / 2 == M / 2) ? v0 : (v0 + v1) / 2.0;
    sf[n] = exp(med);
}

hipError_t launch_sf_keys(hipStream_t st, const void* counts_sm, int count_type, int N, int G,
                          const double* logmeans, const uint8_t* gene_mask, unsigned long long* keys) {
    const int gx = (G + 255) / 256 > 256 ? 256 : (G + 255) / 256;
    if (count_type == 1)
        hipLaunchKernelGGL((k_ratio_keys<int64_t>), dim3(gx, N), dim3(256), 0, st, (const int64_t*)counts_sm, N,
                           G, logmeans, gene_mask, keys);
    else
        hipLaunchKernelGGL((k_ratio_keys<int32_t>), dim3(gx, N), dim3(256), 0, st, (const int32_t*)counts_sm, N,
                           G, logmeans, gene_mask, keys);
    return hipGetLastError();
}
// compacted keys for the distributed protocol: idx_work = G + 2 ints; keys [N][*count]
hipError_t launch_sf_compact(hipStream_t st, const double* logmeans, const uint8_t* gene_mask, int G, int* idx_work) {
    hipError_t e0 = hipMemsetAsync(idx_work + G, 0, sizeof(int), st);
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL(k_sf_compact, dim3((G + 255) / 256), dim3(256), 0, st, logmeans, gene_mask, G, idx_work,
                       idx_work + G);
    return hipGetLastError();
}
hipError_t launch_sf_keys_compact(hipStream_t st, const void* counts_sm, int count_type, int N, int G,
                                  const double* logmeans, const int* idx_work, unsigned long long* keys) {
    const int gx = (G + 255) / 256 > 256 ? 256 : (G + 255) / 256;
    if (count_type == 1)
        hipLaunchKernelGGL((k_ratio_keys_c<int64_t>), dim3(gx, N), dim3(256), 0, st, (const int64_t*)counts_sm, N, G,
                           logmeans, idx_work, idx_work + G, keys, 0, 0);
    else
        hipLaunchKernelGGL((k_ratio_keys_c<int32_t>), dim3(gx, N), dim3(256), 0, st, (const int32_t*)counts_sm, N, G,
                           logmeans, idx_work, idx_work + G, keys, 0, 0);
    return hipGetLastError();
}
hipError_t launch_sf_count(hipStream_t st, const unsigned long long* keys, int N, int G, unsigned int* counts) {
    hipLaunchKernelGGL(k_sf_count, dim3(N), dim3(256), 0, st, keys, N, G, counts);
    return hipGetLastError();
}
hipError_t launch_sf_init(hipStream_t st, const unsigned int* total, int N, unsigned long long* prefix,
                          unsigned int* rank) {
    hipLaunchKernelGGL(k_sf_init, dim3((N + 255) / 256), dim3(256), 0, st, total, N, prefix, rank);
    return hipGetLastError();
}
hipError_t launch_sf_hist(hipStream_t st, const unsigned long long* keys, int N, int G,
                          const unsigned long long* prefix, int shift, unsigned int* hist) {
    hipLaunchKernelGGL(k_sf_hist, dim3(N), dim3(1024), 0, st, keys, N, G, prefix, shift, hist);
    return hipGetLastError();
}
hipError_t launch_sf_pick(hipStream_t st, const unsigned int* hist, int N, int shift, unsigned long long* prefix,
                          unsigned int* rank) {
    hipLaunchKernelGGL(k_sf_pick, dim3((2 * N + 255) / 256), dim3(256), 0, st, hist, N, shift, prefix, rank);
    return hipGetLastError();
}
hipError_t launch_sf_finish(hipStream_t st, const unsigned long long* prefix, const unsigned int* total, int N,
                            double* sf) {
    hipLaunchKernelGGL(k_sf_finish, dim3((N + 255) / 256), dim3(256), 0, st, prefix, total, N, sf);
    return hipGetLastError();
}

__global__ void k_log_vec(const double* __restrict__ in, int n, double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = log(in[i]);
}
hipError_t launch_log_vec(hipStream_t st, const double* in, int n, double* out) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_log_vec, dim3((n + 255) / 256), dim3(256), 0, st, in, n, out);
    return hipGetLastError();
}

// ---- mixed designs (dsq_mix.h): slot-ordered copies of per-gene rows, so that the kernels of that family stream contiguous
// rows instead of gathering every sample through the slot permutation (a 64-lane trip of the gather touched 16-32 cache
// lines; 24.6 % of k_alpha_mix's cycles were staging)
constexpr int kMixSlotPad = 0xFFFF;  // padding slot; 0xFFFE: "a count of at least 65 534" (the gene is flagged big)
// counts [G][ldn] int32 (sample order) -> ys [G][Ns] uint16 (slot order), big[g] = 1 when a count does not fit
__global__ __launch_bounds__(256) void k_mix_counts_to_slots(const int32_t* __restrict__ y, int ldn,
                                                             const int32_t* __restrict__ perm, int Ns, int G,
                                                             uint16_t* __restrict__ ys, uint8_t* __restrict__ big) {
    const int g = blockIdx.x;
    if (g >= G) return;
    const int32_t* row = y + (size_t)g * ldn;
    uint16_t* dst = ys + (size_t)g * Ns;
    int any_big = 0;
    for (int s = threadIdx.x; s < Ns; s += 256) {
        const int p = perm[s];
        int v = kMixSlotPad;
        if (p >= 0) {
            v = row[p];
            if (v >= 0xFFFE) { v = 0xFFFE; any_big = 1; }
        }
        dst[s] = (uint16_t)v;
    }
    any_big = __syncthreads_or(any_big);
    if (threadIdx.x == 0 && big != nullptr) big[g] = (uint8_t)(any_big ? 1 : 0);
}
// fp64 rows [G][ldn] (sample order) -> [G][Ns] (slot order), 0 in padding slots
__global__ __launch_bounds__(256) void k_mix_f64_to_slots(const double* __restrict__ m, int ldn,
                                                          const int32_t* __restrict__ perm, int Ns, int G,
                                                          double* __restrict__ ms) {
    const int g = blockIdx.x;
    if (g >= G) return;
    const double* row = m + (size_t)g * ldn;
    double* dst = ms + (size_t)g * Ns;
    for (int s = threadIdx.x; s < Ns; s += 256) {
        const int p = perm[s];
        dst[s] = p >= 0 ? row[p] : 0.0;
    }
}
// mu_hat of the IRLS route in slot order: mu[g][s] = sf * exp(x_c . beta + z . beta_z), UNclamped (dds.py:757-771,
// utils.py:435-437), 0 in padding slots.  One workgroup per gene; written once, read by both dispersion fits (the kernels
// used to rebuild it - an exponential per sample - in every launch and every continuation launch).
__global__ __launch_bounds__(256) void k_mix_mu_slots(const double* __restrict__ beta, const double* __restrict__ sf,
                                                      const MixDesign D, int G, double* __restrict__ mu) {
    __shared__ double cellv[kMixMaxCells];
    __shared__ double bz[kMixMaxQ];
    const int g = blockIdx.x;
    if (g >= G) return;
    const double* b = beta + (size_t)g * D.P;
    if (threadIdx.x < kMixMaxCells) {
        double e = 0.0;
        if ((int)threadIdx.x < D.C)
            for (int j = 0; j < D.P; ++j) e += D.Xc[threadIdx.x * D.P + j] * b[j];
        cellv[threadIdx.x] = e;
    }
    if ((int)threadIdx.x < D.Q) bz[threadIdx.x] = b[D.zcol[threadIdx.x]];
    __syncthreads();
    double* dst = mu + (size_t)g * D.Ns;
    for (int s = threadIdx.x; s < D.Ns; s += 256) {
        const int p = D.perm[s];
        double eta = cellv[D.trip_cell[s >> 6]];
        for (int q = 0; q < D.Q; ++q) eta += D.Zs[(size_t)q * D.Ns + s] * bz[q];
        dst[s] = p >= 0 ? sf[p] * exp(eta) : 0.0;
    }
}
hipError_t launch_mix_counts_to_slots(hipStream_t st, const int32_t* y, int ldn, const MixDesign& D, int G, uint16_t* ys,
                                      uint8_t* big) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mix_counts_to_slots, dim3(G), dim3(256), 0, st, y, ldn, D.perm, D.Ns, G, ys, big);
    return hipGetLastError();
}
hipError_t launch_mix_f64_to_slots(hipStream_t st, const double* m, int ldn, const MixDesign& D, int G, double* ms) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mix_f64_to_slots, dim3(G), dim3(256), 0, st, m, ldn, D.perm, D.Ns, G, ms);
    return hipGetLastError();
}
hipError_t launch_mix_mu_slots(hipStream_t st, const double* beta, const double* sf, const MixDesign& D, int G, double* mu) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mix_mu_slots, dim3(G), dim3(256), 0, st, beta, sf, D, G, mu);
    return hipGetLastError();
}

// Gene-sharded trend exchange: ONE all-gather carries both per-gene vectors of a rank.
// pack:  send[0..len) = a[0..n) then NaN, send[len..2 len) = b[0..n) then NaN   (NaN = "no gene": the trend / prior kernels skip them)
// unzip: recv [world][2][len] -> a_all [world * len], b_all [world * len]
__global__ void k_pack2(const double* __restrict__ a, const double* __restrict__ b, int n, int len, double* __restrict__ send) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * len) return;
    const int k = i < len ? i : i - len;
    send[i] = k < n ? (i < len ? a[k] : b[k]) : __longlong_as_double(0x7ff8000000000000LL);
}
__global__ void k_unzip2(const double* __restrict__ recv, int world, int len, double* __restrict__ a_all, double* __restrict__ b_all) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= world * len) return;
    const int r = i / len, k = i - r * len;
    a_all[i] = recv[(size_t)r * 2 * len + k];
    b_all[i] = recv[(size_t)r * 2 * len + len + k];
}
hipError_t launch_pack2(hipStream_t st, const double* a, const double* b, int n, int len, double* send) {
    if (len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack2, dim3((2 * len + 255) / 256), dim3(256), 0, st, a, b, n, len, send);
    return hipGetLastError();
}
hipError_t launch_unzip2(hipStream_t st, const double* recv, int world, int len, double* a_all, double* b_all) {
    if (world * len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_unzip2, dim3((world * len + 255) / 256), dim3(256), 0, st, recv, world, len, a_all, b_all);
    return hipGetLastError();
}

}  // namespace dsq
