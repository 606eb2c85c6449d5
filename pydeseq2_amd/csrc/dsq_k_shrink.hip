// dsq_k_shrink.hip — apeGLM MAP log-fold-change kernel (gfx950): one gene per wavefront, the n-dim
// L-BFGS-B workspace (12-15 KB) in wave-private LDS.  Algorithmic HBM traffic per gene: 4N bytes of
// counts per objective evaluation (re-reads hit L2), 8 p + 8 p^2 + 1 bytes written.
#include <type_traits>

#include "dsq_dispatch.h"
#include "dsq_launch.h"
#include "dsq_shrink.h"

namespace dsq {

// resident wavefronts per SIMD asked of the compiler (0: whatever the registers allow): the optimiser between two
// objective evaluations is serial code on an LDS workspace, i.e. LDS latency that only other wavefronts can cover
#ifndef DSQ_SHRINK_WAVES
#define DSQ_SHRINK_WAVES 0
#endif
template <int P, bool ALT = false>
__global__ __launch_bounds__(kBlock, DSQ_SHRINK_WAVES > 0 ? DSQ_SHRINK_WAVES : 1) void k_shrink(const int32_t* __restrict__ y, int ldn,
                                                   const double* __restrict__ offset,
                                                   const double* __restrict__ Xt, int ldx, int N, int G,
                                                   const double* __restrict__ size, double sigma0, double sigma,
                                                   int shrink_index, double* __restrict__ beta,
                                                   double* __restrict__ invh, uint8_t* __restrict__ conv,
                                                   double* __restrict__ ih_entry, int optimizer) {
    // ALT: optimizer "BFGS" / "Newton-CG" (their own instantiation: the default's registers and LDS are not theirs to grow)
#ifdef DSQ_SHRINK_COMPACT  // A/B build: the compact-form optimiser for every p > 4
    typedef typename std::conditional<ALT, ShrinkWorkAlt<P>, ShrinkWork<P>>::type WorkT;
#else
    typedef typename std::conditional<ALT, ShrinkWorkAlt<P>, ShrinkWork<P, shrink_on_wave8(P)>>::type WorkT;
#endif
    __shared__ WorkT work[kWavesPerBlock];
    const int w = threadIdx.x >> 6;
    const int g = blockIdx.x * kWavesPerBlock + w;
    if (g >= G) return;
    ShrinkArgs A;
    A.y = y + (size_t)g * ldn; A.offset = offset; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.size = size[g]; A.sigma0 = sigma0; A.sigma = sigma; A.shrink_index = shrink_index;
    double b[P];
    const int ok = shrink_gene<DeviceWave, P>(A, work[w], b, invh ? invh + (size_t)g * P * P : nullptr,
                                              ih_entry ? ih_entry + g : nullptr, optimizer);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        conv[g] = (uint8_t)ok;
    }
}

// designs of 13 ... 48 columns: run-time p, one gene per 64-thread workgroup (its workspace is 17 / 33 / 56 KB of LDS)
// PB: the multiple of 8 the column loops walk (>= p)
// COMPACT: the optimiser's matrices in the LDS workspace (lbfgsb_nd) instead of the wavefront's registers (lbfgsb_wave holds a
// 16 x 16 or 32 x 32 inverse, four / sixteen entries per lane): designs of 33 ... 48 columns (round 6), 55 KB of LDS
template <int PMAX, int PB = PMAX, bool COMPACT = false>
__global__ __launch_bounds__(64) void k_shrink_wide(const int32_t* __restrict__ y, int ldn,
                                                    const double* __restrict__ offset, const double* __restrict__ Xt,
                                                    int ldx, int N, int G, int p, const double* __restrict__ size,
                                                    double sigma0, double sigma, int shrink_index,
                                                    double* __restrict__ beta, double* __restrict__ invh,
                                                    uint8_t* __restrict__ conv) {
#ifdef DSQ_SHRINK_COMPACT
    __shared__ ShrinkWorkWide<PMAX> work;
#else
    __shared__ ShrinkWorkWide<PMAX, !COMPACT> work;
#endif
    const int g = blockIdx.x;
    if (g >= G) return;
    ShrinkArgs A;
    A.y = y + (size_t)g * ldn; A.offset = offset; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.size = size[g]; A.sigma0 = sigma0; A.sigma = sigma; A.shrink_index = shrink_index;
    const int ok = shrink_gene_wide<DeviceWave, PMAX, decltype(work), PB>(A, p, work, beta + (size_t)g * p,
                                                                          invh ? invh + (size_t)g * p * p : nullptr);
    if (threadIdx.x == 0) conv[g] = (uint8_t)ok;
}

hipError_t launch_shrink(hipStream_t st, const int32_t* y, int ldn, const double* offset, const double* Xt, int ldx,
                         int N, int G, int P_, const double* size, double sigma0, double sigma, int shrink_index,
                         double* beta, double* invh, uint8_t* conv, double* ih_entry, int optimizer) {
    if (G <= 0) return hipSuccess;
    if (optimizer != 0) {  // "BFGS" / "Newton-CG": the register kernels (at most 12 design columns)
        if (P_ > DSQ_REG_MAX_P || optimizer < 0 || optimizer > 2) return hipErrorInvalidValue;
        const dim3 grid(genes_to_blocks(G)), block(kBlock);
        DSQ_DISPATCH_P(P_, hipLaunchKernelGGL((k_shrink<P, true>), grid, block, 0, st, y, ldn, offset, Xt, ldx, N, G, size,
                                              sigma0, sigma, shrink_index, beta, invh, conv, ih_entry, optimizer))
        return hipGetLastError();
    }
    if (P_ > DSQ_REG_MAX_P) {
        if (P_ > DSQ_SHRINK_MAX_P || ih_entry != nullptr) return hipErrorInvalidValue;  // (the run-time-p kernel writes the whole inverse)
        if (P_ > 32) {
            // 56 KB of static LDS per one-wavefront workgroup
            if (P_ <= 40)
                hipLaunchKernelGGL((k_shrink_wide<48, 40, true>), dim3(G), dim3(64), 0, st, y, ldn, offset, Xt, ldx, N, G, P_,
                                   size, sigma0, sigma, shrink_index, beta, invh, conv);
            else
                hipLaunchKernelGGL((k_shrink_wide<48, 48, true>), dim3(G), dim3(64), 0, st, y, ldn, offset, Xt, ldx, N, G, P_,
                                   size, sigma0, sigma, shrink_index, beta, invh, conv);
        } else if (P_ <= 16)
            hipLaunchKernelGGL((k_shrink_wide<16, 16>), dim3(G), dim3(64), 0, st, y, ldn, offset, Xt, ldx, N, G, P_, size,
                               sigma0, sigma, shrink_index, beta, invh, conv);
        else if (P_ <= 24)
            hipLaunchKernelGGL((k_shrink_wide<32, 24>), dim3(G), dim3(64), 0, st, y, ldn, offset, Xt, ldx, N, G, P_, size,
                               sigma0, sigma, shrink_index, beta, invh, conv);
        else
            hipLaunchKernelGGL((k_shrink_wide<32, 32>), dim3(G), dim3(64), 0, st, y, ldn, offset, Xt, ldx, N, G, P_, size,
                               sigma0, sigma, shrink_index, beta, invh, conv);
        return hipGetLastError();
    }
    const dim3 grid(genes_to_blocks(G)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL((k_shrink<P, false>), grid, block, 0, st, y, ldn, offset, Xt, ldx, N, G, size,
                                          sigma0, sigma, shrink_index, beta, invh, conv, ih_entry, 0))
    return hipGetLastError();
}

}  // namespace dsq
