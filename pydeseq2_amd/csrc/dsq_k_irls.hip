// dsq_k_irls.hip — NB-GLM IRLS kernel (gfx950): one gene per wavefront.
// Algorithmic HBM traffic per gene and launch: 4N bytes of counts read (re-reads of the
// ~4 IRLS sweeps are L1/L2 hits), 8N (mu) + 8N (hat diagonal) written, O(P) scalars.
#include "dsq_dispatch.h"
#include "dsq_irls.h"
#include "dsq_launch.h"

namespace dsq {

// wide designs: ask for 2 waves/SIMD (a few spilled accumulators cost less than running one
// wave per SIMD with nothing to overlap its fp64 dependency chains)
#ifndef DSQ_IRLS_WAVES_P2
#define DSQ_IRLS_WAVES_P2 4
#endif
constexpr int irls_min_waves(int p) { return p <= 2 ? DSQ_IRLS_WAVES_P2 : (p <= 5 ? 1 : 2); }

template <int P>
__global__ __launch_bounds__(kBlock, irls_min_waves(P)) void k_irls(const int32_t* __restrict__ y, int ldn,
                                                 const double* __restrict__ sf, const double* __restrict__ lsf,
                                                 const double* __restrict__ Xt,
                                                 const double* __restrict__ pinvXt, int ldx, int N,
                                                 int G, int full_rank, const double* __restrict__ disp,
                                                 double min_mu, double beta_tol, double min_beta,
                                                 double max_beta, int maxiter, double* __restrict__ beta,
                                                 double* __restrict__ mu, double* __restrict__ hat,
                                                 uint8_t* __restrict__ conv, int32_t* __restrict__ iters,
                                                 int32_t* __restrict__ fb_count,
                                                 int32_t* __restrict__ fb_list) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = lsf; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta;
    A.max_beta = max_beta; A.maxiter = maxiter; A.full_rank = full_rank != 0;
    double b[P];
    const IrlsOut o = irls_gene<DeviceWave, P>(A, b, mu ? mu + (size_t)g * ldn : nullptr,
                                               hat ? hat + (size_t)g * ldn : nullptr);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        conv[g] = (uint8_t)o.converged;
        if (iters != nullptr) iters[g] = o.iters;
        if (o.fallback) fb_list[atomicAdd(fb_count, 1)] = g;
    }
}

template <int P>
__global__ __launch_bounds__(kBlock) void k_irls_rescue(const int32_t* __restrict__ y, int ldn,
                                                        const double* __restrict__ sf, const double* __restrict__ lsf,
                                                        const double* __restrict__ Xt,
                                                        const double* __restrict__ pinvXt, int ldx,
                                                        int N, int full_rank,
                                                        const double* __restrict__ disp, double min_mu,
                                                        double beta_tol, double min_beta, double max_beta,
                                                        int maxiter, double* __restrict__ beta,
                                                        double* __restrict__ mu, double* __restrict__ hat,
                                                        uint8_t* __restrict__ conv,
                                                        int32_t* __restrict__ iters,
                                                        const int32_t* __restrict__ fb_list, int n_fb) {
    int k = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const bool live = k < n_fb;
    if (!live) return;
    const int g = fb_list[k];
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = lsf; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta;
    A.max_beta = max_beta; A.maxiter = maxiter; A.full_rank = full_rank != 0;
    __shared__ IrlsRescueWork<P> work[kWavesPerBlock];
    double b[P];
    const IrlsOut o = irls_rescue_gene<DeviceWave, P>(A, work[threadIdx.x >> 6], b,
                                                      mu ? mu + (size_t)g * ldn : nullptr,
                                                      hat ? hat + (size_t)g * ldn : nullptr);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        conv[g] = (uint8_t)o.converged;
        if (iters != nullptr) iters[g] = o.iters;
    }
}

// grid_search.grid_fit_beta (grid_search.py:145-221) for every gene of a (small) batch: the same routine the
// rescue kernel falls back to, as a stand-alone entry point (tests pin it against the reference's output)
__global__ __launch_bounds__(kBlock) void k_grid_beta(const int32_t* __restrict__ y, int ldn,
                                                      const double* __restrict__ sf, const double* __restrict__ Xt,
                                                      int ldx, int N, int G, const double* __restrict__ disp,
                                                      double min_mu, double min_beta, double max_beta, int grid_length,
                                                      double* __restrict__ beta) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = nullptr; A.Xt = Xt; A.pinvXt = nullptr; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = 0.0; A.min_beta = min_beta; A.max_beta = max_beta;
    A.maxiter = 0; A.full_rank = false;
    const double a = 1.0 / A.disp;
    double c = 0.0;  // sum lgamma(y + a) - lgamma(y + 1) - N lgamma(a), as irls_init_exact
    for (int n = DeviceWave::lane(); n < N; n += 64) {
        const double yv = (double)A.y[n];
        c += lgamma_pos(yv + a) - lgamma_pos(yv + 1.0);
    }
    const double cst = DeviceWave::sum(c) - N * lgamma_pos(a);
    double b[2];
    grid_fit_beta2<DeviceWave>(A, a, cst, b, grid_length);
    if ((threadIdx.x & 63) == 0) { beta[2 * g] = b[0]; beta[2 * g + 1] = b[1]; }
}

hipError_t launch_grid_beta(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt, int ldx,
                            int N, int G, const double* disp, double min_mu, double min_beta, double max_beta,
                            int grid_length, double* beta) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_grid_beta, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, ldn, sf, Xt, ldx, N, G, disp,
                       min_mu, min_beta, max_beta, grid_length, beta);
    return hipGetLastError();
}

hipError_t launch_irls(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* lsf,
                       const double* Xt,
                       const double* pinvXt, int ldx, int N, int G, int P_, int full_rank,
                       const double* disp, double min_mu, double beta_tol, double min_beta,
                       double max_beta, int maxiter, double* beta, double* mu, double* hat,
                       uint8_t* conv, int32_t* iters, int32_t* fb_count, int32_t* fb_list) {
    if (G <= 0) return hipSuccess;
    const dim3 grid(genes_to_blocks(G)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_irls<P>, grid, block, 0, st, y, ldn, sf, lsf, Xt, pinvXt, ldx, N,
                                          G, full_rank, disp, min_mu, beta_tol, min_beta, max_beta,
                                          maxiter, beta, mu, hat, conv, iters, fb_count, fb_list))
    return hipGetLastError();
}

hipError_t launch_irls_rescue(hipStream_t st, const int32_t* y, int ldn, const double* sf,
                              const double* lsf, const double* Xt, const double* pinvXt, int ldx, int N, int P_,
                              int full_rank, const double* disp, double min_mu, double beta_tol,
                              double min_beta, double max_beta, int maxiter, double* beta, double* mu,
                              double* hat, uint8_t* conv, int32_t* iters, const int32_t* fb_list,
                              int n_fb) {
    if (n_fb <= 0) return hipSuccess;
    const dim3 grid(genes_to_blocks(n_fb)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_irls_rescue<P>, grid, block, 0, st, y, ldn, sf, lsf, Xt, pinvXt,
                                          ldx, N, full_rank, disp, min_mu, beta_tol, min_beta, max_beta,
                                          maxiter, beta, mu, hat, conv, iters, fb_list, n_fb))
    return hipGetLastError();
}

}  // namespace dsq
