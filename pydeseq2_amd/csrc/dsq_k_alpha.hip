// dsq_k_alpha.hip — dispersion MLE / MAP kernel (gfx950): one gene per wavefront.
// Algorithmic HBM traffic per gene and launch: 4N (counts) + 8N (mu) bytes read once
// (re-reads during the ~5 L-BFGS-B evaluations hit L1/L2: 12 KB per gene at N = 1000),
// 8 + 1 (+4) bytes written.  Compute: ~5 evaluations x N x (lgamma + digamma + 3 log).
#include "dsq_alpha.h"
#include "dsq_dispatch.h"
#include "dsq_launch.h"

#ifndef DSQ_ALPHA_WAVES
#define DSQ_ALPHA_WAVES 1
#endif

namespace dsq {

template <int P>
__global__ __launch_bounds__(kBlock, DSQ_ALPHA_WAVES) void k_alpha(const int32_t* __restrict__ y,
                                                  const double* __restrict__ mu, int ldn,
                                                  const double* __restrict__ Xt, int ldx, int N, int G,
                                                  const double* __restrict__ alpha_hat,
                                                  double min_disp, double max_disp, double prior_var,
                                                  int cr_reg, int prior_reg, double* __restrict__ alpha,
                                                  uint8_t* __restrict__ conv, int32_t* __restrict__ nfev,
                                                  int32_t* __restrict__ grid_count,
                                                  int32_t* __restrict__ grid_list) {
    // the (wave-uniform) optimiser state lives in LDS, not in every lane's registers
    __shared__ Lbfgsb1d machine[kWavesPerBlock];
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    const AlphaOut o = fit_alpha_gene<DeviceWave, P, false>(y + (size_t)g * ldn, mu + (size_t)g * ldn, Xt,
                                                            ldx, N, alpha_hat[g], min_disp, max_disp,
                                                            prior_var, cr_reg != 0, prior_reg != 0,
                                                            machine[threadIdx.x >> 6]);
    if ((threadIdx.x & 63) == 0) {
        alpha[g] = o.alpha;
        conv[g] = (uint8_t)o.converged;
        if (nfev != nullptr) nfev[g] = o.nfev;
        if (!o.converged) grid_list[atomicAdd(grid_count, 1)] = g;
    }
}

// grid-search fallback for the (rare) genes whose L-BFGS-B run reported success = False
template <int P>
__global__ __launch_bounds__(kBlock) void k_alpha_grid(const int32_t* __restrict__ y,
                                                       const double* __restrict__ mu, int ldn,
                                                       const double* __restrict__ Xt, int ldx, int N,
                                                       double min_disp, double max_disp,
                                                       double* __restrict__ alpha,
                                                       const int32_t* __restrict__ grid_list, int n_grid) {
    const int k = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (k >= n_grid) return;
    const int g = grid_list[k];
    const double a = grid_alpha_gene<DeviceWave, P>(y + (size_t)g * ldn, mu + (size_t)g * ldn, Xt, ldx, N,
                                                    min_disp, max_disp);
    if ((threadIdx.x & 63) == 0) alpha[g] = a;
}

hipError_t launch_alpha(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt,
                        int ldx, int N, int G, int P_, const double* alpha_hat, double min_disp,
                        double max_disp, double prior_var, int cr_reg, int prior_reg, double* alpha,
                        uint8_t* conv, int32_t* nfev, int32_t* grid_count, int32_t* grid_list) {
    if (G <= 0) return hipSuccess;
    const dim3 grid(genes_to_blocks(G)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_alpha<P>, grid, block, 0, st, y, mu, ldn, Xt, ldx, N, G,
                                          alpha_hat, min_disp, max_disp, prior_var, cr_reg, prior_reg,
                                          alpha, conv, nfev, grid_count, grid_list))
    return hipGetLastError();
}

hipError_t launch_alpha_grid(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt,
                             int ldx, int N, int P_, double min_disp, double max_disp, double* alpha,
                             const int32_t* grid_list, int n_grid) {
    if (n_grid <= 0) return hipSuccess;
    const dim3 grid(genes_to_blocks(n_grid)), block(kBlock);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_alpha_grid<P>, grid, block, 0, st, y, mu, ldn, Xt, ldx, N,
                                          min_disp, max_disp, alpha, grid_list, n_grid))
    return hipGetLastError();
}

}  // namespace dsq
