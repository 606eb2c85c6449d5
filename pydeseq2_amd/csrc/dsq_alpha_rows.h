// dsq_alpha_rows.h — pieces shared by the row kernels of the dispersion fit (dsq_k_alpha_rows.hip: designs with <= 4
// cells == columns, per-cell sums in registers; dsq_k_alpha_rowsc.hip: designs with up to 32 cells, per-cell tables in LDS).
#pragma once
#include "dsq_alpha.h"
#include "dsq_launch.h"

namespace dsq {

constexpr int kRowLanes = 16;    // lanes per gene
constexpr int kRowSlots = 4;     // genes per wavefront
constexpr int kRowWaves = 4;     // wavefronts per workgroup
constexpr int kRowBlock = 64 * kRowWaves;

struct RowGene {  // per-slot record in LDS
    Lbfgsb1d m;
    double cst, la_hat;
    double q[4];   // mu_hat / size factor of the design's cells
    int g;         // gene index, -1: the slot is empty
    int n_tail;    // tail-count entries in use: min(max count, kRowTail), rounded up to the row width
    int n_big;     // samples with a count >= kRowTail
    int pad_;
};

DSQ_HD size_t row_slot_bytes(int npad) {
    return (sizeof(RowGene) + (size_t)npad * 2 + (size_t)kRowTail * 2 + 15) & ~(size_t)15;
}

// lgamma(z), digamma(z) for z >= 256 (truncated Stirling tails, as lgamma_digamma_diff<BIG>)
DSQ_D void stirling_big(double z, double& lg, double& psi) {
    const double l = flog_t(z), rc = frcp(z);
    lg = (z - 0.5) * l - z + kHalfLog2Pi + stirling_tail_big(rc);
    psi = l + digamma_tail_big(rc);
}


}  // namespace dsq
