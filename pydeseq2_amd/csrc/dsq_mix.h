// dsq_mix.h — MIXED designs: a few categorical columns (few distinct rows: "cells") + a few continuous covariates
// (BASELINE configs[4]: a 2-level and a 4-level factor + three continuous covariates, p = 8, N = 5000).
//
// The reference is design-agnostic per gene (utils.py:441-564 fit_alpha_mle, utils.py:273-438 irls_solver): every
// evaluation forms X^T W X from N outer products x_n x_n^T - p (p + 1) / 2 = 36 accumulations per sample at p = 8, twice for
// the dispersion fit (X^T W X and X^T dW X).  With x_n = [x_c(n) ; z_n] (x_c: the categorical part shared by the samples of
// design cell c, z_n: the Q continuous covariates) the matrix splits into
//     categorical x categorical :  sum_c (sum_{n in c} w_n)       x_c x_c^T
//     categorical x continuous  :  sum_c x_c (sum_{n in c} w_n z_n)^T
//     continuous  x continuous  :  sum_n w_n z_n z_n^T
// i.e. 1 + Q + Q (Q + 1) / 2 = 10 accumulations per sample at Q = 3, all in registers WHEN THE SAMPLES OF A CELL ARE
// CONTIGUOUS: the kernels walk the samples in "slot" order - sorted by cell, every cell padded to whole loop iterations
// of kMixU 64-sample trips (padding slots carry count 0, mean 0, covariates 0 and contribute exactly zero) - so the cell
// of a loop iteration is wave-uniform, the per-cell sums are plain per-lane accumulators that are folded into the matrix
// when the cell changes, and nothing needs LDS atomics or per-sample selects.
#pragma once
#include <cstdint>

namespace dsq {

constexpr int kMixMaxQ = 3;      // continuous covariates at most
constexpr int kMixMaxP = 8;      // design columns at most
constexpr int kMixMaxCells = 32; // distinct categorical rows at most
constexpr int kMixTail = 512;    // counts below this enter a gene's tail-count table (as kRowTail)
constexpr int kMixU = 2;         // 64-sample trips per loop iteration; every cell is padded to whole iterations
                                 // (and the row to a multiple of 256 slots: the staging passes walk four trips at a time)

// Device-resident description of one mixed design (built once per design by dsq_mix_create, include/deseq_hip.h).
struct MixDesign {
    const int32_t* perm;       // [Ns] slot -> sample index, -1: padding
    const int32_t* slot_of;    // [N] sample index -> slot (the inverse of perm)
    const uint8_t* trip_cell;  // [Ns / 64] design cell of every 64-slot trip
    const double* Zs;          // [Q][Ns] continuous covariates in slot order, 0 in padding slots
    const double* Xc;          // [C][P] the cells' design rows with the continuous columns set to 0
    const double* Ginv;        // [P][P] (X^T X)^-1 (row-major; IRLS start values), null when X is rank deficient
    int Ns, C, Q, P, N;
    int zcol[kMixMaxQ];        // design column of continuous covariate q
    int colq[kMixMaxP];        // per design column: -1 categorical, else its index q among the continuous ones
};

}  // namespace dsq
