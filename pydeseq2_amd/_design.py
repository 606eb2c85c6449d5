"""Host-side preparation of the (tiny, shared) design matrix for the device kernels."""
from __future__ import annotations

import numpy as np


MAX_DESIGN_COLUMNS = 48  # DSQ_MAX_P of include/deseq_hip.h


def pad16(n: int) -> int:
    return (int(n) + 15) & ~15


class DesignPack:
    """Transposed design, its least-squares pseudo-inverse and the design-cell structure.

    * ``Xt``      [P][ldx]  design transposed (lane-coalesced reads of one column)
    * ``pinvXt``  [P][ldx]  rows of (X^T X)^-1 X^T: OLS / QR initial fits
      (utils.py:349-353, 711-713, 846-848) become P dot products per gene
    * ``full_rank``  numpy.linalg.matrix_rank(X) == P (utils.py:349)
    * ``linear_mu``  #unique design rows == P -> linear-model mu_hat route (dds.py:747-750)
    * design cells (identical design rows) for the robust trimmed variance and the
      replaceability rules (utils.py:888-960, dds.py:1311-1313, 1077).
    """

    def __init__(self, X, min_replicates: int = 7):
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        if X.ndim != 2:
            raise ValueError("design matrix must be 2-D")
        if np.isnan(X).any():
            raise ValueError("NaNs are not allowed in the design.")
        self.X = X
        self.N, self.P = X.shape
        if self.P > MAX_DESIGN_COLUMNS:
            # documented limit: the reference's per-gene solvers take any width (utils.py:345-371, numpy / scipy
            # on a p x p system); here the p x p workspaces of the widest kernel family live in a wavefront's LDS segment
            raise ValueError(f"The design matrix has {self.P} columns; the device kernels take at most "
                             f"{MAX_DESIGN_COLUMNS} (DSQ_MAX_P).  Merge or drop design variables.")
        self.ldx = pad16(self.N)
        self.full_rank = bool(np.linalg.matrix_rank(X) == self.P)
        self.Xt = np.zeros((self.P, self.ldx))
        self.Xt[:, : self.N] = X.T
        self.pinvXt = np.zeros((self.P, self.ldx))
        if self.full_rank:
            Q, R = np.linalg.qr(X)
            self.pinvXt[:, : self.N] = np.linalg.solve(R, Q.T)
        rows, inv, cnt = np.unique(X, axis=0, return_inverse=True, return_counts=True)
        inv = np.asarray(inv).reshape(-1)
        self.cell_id, self.cell_sizes = inv, cnt
        self.n_design_cells = len(cnt)
        # cell path of the dispersion / IRLS kernels (include/deseq_hip.h, dsq_cells): per-sample cell index, the
        # cells' design rows and their packed outer products, for designs with at most 64 distinct rows
        # (the launchers pick per kernel: 5..64 cells with P >= 3 -> LDS sums; <= 4 cells with P <= 4 -> register sums)
        self.cell_path = self.n_design_cells <= 64 and ((self.n_design_cells > 4 and self.P >= 3)
                                                        or (self.n_design_cells <= 4 and self.P <= 4))
        if self.cell_path:
            self.cell_of = np.zeros(self.ldx, dtype=np.int32)
            self.cell_of[: self.N] = inv
            self.Xc = np.ascontiguousarray(rows, dtype=np.float64)
            ii, jj = np.tril_indices(self.P)  # row-major lower triangle: entry i(i+1)/2 + j
            self.XXc = np.ascontiguousarray(self.Xc[:, ii] * self.Xc[:, jj])
        self.linear_mu = self.n_design_cells == self.P
        size = cnt[inv]
        self.use_for_max = size >= 3
        self.replaceable = size >= min_replicates
        self.flags = (self.use_for_max.astype(np.uint8) | (self.replaceable.astype(np.uint8) << 1))
        cells = [np.nonzero(inv == c)[0] for c in range(len(cnt)) if cnt[c] >= 3]
        self.whole = len(cells) == 0
        if self.whole:
            self.cell_offsets = np.array([0, self.N], dtype=np.int32)
            self.cell_index = np.arange(self.N, dtype=np.int32)
            self.n_cells, self.max_cell, self.min_cell = 0, self.N, self.N
        else:
            self.cell_offsets = np.concatenate([[0], np.cumsum([len(c) for c in cells])]).astype(np.int32)
            self.cell_index = np.concatenate(cells).astype(np.int32)
            self.n_cells, self.max_cell = len(cells), int(max(len(c) for c in cells))
            self.min_cell = int(min(len(c) for c in cells))
