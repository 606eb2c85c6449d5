"""Device-resident ``deseq2()`` + Wald: the whole hot path without host round trips.

Follows ``DeseqDataSet.deseq2()`` (pydeseq2/dds.py:516-562) stage by stage and then
``DeseqStats.run_wald_test()`` (pydeseq2/ds.py:303-360).  Counts are uploaded once,
transposed on the device into the gene-major int32 layout every per-gene kernel reads,
and the N x G intermediates (mu_hat, mu, hat diagonals, Cook's distances) never leave
HBM unless the caller asks for them.  Only O(G) vectors cross PCIe between stages, for
the two cross-gene steps (dispersion trend / prior, dds.py:799-884) and the bookkeeping
of the outlier refit (dds.py:1042-1110, 1301-1458).

Result fields carry the reference's names: size_factors, normed_means
(var["_normed_means"]), non_zero, genewise_dispersions, fitted_dispersions,
MAP_dispersions, dispersions, LFC (natural log, G x p), cooks_outlier, replaced,
refitted, pvalue / stat / lfcSE, ...
"""
from __future__ import annotations

import os
import time
import warnings
from dataclasses import dataclass, field

import numpy as np
from scipy.stats import f as f_dist

from . import trend as _trend
from ._design import DesignPack, pad16
from ._lib import ALT, HOOK_FN, I32, I64, SAMPLE_MAJOR, Context, DeviceArray, DsqCells, _PinnedPool, _PinnedSlab  # noqa: F401

import ctypes as C
import functools

_vp, c_double = C.c_void_p, C.c_double


@dataclass
class DeseqResult:
    size_factors: np.ndarray = None
    normed_means: np.ndarray = None
    non_zero: np.ndarray = None
    mom_dispersions: np.ndarray = None
    genewise_dispersions: np.ndarray = None
    genewise_converged: np.ndarray = None
    trend_coeffs: np.ndarray = None
    disp_function_type: str = "parametric"
    mean_disp: float = None
    fitted_dispersions: np.ndarray = None
    squared_logres: float = None
    prior_disp_var: float = None
    MAP_dispersions: np.ndarray = None
    MAP_converged: np.ndarray = None
    outlier_genes: np.ndarray = None
    dispersions: np.ndarray = None
    LFC: np.ndarray = None
    LFC_converged: np.ndarray = None
    replaced: np.ndarray = None
    refitted: np.ndarray = None
    new_all_zeroes: np.ndarray = None
    cooks_outlier: np.ndarray = None
    pvalue: np.ndarray = None
    stat: np.ndarray = None
    lfcSE: np.ndarray = None
    timings: dict = field(default_factory=dict)
    kernel_ms: dict = field(default_factory=dict)


def _scatter(G, idx, v, fill=np.nan):
    out = np.full(G, fill)
    out[idx] = v
    return out


@functools.lru_cache(maxsize=64)
def _trigamma(x):
    """polygamma(1, x) (dds.py:882): a constant of the data set's shape, 40 us of scipy per call."""
    from scipy.special import polygamma

    return float(polygamma(1, x))


class _Respeculate(Exception):
    """The pass was enqueued on the previous pass's non-zero mask and the matrix has another one: run it again, in order."""


class _Step:
    """State of one pass of the path: device handles and host scalars the stages hand to each other."""


class _View:
    """A vector inside a larger device buffer (only the address; the buffer is pool-owned)."""

    __slots__ = ("ptr",)

    def __init__(self, ptr):
        self.ptr = ptr


class DeseqPipeline:
    """Holds the device-resident state of one dataset on one GPU.

    Parameters mirror ``DeseqDataSet.__init__`` (dds.py:206-229) where they affect the
    numerics: min_mu, min_disp, max_disp (raised to n_obs, dds.py:312), refit_cooks,
    min_replicates, beta_tol, fit_type.
    """

    def __init__(self, counts, design_matrix, *, ctx: Context | None = None, device: int = 0, min_mu=0.5,
                 min_disp=1e-8, max_disp=10.0, refit_cooks=True, min_replicates=7, beta_tol=1e-8,
                 fit_type="parametric", keep_cooks=True, size_factors_fit_type="ratio", control_genes=None,
                 irls_maxiter=250):
        self.ctx = ctx if ctx is not None else Context(device)
        counts = np.asarray(counts)
        if counts.ndim != 2:
            raise ValueError("counts must be samples x genes")
        if counts.dtype.kind == "f":
            if np.isnan(counts).any():
                raise ValueError("NaNs are not allowed in the count matrix.")
            if (counts % 1 != 0).any():
                raise ValueError("The count matrix should only contain integers.")
            counts = counts.astype(np.int64)
        elif counts.dtype.kind not in "iu":
            raise ValueError("The count matrix should only contain numbers.")
        if counts.dtype not in (np.int32, np.int64):
            counts = counts.astype(np.int64)
        self.N, self.G = counts.shape
        self.design = DesignPack(design_matrix, min_replicates)
        if self.design.N != self.N:
            raise ValueError("design matrix and counts disagree on the number of samples")
        self.P = self.design.P
        self.min_mu, self.min_disp = float(min_mu), float(min_disp)
        self.max_disp = float(max(max_disp, self.N))  # dds.py:312
        self.refit_cooks, self.min_replicates = bool(refit_cooks), int(min_replicates)
        self.beta_tol, self.fit_type = float(beta_tol), fit_type
        self.irls_maxiter = int(irls_maxiter)  # `maxiter` of Inference.irls (inference.py:46-119; dds.py never changes it)
        self.keep_cooks = keep_cooks
        if size_factors_fit_type not in ("ratio", "poscounts", "iterative"):
            raise ValueError("size_factors_fit_type: 'ratio' (median of ratios), 'poscounts' or 'iterative'")
        self.size_factors_fit_type = size_factors_fit_type
        self._control_mask = None
        if control_genes is not None:  # boolean mask or integer indices (dds.py:640-650)
            m = np.zeros(self.G if hasattr(self, "G") else counts.shape[1], dtype=np.uint8)
            m[np.asarray(control_genes)] = 1
            self._control_mask = m
        self.ldn = pad16(self.N)
        ctx_ = self.ctx
        # ---- resident inputs: the sample-major matrix narrowed to int32 on its way up (pinned staging chunks,
        # dsq_upload_counts_i32) - kept for the per-sample medians - and its gene-major transposition
        counts = np.ascontiguousarray(counts)
        self._count_type = I32
        self.d_raw = DeviceArray(ctx_, (self.N, self.G), np.int32)
        bad = C.c_int(0)
        ctx_.call("dsq_upload_counts_i32", _vp(counts.ctypes.data), I32 if counts.dtype == np.int32 else I64,
                  C.c_size_t(counts.size), _vp(self.d_raw.ptr), C.byref(bad))
        if bad.value:
            raise ValueError("The count matrix should only contain non-negative integers below 2^31.")
        self.d_y = DeviceArray(ctx_, (self.G, self.N), np.int32, ld=self.ldn)
        ctx_.call("dsq_dev_counts_to_gene_major", _vp(self.d_raw.ptr), I32, SAMPLE_MAJOR, self.N,
                  self.G, _vp(self.d_y.ptr), self.ldn, C.byref(bad))
        D = self.design
        self.d_Xt = DeviceArray.from_host(ctx_, D.Xt)
        self.d_pinv = DeviceArray.from_host(ctx_, D.pinvXt)
        self.d_flags = DeviceArray.from_host(ctx_, D.flags)
        self.d_cell_off = DeviceArray.from_host(ctx_, D.cell_offsets)
        self.d_cell_idx = DeviceArray.from_host(ctx_, D.cell_index)
        self._cells = None  # dsq_cells: designs with 5..64 distinct rows take the per-cell kernels (P >= 3)
        if D.cell_path and not os.environ.get("DSQ_NO_CELL_PATH"):
            self._d_cell_of = DeviceArray.from_host(ctx_, D.cell_of)
            self._d_Xc = DeviceArray.from_host(ctx_, D.Xc)
            self._d_XXc = DeviceArray.from_host(ctx_, D.XXc)
            self._cells = DsqCells(self._d_cell_of.ptr, self._d_Xc.ptr, self._d_XXc.ptr, int(D.n_design_cells))
        # dispersion fits with four genes per wavefront (csrc/dsq_k_alpha_rows.hip: linear-model mu_hat, <= 4 design cells):
        # which genes may take it depends on the counts only (1 = stays on the one-gene-per-wavefront kernel)
        self._row_flags, self._row_lists = None, None
        # 1: <= 4 cells == columns (linear-model mu_hat), 2: up to 32 cells, p <= 8 (either mu_hat route), 0: none
        self._row_mode = 0 if (self._cells is None or os.environ.get("DSQ_NO_ALPHA_ROWS")) else \
            int(ctx_.lib.dsq_alpha_rows_eligible(self.N, self.P, int(D.n_design_cells)))
        if self._row_mode == 1 and not D.linear_mu:
            self._row_mode = 0
        # 3: MIXED design - categorical columns with few distinct rows + up to three continuous covariates (dsq_mix_create,
        # csrc/dsq_mix.h: the analysis and the eligibility rule live in the library).  IRLS mu_hat route only.
        self._mix, self._mix_slots, self._slot_of, self._mix_ys_cache = None, 0, None, None
        if self._row_mode == 0 and not D.linear_mu and not os.environ.get("DSQ_NO_ALPHA_MIX"):
            mp = _vp()
            ctx_.call("dsq_mix_create", _vp(D.X.ctypes.data), self.N, self.P, C.byref(mp))
            if mp.value:
                self._mix, self._row_mode = mp.value, 3
                ns = C.c_int()
                if ctx_.lib.dsq_mix_info(mp, C.byref(ns), None, None) != 0:
                    raise RuntimeError("dsq_mix_info failed")
                self._mix_slots = int(ns.value)
                # the Cook's layer of such a design is written in slot order (coalesced rows, csrc/dsq_k_irls_mix.hip):
                # whoever reads it - the outlier replacement on the device, layer() on the host - goes through this map
                self._slot_of = np.empty(self.N, dtype=np.int32)
                if ctx_.lib.dsq_mix_slots(mp, _vp(self._slot_of.ctypes.data)) != 0:
                    raise RuntimeError("dsq_mix_slots failed")
        if self._row_mode:
            d_fl = DeviceArray(ctx_, (self.G,), np.int32)
            ctx_.call("dsq_dev_alpha_row_split", _vp(self.d_y.ptr), self.ldn, self.N, self.G, _vp(d_fl.ptr))
            self._row_flags = d_fl.to_host()  # -1: not for the row kernel, else the gene's number of counts >= 512
            d_fl.free()
        self.keep_layers = False   # True: the LFC fit also writes the N x G layers mu / hat diagonals (else on demand)
        self.overlap = not os.environ.get("DSQ_NO_OVERLAP")  # robust dispersions on a side stream under the trend fit
        self._robust_early = bool(os.environ.get("DSQ_ROBUST_EARLY"))  # (measurement switch: fork before the genewise fit)
        self._robust_late = bool(os.environ.get("DSQ_ROBUST_LATE"))    # (measurement switch: fork after the genewise stage)
        # the MAP launch waits for the side stream (robust dispersions) where that kernel is about as long as the tail it hides
        # under (row kernels: c2 / c3 5.92 -> 5.80 ms); the mixed-design family's lean kernel runs 2-3 ms past that tail at
        # 5000 samples, where sharing the compute units costs less than idling (c5: 51.7 vs 52.9 ms).  DSQ_MAP_WAIT=0 / 1.
        mw = os.environ.get("DSQ_MAP_WAIT")
        # (c4, the many-cell row kernel: the robust kernel ends inside the tail - 7.09 / 7.12 ms without the wait, 7.11 / 7.20
        # with it)
        self._map_waits_side = (self._row_mode == 1) if mw is None else (mw != "0")
        # share of the genes whose robust dispersions run under the genewise stage's tail (the rest: under the MAP stage's)
        self._robust_split = float(os.environ.get("DSQ_ROBUST_SPLIT", "0.78"))
        # LFC fit in two launches: the genes whose MAP dispersion is final after the MAP stage's full-size launch are fitted
        # underneath that stage's latency-bound tail (_fork_lfc); DSQ_LFC_OVERLAP=0: one launch after the stage (A/B switch)
        # Measured (same box each, ms per step with / without): mixed designs - c5 shard of 7500 genes 7.15-7.18 / 7.73-7.79,
        # 15000 genes 13.0 / 13.3-13.5, 30000 genes 23.3-23.5 / 23.5-23.9 (a tail of 0.8 ms - continuation 0.6, host round
        # trip, grid pass 0.16 - on a few workgroups), the whole c5 matrix 45.0 / 44.75 (the robust-dispersion kernel of the
        # side stream runs 2-3 ms past the tail there, see above: nothing idle to fill).  Many cells: c4 6.80-6.84 / 6.77-6.84,
        # its shards the same to 0.05 ms (one box showed 7.08 / 7.37; three others nothing).  <= 4 cells: c3 5.74 / 5.60, c2
        # 1.84 / 1.69, c3 shard 1.80 / 1.79 - a 0.35 ms tail that already carries the robust dispersions' part two, the device is
        # 88 % busy without the fork, and the small launches beside a full-size kernel cost more than the rest is worth.
        # General kernels 16.95 / 16.67.  Hence: on for the mixed-design family up to DSQ_LFC_OVERLAP_MAX_WORK = 1.5e8 counts
        # per device (the shards of a multi-GPU job), off elsewhere.
        lo = os.environ.get("DSQ_LFC_OVERLAP")
        max_work = float(os.environ.get("DSQ_LFC_OVERLAP_MAX_WORK", "1.5e8"))
        auto = self._row_mode == 3 and float(self.G) * self.N <= max_work
        self._lfc_overlap = auto if lo is None else (lo != "0")
        self._lfc_min_genes = int(os.environ.get("DSQ_LFC_OVERLAP_MIN_GENES", "2048"))
        self._lfc_forked = False
        self.lfc_forks = 0  # passes whose LFC fit ran in two launches (tests)
        self._work = None
        self.layers = {}
        self.time_kernels = False
        self.collect_nfev = False  # profiling aid: log the L-BFGS-B evaluation count of every dispersion launch
        self.kernel_log = {}
        self._pool_free, self._pool_used = [], []
        self._inflight = []
        self._side_pending = False
        self._pinned = _PinnedPool(ctx_)
        # Which genes have a count at all depends on the counts only: the mask every pass compacts to is computed here, once,
        # so that the first deseq2() call already enqueues its genewise stage without waiting for it (each pass still
        # re-derives the mask on the device and compares), and the gene lists of the row kernels are built off the step.
        d_lm0, d_nz0 = DeviceArray(ctx_, (self.G,), np.float64), DeviceArray(ctx_, (self.G,), np.uint8)
        ctx_.call("dsq_dev_logmeans", _vp(self.d_y.ptr), self.ldn, self.N, self.G, _vp(d_lm0.ptr), _vp(d_nz0.ptr))
        self._nz_pred = d_nz0.to_host().astype(bool)
        d_lm0.free(); d_nz0.free()
        self._row_lists_for(self._nz_pred)
        # Cook's cutoff F.ppf(0.99, p, N - p) (dds.py:1073, 1324): a scipy call of ~0.1 ms, off the step's path
        self._cooks_cutoff = float(f_dist.ppf(0.99, self.P, self.N - self.P)) if self.N > self.P else float("nan")
        ctx_.sync()

    # ------------------------------------------------------------------ helpers
    # Device buffers are pooled: a deseq2() step re-uses the allocations of the previous one
    # (hipMalloc/hipFree of the 0.5 GB N x G layers every step costs milliseconds and hipFree
    # synchronises the device).  Buffers handed out during a step stay valid until the next step.
    def _take(self, nbytes):
        nbytes = max(int(nbytes), 8)
        best = None
        for i, (cap, ptr) in enumerate(self._pool_free):
            if cap >= nbytes and (best is None or cap < self._pool_free[best][0]) and cap <= 2 * nbytes + 4096:
                best = i
        if best is not None:
            cap, ptr = self._pool_free.pop(best)
        else:
            cap, ptr = nbytes, self.ctx.malloc(nbytes)
        self._pool_used.append((cap, ptr))
        return ptr

    def _pool_reset(self):
        if getattr(self, "_side_pending", False) or getattr(self, "_lfc_forked", False):
            # a previous step died between a fork and its join (side stream, forked LFC launch): their kernels may still be
            # writing pooled buffers
            self.ctx.call("dsq_side_abort")
            self._side_pending = False
            self._lfc_forked = False
        self._pool_free.extend(self._pool_used)
        self._pool_used = []
        self._serial = getattr(self, "_serial", 0) + 1  # an open pass (_Step) lives until the next reset
        if getattr(self, "_inflight", None):  # staging slabs of _up(): make sure their copies have run
            self.ctx.sync()
        self._inflight = []
        self.layers = {}  # they point into the buffers that have just been recycled

    def _pooled(self, shape, dtype, ld=None):
        arr = DeviceArray.__new__(DeviceArray)
        arr.ctx = self.ctx
        arr.shape = tuple(shape)
        arr.dtype = np.dtype(dtype)
        arr.ld = ld if ld is not None else (arr.shape[-1] if len(arr.shape) > 1 else None)
        n = arr.shape[0] * (arr.ld if len(arr.shape) > 1 else 1)
        arr.nbytes = int(n) * arr.dtype.itemsize
        arr.ptr = self._take(arr.nbytes)
        arr.free = lambda: None  # owned by the pool
        return arr

    def _dvec(self, n, dtype=np.float64):
        return self._pooled((max(int(n), 1),), dtype)

    def _dmat(self, rows, dtype=np.float64):
        return self._pooled((max(int(rows), 1), self.N), dtype, ld=self.ldn)

    def _up(self, arr, dtype=np.float64):
        """Host array -> pooled device buffer.  Small arrays (index lists of the refit, ...) go through a page-locked
        staging slab and an asynchronous copy: a pageable source makes the copy - and the host - wait for everything
        queued before it.  The slab stays referenced until the pool is reset (the copy is stream-ordered)."""
        arr = np.ascontiguousarray(arr, dtype=dtype)
        d = self._pooled(arr.shape if arr.ndim else (1,), arr.dtype)
        if arr.size:
            if arr.nbytes <= 65536:
                hs = self._host_slab(arr.nbytes)
                hs.view(0, arr.size, arr.dtype)[:] = arr.reshape(-1)
                self.ctx.call("dsq_h2d_async", _vp(d.ptr), _vp(hs.ptr), C.c_size_t(arr.nbytes))
                self._inflight.append(hs)
            else:
                self.ctx.h2d(d.ptr, arr)
        return d

    def _down_nonzero(self):
        """Mask of genes with at least one count (cached by deseq2() before the iterative size factors)."""
        return self._nz_cache

    def _all_genes_host(self, d_vec, n):
        """Host copy of a per-gene device vector over ALL genes of the data set (the gene-sharded pipeline
        concatenates the ranks' vectors; here: this pipeline's n genes)."""
        return self._down(d_vec, n)

    def _down(self, darr, n, dtype=np.float64):
        out = np.empty(int(n), dtype=dtype)
        if n:
            self.ctx.d2h(out, darr.ptr)
        return out

    def _k(self, name, genes, cname, *args):
        """Launch a per-gene stage; with ``time_kernels`` bracket it with HIP events on the
        context's stream and record (milliseconds, genes) under ``name`` (this synchronises the host
        after every stage: profiling mode only).  The dispersion kernel's own duration comes from
        events recorded inside the C call and is logged in every mode."""
        if self.time_kernels:
            self.ctx.timer_start()
            self.ctx.call(cname, *args)
            ms = self.ctx.timer_stop()
            self.kernel_log.setdefault(name, []).append((ms, int(genes)))
        else:
            self.ctx.call(cname, *args)
        if cname in ("dsq_dev_alpha_mle", "dsq_dev_alpha_mle2", "dsq_dev_alpha_mle3", "dsq_dev_alpha_mle4"):
            kms, ng = C.c_float(), C.c_int()
            self.ctx.call("dsq_last_alpha_kernel", C.byref(kms), C.byref(ng))
            if kms.value >= 0.0:  # (-1: a deferred launch, nobody waited for it)
                self.kernel_log.setdefault("k_alpha", []).append((float(kms.value), int(genes)))
                self.kernel_log.setdefault("k_alpha_stage", []).append((name, int(genes)))
                self.kernel_log.setdefault("grid_fallback_genes", []).append((float(ng.value), int(genes)))

    # ------------------------------------------------------------------ result slabs
    # All per-gene result vectors of a step live in ONE device buffer laid out like one page-locked
    # host buffer, so that they come back with two DMA copies at the end of the step instead of ~20
    # pageable round trips in between.  _F64 / _U8 name the vectors (each Gs long; "beta" is Gs x P).
    _F64 = ("nm", "mom", "gw", "fit", "map", "disp", "p", "stat", "se", "rd")
    # ("naz": all counts zero after the outlier replacement - written by k_replace for the refit's sub-problem only)
    _U8 = ("gconv", "mconv", "outl", "lconv", "any_all", "any_use", "any_use_nr", "few_above", "naz")

    def _slab_layout(self, Gs):
        off, o = {}, 0
        for k in self._F64:
            off[k] = o
            o += 8 * Gs
        off["beta"] = o
        o += 8 * Gs * self.P
        n_f64 = o
        for k in self._U8:
            off[k] = o
            o += Gs
        return off, n_f64, ((o + 63) // 64) * 64

    def _dev_slab(self, Gs):
        """Device result buffer of a (sub-)problem of Gs genes: namespace of _View objects."""
        off, n_f64, total = self._slab_layout(Gs)
        base = self._take(total)
        v = {k: _View(base + o) for k, o in off.items()}
        v["_base"], v["_off"], v["_n_f64"], v["_total"], v["_Gs"] = base, off, n_f64, total, Gs
        return v

    def _host_slab(self, nbytes):
        """Page-locked host buffer from the pipeline's free list (returned to it when the last numpy
        view of the previous result dies)."""
        return self._pinned.take(nbytes)

    def _fetch(self, slab, names=None):
        """Device slab -> numpy views over a pinned host slab (dict name -> array)."""
        tok = self._fetch_begin(slab, names)
        self.ctx.sync()
        return self._fetch_end(tok)

    def _fetch_begin(self, slab, names=None):
        """Enqueue the copies of _fetch without waiting; _fetch_end(token) after any later synchronisation of the
        stream gives the views (the refit's small vectors ride on the final fetch's synchronisation)."""
        Gs, off = slab["_Gs"], slab["_off"]
        width = lambda k: Gs * (8 * self.P if k == "beta" else (8 if k in self._F64 else 1))  # noqa: E731
        if names is None:
            host, hoff = self._host_slab(slab["_total"]), off
            self.ctx.call("dsq_d2h_async", _vp(host.ptr), _vp(slab["_base"]), C.c_size_t(slab["_total"]))
            names = list(self._F64) + ["beta"] + list(self._U8)
        else:  # a few vectors: a compact host buffer of their own
            hoff, o = {}, 0
            for k in names:
                hoff[k] = o
                o += (width(k) + 63) // 64 * 64
            host = self._host_slab(o)
            for k in names:
                self.ctx.call("dsq_d2h_async", _vp(host.ptr + hoff[k]), _vp(slab["_base"] + off[k]),
                              C.c_size_t(width(k)))
        return host, hoff, names, Gs

    def _fetch_end(self, tok):
        host, hoff, names, Gs = tok
        out = {}
        for k in names:
            if k == "beta":
                out[k] = host.view(hoff[k], Gs * self.P, np.float64).reshape(Gs, self.P)
            elif k in self._F64:
                out[k] = host.view(hoff[k], Gs, np.float64)
            else:
                out[k] = host.view(hoff[k], Gs, np.uint8)
        return out

    # ------------------------------------------------------------------ stages (device in, device out)
    def _cells_arg(self):
        return C.byref(self._cells) if self._cells is not None else None

    def _row_lists_for(self, non_zero):
        """Gene lists of the two dispersion kernels in the index space of the compacted (non-zero) genes; cached: they
        depend on the counts only."""
        if self._row_flags is None:
            return None
        c = self._row_lists
        if c is None or not np.array_equal(c[0], non_zero):
            fl = self._row_flags[non_zero]
            rows, waves = np.nonzero(fl >= 0)[0], np.nonzero(fl < 0)[0].astype(np.int32)
            # high-count genes (samples beyond the tail-count table cost a second sweep per evaluation) first and together:
            # a wavefront then holds four of a kind, and the longer fits start early
            rows = rows[np.argsort(-fl[rows], kind="stable")].astype(np.int32)
            if c is not None:
                c[1].free(); c[3].free()
            d_rows = DeviceArray.from_host(self.ctx, rows if len(rows) else np.zeros(1, np.int32))
            d_waves = DeviceArray.from_host(self.ctx, waves if len(waves) else np.zeros(1, np.int32))
            c = self._row_lists = (non_zero.copy(), d_rows, len(rows), d_waves, len(waves))
        return (c[1], c[2], c[3], c[4]) if c[2] > 0 else None

    def _mix_slots_for(self, d_y, Gs, persistent_key=None):
        """Slot-ordered uint16 copy of a gene-major count matrix + per-gene "a count beyond 16 bits" flags for the
        mixed-design kernels (dsq_dev_mix_counts_to_slots).  The copy of the pipeline's own matrix depends on the counts
        only: it is built once (per non-zero mask) and kept; a sub-problem's (the refit's replaced counts) is pooled."""
        if self._mix is None:
            return None
        if persistent_key is not None:
            c = self._mix_ys_cache
            if c is not None and np.array_equal(c[0], persistent_key):
                return c[1], c[2]
            if c is not None:
                c[1].free(); c[2].free()
            d_ys = DeviceArray(self.ctx, (max(Gs, 1) * self._mix_slots,), np.uint16)
            d_big = DeviceArray(self.ctx, (max(Gs, 1),), np.uint8)
        else:
            d_ys = self._pooled((max(Gs, 1) * self._mix_slots,), np.uint16)
            d_big = self._pooled((max(Gs, 1),), np.uint8)
        self.ctx.call("dsq_dev_mix_counts_to_slots", _vp(d_y.ptr), self.ldn, Gs, _vp(self._mix), _vp(d_ys.ptr),
                      _vp(d_big.ptr))
        if persistent_key is not None:
            self._mix_ys_cache = (np.array(persistent_key, copy=True), d_ys, d_big)
        return d_ys, d_big

    def _mix_bind(self, slots, d_mu_slots=None, genes=0):
        """Hand the slot-ordered copies to the next fit of the context (one-shot, csrc: dsq_mix_bind2: a fit over another
        number of genes than the copies were built for ignores them)."""
        if slots is not None:
            self.ctx.call("dsq_mix_bind2", _vp(slots[0].ptr), _vp(slots[1].ptr), _vp(d_mu_slots.ptr) if d_mu_slots else None,
                          int(genes))

    def _stage_genewise(self, d_y, Gs, d_sf, S, row_lists=None, pre_alpha=None, mix_slots=None):
        """MoM -> mu_hat -> genewise alpha for Gs genes (dds.py:713-797).  Writes S[nm, mom, gw (raw,
        unclipped), gconv]; returns the description of mu_hat the MAP fit needs: the device matrix, or - for
        the designs with the linear-model mu_hat (dds.py:747-756) - only the per-gene OLS coefficients from
        which the dispersion kernel rebuilds mu_hat = max(sf * X coef, min_mu) while staging."""
        D = self.design
        mh = type("MuHat", (), {})()
        mh.d_mu, mh.d_coef, mh.d_cell_mu, mh.d_beta, mh.row_lists = None, None, None, None, row_lists
        mh.mix_slots, mh.d_mu_slots, mh.d_beta_fit = mix_slots, None, None
        if D.linear_mu:  # dds.py:747-756: MoM and the linear-model mu_hat share their sweeps
            mh.d_coef = self._dvec(Gs * self.P)
            # rows too long for the LDS staging of launch_alpha, or a design wider than the register kernels
            # (the LDS / matrix-core path of dsq_k_wide.hip reads mu_hat): materialise it
            if self.ctx.lib.dsq_alpha_needs_mu(self.N, self.P, int(D.n_design_cells) if self._cells is not None else 0):
                mh.d_mu = self._dmat(Gs)
            self._k("mom_lin_mu", Gs, "dsq_dev_mom_lin_coef", _vp(d_y.ptr), self.ldn, _vp(d_sf.ptr),
                    _vp(self.d_Xt.ptr), _vp(self.d_pinv.ptr), D.ldx, self.N, Gs, self.P, c_double(self.min_disp),
                    c_double(self.max_disp), c_double(self.min_mu), _vp(S["nm"].ptr), _vp(S["mom"].ptr),
                    _vp(mh.d_mu.ptr) if mh.d_mu else None, _vp(mh.d_coef.ptr))
        else:  # dds.py:757-765: IRLS with the MoM dispersions, mu only is kept
            # designs the many-cell row kernel takes: mu_hat stays in its per-cell form sf_n * exp(x_c . beta)
            # (dsq_dev_cell_mu) - the N x G matrix is neither written by the IRLS kernel nor read by the two fits
            per_cell = (self._row_mode == 2 and row_lists is not None and row_lists[3] == 0
                        and not self.ctx.lib.dsq_alpha_needs_mu(self.N, self.P, int(D.n_design_cells)))
            # mixed designs: the dispersion kernel rebuilds mu_hat = sf * exp(X beta) from the coefficients of this fit
            # (every gene on it: no count beyond its 16-bit staging); otherwise it gathers its rows from the matrix
            from_beta = self._row_mode == 3 and row_lists is not None and row_lists[3] == 0
            mh.d_mu = None if (per_cell or from_beta) else self._dmat(Gs)
            self._k("mom", Gs, "dsq_dev_mom", _vp(d_y.ptr), self.ldn, _vp(d_sf.ptr), _vp(self.d_Xt.ptr),
                    _vp(self.d_pinv.ptr), D.ldx, self.N, Gs, self.P, c_double(self.min_disp), c_double(self.max_disp),
                    _vp(S["nm"].ptr), None, None, _vp(S["mom"].ptr))
            d_b, d_c = self._dvec(Gs * self.P), self._dvec(Gs, np.uint8)
            # the iteration counts of this fit order the genes of the LFC fit (dsq_irls_order_hint)
            S["_irls_it"] = self._dvec(Gs, np.int32)
            self._mix_bind(mix_slots, genes=Gs)
            self._k("irls_mu", Gs, "dsq_dev_lfc_fit2", _vp(d_y.ptr), self.ldn, _vp(d_sf.ptr), _vp(self.d_Xt.ptr),
                    _vp(self.d_pinv.ptr), D.ldx, self.N, Gs, self.P, int(D.full_rank), _vp(S["mom"].ptr),
                    c_double(self.min_mu), c_double(self.beta_tol), c_double(-30.0), c_double(30.0), self.irls_maxiter,
                    _vp(d_b.ptr), _vp(mh.d_mu.ptr) if mh.d_mu else None, None, _vp(d_c.ptr), _vp(S["_irls_it"].ptr),
                    self._cells_arg(),
                    None, None, c_double(0.0), None, None, None, None, None,
                    None, None, c_double(0.0), 0, None, None, None, _vp(self._mix) if self._mix else None, 0)
            mh.d_beta_fit = d_b  # (whatever the route: layer("_mu_hat") rebuilds the matrix from these on demand)
            if from_beta:
                mh.d_beta = d_b
                if mix_slots is not None:
                    # mu_hat = sf * exp(X beta) in slot order, written ONCE here and streamed by both dispersion fits (they
                    # used to rebuild it - one exponential per sample - in every launch and continuation launch)
                    mh.d_mu_slots = self._pooled((max(Gs, 1) * self._mix_slots,), np.float64)
                    self.ctx.call("dsq_dev_mix_mu_slots", _vp(self._mix), _vp(d_b.ptr), _vp(d_sf.ptr), Gs,
                                  _vp(mh.d_mu_slots.ptr))
            if per_cell:
                mh.d_cell_mu = self._dvec(Gs * int(D.n_design_cells))
                self.ctx.call("dsq_dev_cell_mu", _vp(d_b.ptr), self._cells_arg(), Gs, self.P, _vp(mh.d_cell_mu.ptr))
        mh.nll_const = self._dvec(Gs)  # sum lgamma(y+1) - y log(mu_hat): stored here, re-used by the MAP fit
        if pre_alpha is not None:
            pre_alpha()
        self._alpha_fit("alpha_mle", d_y, mh, Gs, d_sf, S["mom"], 1.0, 0, S["gw"], S["gconv"], 1)
        return mh

    def _alpha_fit(self, name, d_y, mh, Gs, d_sf, d_start, prior_var, prior_reg, d_out, d_conv, const_mode):
        d_nfev = self._dvec(Gs, np.int32) if self.collect_nfev else None
        d_cell_mu = getattr(mh, "d_cell_mu", None)
        d_beta = getattr(mh, "d_beta", None)
        use_coef = mh.d_mu is None and d_cell_mu is None and d_beta is None
        use_cell = d_cell_mu is not None
        mix = self._mix if (self._row_mode == 3 and getattr(mh, "row_lists", None) is not None) else None
        # (d_rows, n_rows, d_waves, n_waves) or None; the mixed-design kernel also takes its rows from a mu_hat matrix
        rows = getattr(mh, "row_lists", None) if (mh.d_mu is None or mix) else None
        if mix and rows:
            self._mix_bind(getattr(mh, "mix_slots", None), getattr(mh, "d_mu_slots", None), genes=Gs)
        self._k(name, Gs, "dsq_dev_alpha_mle4", _vp(d_y.ptr), _vp(mh.d_mu.ptr) if mh.d_mu else None, self.ldn,
                _vp(self.d_Xt.ptr), self.design.ldx, self.N, Gs, self.P, _vp(d_start.ptr), c_double(self.min_disp),
                c_double(self.max_disp), c_double(prior_var), 1, int(prior_reg), _vp(d_out.ptr), _vp(d_conv.ptr),
                _vp(d_nfev.ptr) if d_nfev else None, _vp(mh.nll_const.ptr), const_mode, self._cells_arg(),
                _vp(mh.d_coef.ptr) if use_coef else None, _vp(d_sf.ptr) if (use_coef or use_cell or d_beta) else None,
                c_double(self.min_mu), _vp(rows[0].ptr) if rows else None, rows[1] if rows else 0,
                _vp(rows[2].ptr) if rows and rows[3] else None, rows[3] if rows else 0,
                _vp(d_cell_mu.ptr) if use_cell else None, _vp(mix) if (mix and rows) else None,
                _vp(d_beta.ptr) if d_beta else None)
        if d_nfev:
            self.kernel_log.setdefault("nfev", []).append((float(self._down(d_nfev, Gs, np.int32).sum()), Gs))

    def _stage_map(self, d_y, mh, Gs, d_sf, prior_var, squared_logres, S, fork=None):
        """MAP dispersions (dds.py:886-935) from S[fit] -> S[map (raw), mconv], then the final
        dispersions S[disp] and the dispersion-outlier flags S[outl].  ``fork`` = (part vector, late flags, forked()):
        the stage was set up for an LFC fit in two launches (_st_map) - the launches behind the hook's point write their
        convergence flags to the late vector, and the selection covers the genes the fork has not taken (all of them when
        the hook did not fork after all)."""
        if fork is not None:
            self.ctx.call("dsq_alpha_set_late_flags", _vp(fork[1].ptr))
        self._alpha_fit("alpha_map", d_y, mh, Gs, d_sf, S["fit"], prior_var, 1, S["map"], S["mconv"], 2)
        if fork is not None:
            d_part, d_late, forked = fork
            if not forked():
                self.ctx.call("dsq_memset", _vp(d_part.ptr), 0, C.c_size_t(Gs))
            self.ctx.call("dsq_dev_select_dispersions_part", _vp(S["gw"].ptr), _vp(S["map"].ptr), _vp(S["fit"].ptr), Gs,
                          c_double(self.min_disp), c_double(self.max_disp), c_double(squared_logres),
                          _vp(S["disp"].ptr), _vp(S["outl"].ptr), _vp(S["mconv"].ptr), _vp(d_late.ptr), _vp(d_part.ptr),
                          0, 0)
        else:
            self.ctx.call("dsq_dev_select_dispersions", _vp(S["gw"].ptr), _vp(S["map"].ptr), _vp(S["fit"].ptr), Gs,
                          c_double(self.min_disp), c_double(self.max_disp), c_double(squared_logres),
                          _vp(S["disp"].ptr), _vp(S["outl"].ptr))

    def _stage_lfc(self, d_y, Gs, d_sf, S, wald, cooks=None, mix_slots=None, bufs=None, part=None):
        """IRLS LFC fit (dds.py:937-984) with S[disp] -> S[beta, lconv] and, fused into its epilogue, the Wald
        statistics S[p, stat, se] (ds.py:303-360; wald = (ridge, contrast, lfc_null, alt)) and - cooks =
        (robust dispersions, cutoff, cooks layer) - the per-sample half of the Cook's stage (dds.py:986-1040,
        1066-1110): S[any_all, any_use, any_use_nr, few_above].  Returns device (mu, hat) when keep_layers."""
        D = self.design
        want = self.keep_layers and cooks is not None
        if bufs is not None:  # the second launch of a fit in two (part): the layers of the first
            d_mu, d_hat = bufs
        else:
            d_mu = self._dmat(Gs) if want else None
            d_hat = self._dmat(Gs) if want else None
        ridge, contrast, lfc_null, alt = wald
        ck = [None, None, c_double(0.0), None, None, None, None, None]
        cooks_ld = 0
        if cooks is not None:
            d_rd, cutoff, d_cooks = cooks
            cooks_ld = self._cooks_ld()
            ck = [_vp(d_rd.ptr), _vp(self.d_flags.ptr), c_double(cutoff), _vp(d_cooks.ptr)] + \
                 [_vp(S[x].ptr) for x in ("any_all", "any_use", "any_use_nr", "few_above")]
        if S.get("_irls_it") is not None:
            self.ctx.call("dsq_irls_order_hint", _vp(S["_irls_it"].ptr), Gs)
        self._mix_bind(mix_slots, genes=Gs)
        if part is not None:  # (device vector, value of this launch's genes, phase): csrc dsq_lfc_set_part, one-shot
            self.ctx.call("dsq_lfc_set_part", _vp(part[0].ptr), int(part[1]), int(part[2]))
        self._k("lfc_fit", Gs, "dsq_dev_lfc_fit2", _vp(d_y.ptr), self.ldn, _vp(d_sf.ptr), _vp(self.d_Xt.ptr),
                _vp(self.d_pinv.ptr), D.ldx, self.N, Gs, self.P, int(D.full_rank), _vp(S["disp"].ptr),
                c_double(self.min_mu), c_double(self.beta_tol), c_double(-30.0), c_double(30.0), self.irls_maxiter,
                _vp(S["beta"].ptr), _vp(d_mu.ptr) if d_mu else None, _vp(d_hat.ptr) if d_hat else None,
                _vp(S["lconv"].ptr), None, self._cells_arg(), *ck,
                _vp(ridge.ctypes.data), _vp(contrast.ctypes.data), c_double(lfc_null), alt,
                _vp(S["p"].ptr), _vp(S["stat"].ptr), _vp(S["se"].ptr), _vp(self._mix) if self._mix else None, cooks_ld)
        return d_mu, d_hat

    def _cooks_ld(self):
        """Row pitch of a slot-ordered Cook's layer (mixed designs the IRLS kernel takes), else 0: sample order, pitch ldn."""
        if self._mix and self.ctx.lib.dsq_mix_takes_irls(_vp(self._mix), int(self.design.full_rank)):
            return self._mix_slots
        return 0

    # ------------------------------------------------------------------ cross-gene steps (hooks)
    # The only places where a gene needs other genes; DistDeseqPipeline (distributed.py) overrides
    # them with their multi-GPU versions.
    def _size_factors(self, d_lm):
        """Median-of-ratios size factors on the device (preprocessing.py:59-102) -> device array [N]."""
        if self._work is None:
            self._work = DeviceArray(self.ctx, (self.ctx.lib.dsq_size_factors_work_doubles(self.N, self.G),),
                                     np.float64)
        d_sf = self._dvec(self.N)
        d_lm, d_mask = self._sf_inputs(d_lm)
        self._k("size_factors", self.G, "dsq_dev_size_factors", _vp(self.d_raw.ptr), self._count_type, self.N,
                self.G, _vp(d_lm.ptr), _vp(d_mask.ptr) if d_mask is not None else None, _vp(self._work.ptr),
                _vp(d_sf.ptr))
        return self._sf_finish(d_sf)

    def _sf_inputs(self, d_lm):
        """(log means, gene mask or None) the median of ratios runs on: the plain log means of the genes
        without zeros, or — "poscounts", dds.py:655-680 — the log geometric means over the positive counts;
        restricted to the control genes when there are any (dds.py:631-653)."""
        d_mask = None
        if self.size_factors_fit_type == "poscounts":
            d_lm, d_use = self._dvec(self.G), self._dvec(self.G, np.uint8)
            self.ctx.call("dsq_dev_logmeans_poscounts", _vp(self.d_y.ptr), self.ldn, self.N, self.G, _vp(d_lm.ptr),
                          _vp(d_use.ptr))
            d_mask = d_use
            if self._control_mask is not None:
                d_mask = self._up(self._down(d_use, self.G, np.uint8) & self._control_mask, np.uint8)
        elif self._control_mask is not None:
            d_mask = self._up(self._control_mask, np.uint8)
        return d_lm, d_mask

    def _sf_finish(self, d_sf):
        if self.size_factors_fit_type == "poscounts":  # normalise to a geometric mean of 1
            sf = self._down(d_sf, self.N)
            d_sf = self._up(sf / np.exp(np.mean(np.log(sf))))
        return d_sf

    def _fit_trend(self, Gn):
        """Parametric trend on the device (dds.py:1199-1275) -> coeffs[2] or None."""
        d_gw, d_nm = self._last_gw_dev
        return self._run_trend_kernel(d_gw, d_nm, Gn)

    def _run_trend_kernel(self, d_gw, d_nm, n):
        c2, ok, n_outer = (C.c_double * 2)(), C.c_int(0), C.c_int(0)
        d_keep = self._dvec(n, np.uint8)
        self._k("trend_fit", n, "dsq_dev_trend_fit", _vp(d_gw.ptr), _vp(d_nm.ptr), int(n),
                c_double(self.min_disp), c_double(self.max_disp), _vp(d_keep.ptr), c2, C.byref(ok),
                C.byref(n_outer))
        return np.array([c2[0], c2[1]]) if ok.value else None

    def _trend_prior_fused(self, Gn, d_fit):
        """(coeffs or None, squared_logres): dsq_dev_trend_prior - the parametric trend (dds.py:1199-1275), its fitted
        values and the MAD prior (dds.py:866-884) enqueued back to back, one synchronisation."""
        d_gw, d_nm = self._last_gw_dev
        c2, ok, n_outer, sq = (C.c_double * 2)(), C.c_int(0), C.c_int(0), C.c_double()
        d_keep = self._dvec(Gn, np.uint8)
        d_work = self._dvec(self.ctx.lib.dsq_prior_mad_work_doubles(int(Gn)))
        self.ctx.call("dsq_dev_trend_prior", _vp(d_gw.ptr), _vp(d_nm.ptr), int(Gn), c_double(self.min_disp),
                      c_double(self.max_disp), _vp(d_keep.ptr), _vp(d_fit.ptr), _vp(d_work.ptr), c2, C.byref(ok),
                      C.byref(n_outer), C.byref(sq))
        if not ok.value:
            return None, None
        return np.array([c2[0], c2[1]]), float(sq.value)

    def _mean_trend(self, Gn):
        """Mean-based trend (dds.py:1277-1299) over this pipeline's (clipped) genewise dispersions."""
        d_gw, _ = self._last_gw_dev
        gw = np.clip(self._down(d_gw, Gn), self.min_disp, self.max_disp)
        return _trend.mean_trend(gw, self.min_disp)

    def _prior(self, Gn, d_fit, r):
        """(squared_logres, prior_disp_var) (dds.py:866-884); the two medians run on the device."""
        d_gw, _ = self._last_gw_dev
        sq = C.c_double()
        d_work = self._dvec(self.ctx.lib.dsq_prior_mad_work_doubles(int(Gn)))
        self._k("prior_mad", Gn, "dsq_dev_prior_mad", _vp(d_gw.ptr), _vp(d_fit.ptr), Gn,
                c_double(self.min_disp), c_double(self.max_disp), _vp(d_work.ptr), C.byref(sq))
        sq = float(sq.value)
        return sq, float(max(sq - _trigamma((self.N - self.P) / 2), 0.25))

    # ------------------------------------------------------------------ the pipeline
    # One pass = one _Step: the device handles and host scalars the stages hand to each other.  deseq2() runs all
    # stages back to back (the benchmarked path: robust dispersions forked under the tails of the dispersion stages,
    # read-backs speculated on the previous pass's mask); the user-level façade (api.py) opens a step and advances
    # it stage by stage as the reference's fit_* methods are called (dds.py:584-1110), reading vectors in between.
    STAGES = ("open", "size_factors", "genewise", "trend", "map", "lfc", "refit", "finish")

    def deseq2(self, contrast=None, lfc_null=0.0, alt_hypothesis=None, profile=False,
               stop_after_trend=False, stop_after_size_factors=False, size_factors=None) -> DeseqResult:
        """Run size factors -> dispersions -> LFC -> Cook's (+refit) -> Wald.

        The per-gene vectors stay in HBM from the first kernel to the Wald test; the host sees
        scalars (trend coefficients, prior variance), the Cook's flags that decide the refit, and at
        the end one block copy of all result vectors.
        """
        upto = "size_factors" if stop_after_size_factors else ("trend" if stop_after_trend else "finish")
        while True:
            st = self.begin_step(contrast, lfc_null, alt_hypothesis, profile=profile, size_factors=size_factors,
                                 upto=upto)
            try:
                return self.advance(st, upto)
            except _Respeculate:  # the non-zero mask the pass was enqueued on is not this matrix's: once more, in order
                continue

    def begin_step(self, contrast=None, lfc_null=0.0, alt_hypothesis=None, *, profile=False, size_factors=None,
                   upto=None):
        """Open a pass.  ``upto``: the stage the caller will stop at when it is known in advance ("finish": the whole path,
        which lets the robust dispersions and the read-backs overlap the dispersion stages); None: stage by stage."""
        if alt_hypothesis not in ALT:
            raise KeyError(alt_hypothesis)
        if lfc_null < 0 and alt_hypothesis in {"greaterAbs", "lessAbs"}:
            raise ValueError(f"The alternative hypothesis being {alt_hypothesis}, please provide a "
                             f"positive lfc_null value (got {lfc_null}).")
        if contrast is None:
            contrast = np.zeros(self.P)
            contrast[-1] = 1.0
        st = _Step()
        st.contrast = np.ascontiguousarray(contrast, dtype=np.float64)
        st.lfc_null, st.alt, st.profile = float(lfc_null), alt_hypothesis, bool(profile)
        st.size_factors_in = size_factors
        st.whole = upto == "finish"          # the pass is known to run to its end: overlap what can be overlapped
        st.stop_at = upto
        st.r = DeseqResult()
        st.done = "open"
        st.t = [None]
        self._pool_reset()
        st.serial = self._serial
        st.t0 = st.t_last = self._tick(st)
        return st

    def step_alive(self, st):
        """Whether the device buffers of an open pass are still its own (no other pass has recycled the pool since)."""
        return st is not None and getattr(st, "serial", -1) == getattr(self, "_serial", 0)

    def advance(self, st, upto="finish"):
        """Run the stages of an open pass up to and including ``upto``; returns the pass's DeseqResult (complete after
        "finish", otherwise holding what the stages so far publish)."""
        order = self.STAGES
        while order.index(st.done) < order.index(upto):
            nxt = order[order.index(st.done) + 1]
            getattr(self, "_st_" + nxt)(st)
            st.done = nxt
        return st.r

    def _tick(self, st):
        if st.profile:
            self.ctx.sync()
        return time.perf_counter()

    def _lap(self, st, name):
        now = self._tick(st)
        st.r.timings[name] = now - st.t_last
        st.t_last = now

    # ---- size factors (dds.py:692-708) and the compaction to the genes with a count (dds.py:729-731)
    def _st_size_factors(self, st):
        ctx, N, G = self.ctx, self.N, self.G
        r, size_factors = st.r, st.size_factors_in
        d_lm, d_nz = self._dvec(G), self._dvec(G, np.uint8)
        self._k("logmeans", G, "dsq_dev_logmeans", _vp(self.d_y.ptr), self.ldn, N, G, _vp(d_lm.ptr), _vp(d_nz.ptr))
        st.spec = None
        sf = None
        if size_factors is None and self.size_factors_fit_type != "iterative":
            d_sf = self._size_factors(d_lm)
            pred = getattr(self, "_nz_pred", None)
            if pred is not None and st.whole and not (st.profile or self.time_kernels):
                # The host needs the size factors only for the result and for the NaN test below, and the non-zero
                # mask only for the number of genes it compacts to.  Neither changes between two passes over the
                # same counts: read both back asynchronously, enqueue the genewise stage on the previous pass's mask
                # and compare once that stage has synchronised anyway (a mismatch re-runs the pass in order).
                hs = self._host_slab(8 * N + G)
                ctx.call("dsq_d2h_async", _vp(hs.ptr), _vp(d_sf.ptr), C.c_size_t(8 * N))
                ctx.call("dsq_d2h_async", _vp(hs.ptr + 8 * N), _vp(d_nz.ptr), C.c_size_t(G))
                st.spec = (hs.view(0, N, np.float64), hs.view(8 * N, G, np.uint8))
            else:
                sf = self._down(d_sf, N)
            if st.spec is None and np.isnan(sf).any():  # dds.py:682-690
                warnings.warn("Every gene contains at least one zero, cannot compute log geometric means. "
                              "Switching to iterative mode.", UserWarning, stacklevel=3)
                size_factors = "iterative"
        if size_factors is None and self.size_factors_fit_type == "iterative":
            size_factors = "iterative"
        if size_factors is not None:
            if isinstance(size_factors, str):
                from .sizefactors import iterative_size_factors

                self._nz_cache = self._down(d_nz, G, np.uint8).astype(bool)
                size_factors = iterative_size_factors(self)
                self._pool_reset()
                d_lm, d_nz = self._dvec(G), self._dvec(G, np.uint8)
                self._k("logmeans", G, "dsq_dev_logmeans", _vp(self.d_y.ptr), self.ldn, N, G, _vp(d_lm.ptr),
                        _vp(d_nz.ptr))
            sf = np.ascontiguousarray(size_factors, dtype=np.float64)
            d_sf = self._up(sf)
        non_zero = pred.copy() if st.spec is not None else self._down(d_nz, G, np.uint8).astype(bool)
        r.size_factors, r.non_zero = sf, non_zero
        self.d_sf = st.d_sf = d_sf
        st.non_zero = non_zero
        if st.stop_at == "size_factors":
            return
        Gn = st.Gn = int(non_zero.sum())
        st.all_nz = Gn == G
        st.nzi = None if st.all_nz else np.nonzero(non_zero)[0]
        self._lap(st, "size_factors")
        if st.all_nz:
            st.d_ynz = self.d_y
        else:
            d_idx = self._up(st.nzi.astype(np.int32), np.int32)
            st.d_ynz = self._dmat(Gn, np.int32)
            ctx.call("dsq_dev_gather_rows_i32", _vp(self.d_y.ptr), self.ldn, _vp(d_idx.ptr), Gn, N,
                     _vp(st.d_ynz.ptr))
        st.S = self._dev_slab(Gn)
        # mixed designs: the counts in slot order (built once per non-zero mask, outside the steady-state step)
        st.mix_slots = self._mix_slots_for(st.d_ynz, Gn, persistent_key=non_zero) if (self._mix and Gn > 0) else None
        st.robust_done = [False, False]

    # ---- the robust dispersions of the Cook's stage (utils.py:914-960) depend on counts, size factors and design
    # cells only: in a whole pass they run on a side stream underneath the latency-bound kernels of the path (the
    # continuation of the parked dispersion fits, the grid-search pass, the trend / prior kernels on their reserved
    # compute units); stage by stage they are launched by the LFC stage, their first reader.
    # Two parts (row-kernel designs, DSQ_ROBUST_SPLIT): the kernel (0.91 ms at c3) is a little longer than the
    # latency-bound tail of the genewise stage it hides under (continuation, grid pass, trend, prior: 0.73 ms), and a MAP
    # launch beside its end shares every compute unit with it.  Part one - the genes that fit under that tail - is
    # forked inside the genewise fit; part two runs under the SAME tail of the MAP stage (its continuation and grid pass)
    # and is joined before the LFC fit, whose epilogue is the first reader.
    def _robust_cut(self, st):
        split = self._robust_split if (st.whole and self.overlap and self._map_waits_side and st.Gn >= 4096) else 1.0
        return st.Gn if split >= 1.0 else max(1, min(st.Gn - 1, int(st.Gn * split) & ~3))

    def _launch_robust(self, st, part=0, side=True):
        D, ctx, Gn = self.design, self.ctx, st.Gn
        g0, g1 = (0, st.g_cut) if part == 0 else (st.g_cut, Gn)
        st.robust_done[part] = True
        if g1 <= g0:
            return
        side = side and self.overlap
        if side:
            ctx.call("dsq_side_begin")
            self._side_pending = True  # until dsq_side_wait: _pool_reset must not recycle what the side stream writes
        try:
            self._k("robust_disp", g1 - g0, "dsq_dev_robust_disp2", _vp(st.d_ynz.ptr + 4 * self.ldn * g0), self.ldn,
                    _vp(st.d_sf.ptr), _vp(self.d_cell_off.ptr), _vp(self.d_cell_idx.ptr), D.n_cells, int(D.whole),
                    D.max_cell, D.min_cell, self.N, g1 - g0, _vp(st.S["rd"].ptr + 8 * g0))
        except BaseException:
            if side:  # back on the main stream, both streams drained (a shared Context stays usable)
                ctx.call("dsq_side_abort")
                self._side_pending = False
            raise
        if side:
            ctx.call("dsq_side_end")

    def _with_alpha_hook(self, fn, launch):
        """Run ``fn()`` (a dispersion stage) with ``launch()`` fired from inside it, when its full-size kernel is
        enqueued and only the continuation of the parked fits and the grid pass remain (dsq_set_alpha_hook); returns
        whether the hook fired."""
        # (a ctypes callback object sits in a reference cycle of its own and is only freed by the garbage collector: what its
        # closure can reach must not include the pass - whose result vectors are views of a page-locked slab that would then
        # return to the pool late, and the next pass would allocate a fresh 6 MB slab: 0.3 ms per step at c3 - so the
        # launcher travels in a cell that is emptied when the call returns)
        state = {"fired": False, "error": None, "launch": launch}

        def _hook(_arg):
            state["fired"] = True
            try:
                state["launch"]()
            except BaseException as e:  # (a ctypes callback cannot propagate it)
                state["error"] = e

        self._alpha_hook = HOOK_FN(_hook)  # kept alive until the call has returned
        self.ctx.call("dsq_set_alpha_hook", C.cast(self._alpha_hook, C.c_void_p), None)
        try:
            out = fn()
        finally:
            self.ctx.call("dsq_set_alpha_hook", None, None)  # (not fired: no gene reached the fit)
            self._alpha_hook = None
            state["launch"] = None
        if state["error"] is not None:
            raise state["error"]
        return out, state["fired"]

    # ---- genewise dispersions (dds.py:713-797)
    def _st_genewise(self, st):
        ctx, r, S, Gn = self.ctx, st.r, st.S, st.Gn
        st.g_cut = self._robust_cut(st)
        want_robust = st.whole
        early = want_robust and self.overlap and self._robust_early
        # fork point: inside the genewise fit (the hook); without it (profiling mode, DSQ_ROBUST_LATE) after the stage
        mid = want_robust and self.overlap and not early and not self.time_kernels and not self._robust_late
        fit = lambda: self._stage_genewise(st.d_ynz, Gn, st.d_sf, S, self._row_lists_for(st.non_zero),  # noqa: E731
                                           pre_alpha=(lambda: self._launch_robust(st)) if early else None,
                                           mix_slots=st.mix_slots)
        if mid:
            st.mh, fired = self._with_alpha_hook(fit, lambda: self._launch_robust(st))
        else:
            st.mh, fired = fit(), False
        if st.spec is not None:  # the genewise stage has synchronised behind the two read-backs
            if Gn == 0:
                ctx.sync()
            sf = np.array(st.spec[0])
            if np.isnan(sf).any() or not np.array_equal(st.spec[1].view(np.bool_), st.non_zero):
                self._nz_pred = None
                if self._side_pending:
                    ctx.call("dsq_side_abort")
                    self._side_pending = False
                raise _Respeculate()
            r.size_factors = sf
            st.spec = None
        self._last_gw_dev = (S["gw"], S["nm"])  # raw genewise dispersions / normalised means
        if want_robust and not early and not fired:
            self._launch_robust(st)
        self._lap(st, "genewise")

    # ---- trend (dds.py:799-838) + prior (dds.py:840-884): the cross-gene steps
    def _st_trend(self, st):
        r, S, Gn, N, P = st.r, st.S, st.Gn, self.N, self.P
        coeffs = None
        fused_sq = None
        only_trend = st.stop_at == "trend"
        if self.fit_type == "parametric":
            cls = type(self)
            # (a subclass that overrides only the two separate hooks - as the gene-sharded pipeline once did - must not be
            # bypassed by the fused call: it is taken when the class brings its own, or leaves all three alone)
            fused_ok = (cls._trend_prior_fused is not DeseqPipeline._trend_prior_fused
                        or (cls._fit_trend is DeseqPipeline._fit_trend and cls._prior is DeseqPipeline._prior))
            if fused_ok and not (only_trend or self.time_kernels):
                # trend fit, fitted values and prior in one call (one synchronisation instead of two and a launch gap);
                # the gene-sharded pipeline runs the same call on the all-gathered vectors (distributed.py)
                coeffs, fused_sq = self._trend_prior_fused(Gn, S["fit"])
            else:
                coeffs = self._fit_trend(Gn)
            if coeffs is None:
                warnings.warn("The dispersion trend curve fitting did not converge. "
                              "Switching to a mean-based dispersion trend.", UserWarning, stacklevel=3)
        elif self.fit_type != "mean":
            raise NotImplementedError(f"Expected 'parametric' or 'mean' trend curve fit types, received "
                                      f"{self.fit_type}")
        st.coeffs = coeffs
        if coeffs is not None:
            r.trend_coeffs, r.disp_function_type = coeffs, "parametric"
            st.a0, st.a1 = float(coeffs[0]), float(coeffs[1])
        else:
            r.disp_function_type = "mean"
            r.mean_disp = self._mean_trend(Gn)
            st.a0, st.a1 = float(r.mean_disp), 0.0
        if only_trend:  # vst_fit (dds.py:384-438): size factors, genewise dispersions, trend
            self.publish(st, ("nm", "mom", "gw", "gconv"))
            return
        if fused_sq is None or coeffs is None:
            self.ctx.call("dsq_dev_trend_eval", _vp(S["nm"].ptr), Gn, c_double(st.a0), c_double(st.a1), _vp(S["fit"].ptr))
        if (N - P) <= 3:
            warnings.warn("As the residual degrees of freedom is less than 3, the distribution of log "
                          "dispersions is especially asymmetric and likely to be poorly estimated by the MAD.",
                          UserWarning, stacklevel=3)
        if fused_sq is not None and coeffs is not None:
            r.squared_logres = fused_sq
            r.prior_disp_var = float(max(fused_sq - _trigamma((N - P) / 2), 0.25))
        else:
            r.squared_logres, r.prior_disp_var = self._prior(Gn, S["fit"], r)
        self._lap(st, "trend_prior")

    # ---- MAP dispersions + dispersion outliers (dds.py:886-935)
    def _st_map(self, st):
        ctx, r, S, Gn = self.ctx, st.r, st.S, st.Gn
        if self.overlap and self._map_waits_side and self._side_pending:
            # the robust-dispersion kernel of the side stream is (0.9 ms at c3) a little longer than the latency-bound tail
            # it runs under; a MAP launch that starts beside its last 0.15-0.25 ms shares every compute unit with it for
            # its whole life (persistent workgroups) and pays more than the wait costs (A/B: DSQ_MAP_WAIT=0)
            ctx.call("dsq_side_wait")
            self._side_pending = False
        second = st.g_cut < Gn and st.whole
        # The LFC fit of a gene needs that gene's final dispersion only (dds.py:937-984).  In a whole pass the genes whose MAP
        # fit is finished after the stage's full-size launch (~99 %: converged, not parked) start their LFC fit from the hook,
        # on a stream of their own; the continuation of the parked fits, the host round trip and the grid-search pass - the
        # latency-bound 0.3-0.4 ms that used to stand between the two full-size kernels - run beside it, and the rest of the
        # genes follow in a second launch (_st_lfc).
        fork = (self._lfc_overlap and self.overlap and st.whole and not self.time_kernels and not st.profile
                and Gn >= self._lfc_min_genes
                and bool(ctx.lib.dsq_lfc_takes_parts(self.N, self.P, self._cells_arg(), _vp(self._mix) if self._mix else None,
                                                     int(self.design.full_rank))))
        st.lfc_forked = False
        if fork:
            self._lfc_prepare(st)
            st.d_part = self._dvec(Gn, np.uint8)
            # "not finished" in both flag vectors: the stage's full-size launch writes 0 / 1 into the first, its later
            # launches into the second (csrc dsq_alpha_set_late_flags) - the fork selects its genes by the first alone
            st.d_mconv_late = self._dvec(Gn, np.uint8)
            ctx.call("dsq_memset", _vp(S["mconv"].ptr), 0xFF, C.c_size_t(Gn))
            ctx.call("dsq_memset", _vp(st.d_mconv_late.ptr), 0xFF, C.c_size_t(Gn))
            # (the first launch's small operations here, on an idle device, instead of beside the stage's tail)
            ctx.call("dsq_lfc_prepare", _vp(st.d_sf.ptr), self.N, _vp(st.wald_args[0].ctypes.data),
                     _vp(st.wald_args[1].ctypes.data), self.P)

        def launch():
            # the fork first: its launch starts behind the robust dispersions the side stream holds NOW (part one); the genes
            # of part two - computed beside it - take the second launch (ready_limit)
            if fork:
                self._fork_lfc(st)
            if second or not st.robust_done[1]:
                self._launch_robust(st, 1)

        fit = lambda: self._stage_map(st.d_ynz, st.mh, Gn, st.d_sf, r.prior_disp_var, r.squared_logres, S,  # noqa: E731
                                      fork=(st.d_part, st.d_mconv_late, lambda: st.lfc_forked) if fork else None)
        if (second or fork) and not self.time_kernels:
            _, fired = self._with_alpha_hook(fit, launch)
        else:
            fit()
            fired = False
        if second and not fired:
            self._launch_robust(st, 1)
        self._lap(st, "MAP")

    def _lfc_prepare(self, st):
        """Arguments and buffers of the LFC fit (the Wald test's ridge and contrast, the Cook's layer) - once per pass."""
        P, Gn = self.P, st.Gn
        st.cutoff = self._cooks_cutoff
        ridge = np.ascontiguousarray(np.diag(np.repeat(1e-6, P)))
        st.wald_args = (ridge, st.contrast, float(np.log(2) * st.lfc_null), ALT[st.alt])
        st.cld = cld = self._cooks_ld()
        st.d_cooks = self._pooled((max(Gn, 1), cld), np.float64, ld=cld) if cld else self._dmat(Gn)

    def _fork_lfc(self, st):
        """From inside the MAP stage, its full-size launch enqueued (the hook): final dispersions of the finished genes and
        their LFC fit on the forked stream (csrc dsq_lfc_fork_begin: behind that launch and the robust dispersions)."""
        ctx, r, S, Gn = self.ctx, st.r, st.S, st.Gn
        ctx.call("dsq_lfc_fork_begin")
        self._lfc_forked = True  # until the second launch has joined (_pool_reset: dsq_side_abort)
        try:
            ctx.call("dsq_dev_select_dispersions_part", _vp(S["gw"].ptr), _vp(S["map"].ptr), _vp(S["fit"].ptr), Gn,
                     c_double(self.min_disp), c_double(self.max_disp), c_double(r.squared_logres), _vp(S["disp"].ptr),
                     _vp(S["outl"].ptr), _vp(S["mconv"].ptr), None, _vp(st.d_part.ptr), 1,
                     int(Gn if st.robust_done[1] else st.g_cut))
            st.lfc_bufs = self._stage_lfc(st.d_ynz, Gn, st.d_sf, S, st.wald_args, cooks=(S["rd"], st.cutoff, st.d_cooks),
                                          mix_slots=st.mix_slots, part=(st.d_part, 1, 1))
            ctx.call("dsq_lfc_fork_end")
        except BaseException:
            ctx.call("dsq_side_abort")  # back on the main stream, the forked one drained
            self._lfc_forked = False
            raise
        st.lfc_forked = True
        self.lfc_forks += 1

    # ---- LFC (dds.py:937-984) with the per-sample half of Cook's (dds.py:986-1040) and the Wald statistics
    # (ds.py:303-360) in its epilogue: mu and the hat diagonal are consumed from registers
    def _st_lfc(self, st):
        ctx, D, S, Gn, G = self.ctx, self.design, st.S, st.Gn, self.G
        for part in (0, 1):  # stage by stage: nobody has launched the robust dispersions yet
            if not st.robust_done[part]:
                self._launch_robust(st, part, side=False)
        forked = getattr(st, "lfc_forked", False)
        if not forked:
            self._lfc_prepare(st)
        cutoff, cld = st.cutoff, st.cld
        if self.overlap and self._side_pending:
            ctx.call("dsq_side_wait")
            self._side_pending = False
        if forked:  # the genes the MAP stage's tail finished; the call joins the first launch and rescues for both
            d_mu, d_hat = self._stage_lfc(st.d_ynz, Gn, st.d_sf, S, st.wald_args, cooks=(S["rd"], cutoff, st.d_cooks),
                                          mix_slots=st.mix_slots, bufs=st.lfc_bufs, part=(st.d_part, 0, 2))
            self._lfc_forked = False
        else:
            d_mu, d_hat = self._stage_lfc(st.d_ynz, Gn, st.d_sf, S, st.wald_args, cooks=(S["rd"], cutoff, st.d_cooks),
                                          mix_slots=st.mix_slots)
        st.want_refit = self.refit_cooks and D.replaceable.sum() > 0
        # Everything in S is final now except the rows the refit will replace: in a whole pass the block copy of the
        # result vectors starts here, on the side stream, and runs underneath the refit's kernels; the host patches
        # the (few) refitted rows afterwards from the refit's own small result block.  The main stream only carries
        # the one flag vector that decides the refit, so waiting for it does not wait for the block copy.
        st.flags_tok = self._fetch_begin(S, ["any_all"]) if st.want_refit else None
        st.slab_tok = None
        if st.whole:
            self._begin_block_copy(st)
        if st.all_nz and getattr(self, "_arange_G", None) is None:
            self._arange_G = np.arange(G)
        # the layers of THIS fit: cooks is always materialised, mu and the hat diagonals only with keep_layers,
        # otherwise layer() rebuilds them on demand from the fit's coefficients / dispersions (S is not patched by the
        # refit any more, so its vectors ARE those of this fit)
        self.layers = {"nz_idx": self._arange_G if st.all_nz else st.nzi, "mu_LFC": d_mu, "hat_diagonals": d_hat,
                       "cooks": st.d_cooks, "_cooks_ld": cld, "_fit": (st.d_ynz, st.d_sf, S["beta"], S["disp"], Gn)}
        self._lap(st, "LFC_cooks_wald")

    def _begin_block_copy(self, st):
        if self.overlap:
            self.ctx.call("dsq_side_begin")
            self._side_pending = True
            try:
                st.slab_tok = self._fetch_begin(st.S)
            finally:
                self.ctx.call("dsq_side_end")
        else:
            st.slab_tok = self._fetch_begin(st.S)

    # ---- refit (dds.py:1042-1064, 1301-1458): a small sub-problem on the replaced counts.  One host round trip
    # (which genes?), then the whole chain - replacement, MoM, mu_hat, genewise fit, trend value, MAP fit, LFC fit,
    # Wald - is enqueued without another: second passes are launched for the whole batch as a capacity
    # (dsq_set_deferred), and genes that became all-zero keep their original row on the device (their results are
    # discarded by the flag that comes back with the results).
    def _st_refit(self, st):
        ctx, r, S, Gn, N = self.ctx, st.r, st.S, st.Gn, self.N
        st.replaced_nz = np.zeros(Gn, dtype=bool)
        st.patch = None
        if getattr(st, "force_refit", False) and not st.want_refit and self.design.replaceable.sum() > 0:
            # DeseqDataSet.refit() called by hand on a data set built with refit_cooks=False (dds.py:1042-1064)
            st.want_refit, st.flags_tok = True, self._fetch_begin(S, ["any_all"])
        if st.want_refit:
            ctx.sync()
            st.replaced_nz = self._fetch_end(st.flags_tok)["any_all"].astype(bool)  # idx.any(axis=0), dds.py:1325-1326
            rp = np.nonzero(st.replaced_nz)[0]
            Gr = len(rp)
            if Gr > 0:
                d_rp = self._up(rp.astype(np.int32), np.int32)
                d_ysub = self._dmat(Gr, np.int32)
                S2 = self._dev_slab(Gr)
                d_az = S2["naz"]  # its own field: no stage of the sub-problem writes it
                ctx.call("dsq_dev_replace_outliers2", _vp(st.d_ynz.ptr), _vp(st.d_cooks.ptr), self.ldn, _vp(st.d_sf.ptr),
                         _vp(self.d_flags.ptr), _vp(d_rp.ptr), Gr, N, c_double(st.cutoff), _vp(d_ysub.ptr),
                         _vp(d_az.ptr), st.cld, _vp(self._mix) if st.cld else None)
                deferred = not (st.profile or self.time_kernels or self.collect_nfev)
                if deferred:
                    ctx.call("dsq_set_deferred", 1)
                sub_slots = self._mix_slots_for(d_ysub, Gr) if self._mix else None  # (the replaced counts' own copy)
                try:
                    s_mu = self._stage_genewise(d_ysub, Gr, st.d_sf, S2, mix_slots=sub_slots)
                    ctx.call("dsq_dev_trend_eval", _vp(S2["nm"].ptr), Gr, c_double(st.a0), c_double(st.a1),
                             _vp(S2["fit"].ptr))
                    self._stage_map(d_ysub, s_mu, Gr, st.d_sf, r.prior_disp_var, r.squared_logres, S2)
                    self._stage_lfc(d_ysub, Gr, st.d_sf, S2, st.wald_args, mix_slots=sub_slots)
                finally:
                    if deferred:
                        ctx.call("dsq_set_deferred", 0)
                st.patch = (rp, self._fetch_begin(S2))  # the sub-problem's whole (small) result block in one copy
        self._lap(st, "refit")

    # ---- host view of the results (reference field names), scattered to all G genes
    def _st_finish(self, st):
        ctx, r, G, Gn = self.ctx, st.r, self.G, st.Gn
        if st.slab_tok is None:
            self._begin_block_copy(st)
        # the Wald statistics (ds.py:303-360) came out of the LFC fits' epilogues
        if self.overlap:
            ctx.call("dsq_side_wait")
            self._side_pending = False
        ctx.sync()
        H = self._fetch_end(st.slab_tok)
        self._lap(st, "wald")
        all_nz, nzi = st.all_nz, st.nzi

        def full(v, fill=np.nan, dtype=None):
            if all_nz:
                return v if dtype is None else v.astype(dtype)
            out = np.full((G,) + v.shape[1:], fill, dtype=dtype or v.dtype)
            out[nzi] = v
            return out

        replaced_nz = st.replaced_nz
        refitted_nz = np.zeros(Gn, dtype=bool)
        new_zero_nz = np.zeros(Gn, dtype=bool)
        gw = H["gw"]  # clipped to [min_disp, max_disp] on the device (dds.py:792-794; k_select_disp)
        nm, fit, disp, beta = H["nm"], H["fit"], H["disp"], H["beta"]
        pv, stt, se = H["p"], H["stat"], H["se"]
        if st.patch is not None:  # dds.py:1368-1458: the refitted genes take their new values
            rp, h2 = st.patch[0], self._fetch_end(st.patch[1])
            naz = h2["naz"].astype(bool)  # all counts zero after the replacement (dds.py:1368-1383)
            new_zero_nz[rp[naz]] = True
            refitted_nz[rp[~naz]] = True
            rf, k = rp[~naz], np.nonzero(~naz)[0]
            nm[rf], fit[rf], disp[rf], beta[rf] = h2["nm"][k], h2["fit"][k], h2["disp"][k], h2["beta"][k]
            gw[rf] = np.clip(h2["gw"][k], self.min_disp, self.max_disp)
            pv[rf], stt[rf], se[rf] = h2["p"][k], h2["stat"][k], h2["se"][k]
            if naz.any():
                zi = rp[naz]
                nm[zi], beta[zi] = 0.0, 0.0            # dds.py:1380-1383
                se[zi], stt[zi], pv[zi] = 0.0, 0.0, 1.0  # ds.py:357-360
        r.normed_means = full(nm, fill=0.0)  # all-zero genes have normed mean 0 (dds.py:708)
        r.mom_dispersions = full(H["mom"])
        r.genewise_dispersions = full(gw)
        r.genewise_converged = full(H["gconv"].astype(float))
        r.fitted_dispersions = full(fit) if st.coeffs is not None else np.full(G, r.mean_disp)
        r.MAP_dispersions = full(H["map"])  # clipped on the device as well (dds.py:905-907)
        r.MAP_converged = full(H["mconv"].astype(float))
        r.outlier_genes = full(H["outl"].view(np.bool_), fill=False)
        r.dispersions = full(disp)
        r.LFC = full(beta)
        r.LFC_converged = full(H["lconv"].astype(float))
        r.replaced = full(replaced_nz, fill=False)
        r.refitted = full(refitted_nz, fill=False)
        r.new_all_zeroes = full(new_zero_nz, fill=False)
        # ---- cooks_outlier (dds.py:1066-1110)
        any_use, any_use_nr = H["any_use"].view(np.bool_), H["any_use_nr"].view(np.bool_)
        co_nz = np.where(refitted_nz, any_use_nr, any_use) if (self.refit_cooks and refitted_nz.any()) else any_use
        r.cooks_outlier = full(co_nz & H["few_above"].view(np.bool_), fill=False)
        r.pvalue, r.stat, r.lfcSE = full(pv), full(stt), full(se)
        self._lap(st, "assemble")
        r.timings["total"] = st.t_last - st.t0
        if not self.keep_cooks:
            self.layers = {}
        self._nz_pred = st.non_zero
        st.slab_tok = st.flags_tok = st.patch = None  # (the result's arrays are the only views of the host slabs left)

    def publish(self, st, names):
        """Host copies of per-gene vectors of an OPEN pass, scattered to all G genes and stored in its DeseqResult under the
        reference's field names (what a fit_* method of the façade shows after its stage): names of the result slab
        ("nm", "mom", "gw", "gconv", "fit", "map", "mconv", "disp", "outl", "beta", "lconv")."""
        r, G = st.r, self.G
        H = self._fetch(st.S, list(names))

        def full(v, fill=np.nan, dtype=None):
            v = np.array(v, dtype=dtype or (float if v.dtype == np.uint8 else v.dtype))
            if st.all_nz:
                return v
            out = np.full((G,) + v.shape[1:], fill, dtype=v.dtype)
            out[st.nzi] = v
            return out

        clip = lambda v: np.clip(v, self.min_disp, self.max_disp)  # noqa: E731
        for k in names:
            if k == "nm":
                r.normed_means = full(H[k], 0.0)
            elif k == "mom":
                r.mom_dispersions = full(H[k])
            elif k == "gw":  # raw on the device until the selection kernel of the MAP stage has clipped it in place
                r.genewise_dispersions = full(clip(H[k]))
            elif k == "gconv":
                r.genewise_converged = full(H[k])
            elif k == "fit":
                r.fitted_dispersions = full(H[k]) if st.coeffs is not None else np.full(G, r.mean_disp)
            elif k == "map":
                r.MAP_dispersions = full(clip(H[k]))
            elif k == "mconv":
                r.MAP_converged = full(H[k])
            elif k == "disp":
                r.dispersions = full(H[k])
            elif k == "outl":
                r.outlier_genes = full(H[k].view(np.bool_), False, dtype=bool)
            elif k == "beta":
                r.LFC = full(H[k])
            elif k == "lconv":
                r.LFC_converged = full(H[k])
            else:
                raise KeyError(k)
        return r

    def mu_hat_host(self, st):
        """layers["_mu_hat"] of an open pass (dds.py:747-771), N x G with NaN columns for the all-zero genes: the matrix
        when the pass materialised it, else rebuilt on the host from what the dispersion kernels rebuild it from - the
        per-gene OLS coefficients (max(sf * X coef, min_mu), utils.py:682-715) or the IRLS coefficients
        (sf * exp(X beta), unclamped as utils.py:435-437 returns it)."""
        mh, Gn = st.mh, st.Gn
        sf = np.asarray(st.r.size_factors, dtype=float)
        if mh.d_mu is not None:
            rows = self.ctx.d2h_rows(mh.d_mu.ptr, Gn, self.N, self.ldn).T
        elif mh.d_coef is not None:
            coef = self._down(mh.d_coef, Gn * self.P).reshape(Gn, self.P)
            rows = np.maximum(sf[:, None] * (self.design.X @ coef.T), self.min_mu)
        else:
            beta = self._down(mh.d_beta_fit, Gn * self.P).reshape(Gn, self.P)
            with np.errstate(over="ignore"):
                rows = sf[:, None] * np.exp(self.design.X @ beta.T)
        if st.all_nz:
            return rows
        out = np.full((self.N, self.G), np.nan)
        out[:, st.nzi] = rows
        return out

    def set_dispersions(self, st, dispersions):
        """Replace the final dispersions of an open pass by the caller's (a user who edits var["dispersions"] between
        fit_MAP_dispersions() and fit_LFC(), as the reference's fields allow): G values, those of the genes with counts
        go to the device."""
        d = np.ascontiguousarray(np.asarray(dispersions, dtype=np.float64)[st.non_zero])
        if d.size:
            self.ctx.h2d(st.S["disp"].ptr, d)

    def vst_transform(self, size_factors, trend_coeffs=None, mean_disp=None):
        """Variance-stabilised counts N x G (dds.py:440-514) from the resident raw counts."""
        ctx = self.ctx
        d_sf = DeviceArray.from_host(ctx, np.ascontiguousarray(size_factors, dtype=np.float64))
        d_out = DeviceArray(ctx, (self.N, self.G), np.float64)
        if trend_coeffs is not None:
            mode, a0, a1 = 0, float(trend_coeffs[0]), float(trend_coeffs[1])
        else:
            mode, a0, a1 = 1, float(mean_disp), 0.0
        ctx.call("dsq_dev_vst", _vp(self.d_raw.ptr), self._count_type, self.N, self.G, _vp(d_sf.ptr), mode,
                 c_double(a0), c_double(a1), _vp(d_out.ptr))
        return d_out.to_host()

    def vst_transform_new(self, counts, logmeans, filtered, trend_coeffs=None, mean_disp=None):
        """Variance-stabilised NEW samples (M x G): their size factors are the medians of
        log(count) - training logmeans over the training-usable genes (preprocessing.py:59-102 as called from
        dds.py:471-484), then the same closed form."""
        from ._lib import I32, I64

        ctx = self.ctx
        counts = np.ascontiguousarray(counts if counts.dtype in (np.int32, np.int64) else counts.astype(np.int64))
        M, G = counts.shape
        if G != self.G:
            raise ValueError("new counts must have the genes of the fitted dataset")
        ct = I32 if counts.dtype == np.int32 else I64
        d_c = DeviceArray.from_host(ctx, counts)
        lm = np.where(np.asarray(filtered, dtype=bool), np.asarray(logmeans, dtype=float), -np.inf)
        d_lm = DeviceArray.from_host(ctx, np.ascontiguousarray(lm))
        d_work = DeviceArray(ctx, (ctx.lib.dsq_size_factors_work_doubles(M, G),), np.float64)
        d_sf = DeviceArray(ctx, (M,), np.float64)
        ctx.call("dsq_dev_size_factors_new", _vp(d_c.ptr), ct, M, G, _vp(d_lm.ptr), None, _vp(d_work.ptr), _vp(d_sf.ptr))
        d_out = DeviceArray(ctx, (M, G), np.float64)
        if trend_coeffs is not None:
            mode, a0, a1 = 0, float(trend_coeffs[0]), float(trend_coeffs[1])
        else:
            mode, a0, a1 = 1, float(mean_disp), 0.0
        ctx.call("dsq_dev_vst", _vp(d_c.ptr), ct, M, G, _vp(d_sf.ptr), mode, c_double(a0), c_double(a1), _vp(d_out.ptr))
        return d_out.to_host()

    def wald(self, res: DeseqResult, contrast, lfc_null=0.0, alt_hypothesis=None, lfc=None, ridge=None):
        """Wald test only (ds.py:303-360) on the dispersions / LFCs of ``res`` (e.g. another contrast or
        alternative hypothesis after ``deseq2()``; ``lfc``: other coefficients, e.g. shrunk ones; ``ridge``: other
        ridge matrix, ds.py:326-334).  Returns (pvalue, stat, lfcSE), NaN for all-zero genes."""
        if alt_hypothesis not in ALT:
            raise KeyError(alt_hypothesis)
        if lfc_null < 0 and alt_hypothesis in {"greaterAbs", "lessAbs"}:
            raise ValueError(f"The alternative hypothesis being {alt_hypothesis}, please provide a "
                             f"positive lfc_null value (got {lfc_null}).")
        G, N, P, D = self.G, self.N, self.P, self.design
        contrast = np.ascontiguousarray(contrast, dtype=np.float64)
        ridge = np.ascontiguousarray(np.diag(np.repeat(1e-6, P)) if ridge is None else ridge, dtype=np.float64)
        ctx = self.ctx
        d_sf = DeviceArray.from_host(ctx, np.ascontiguousarray(res.size_factors, dtype=np.float64))
        d_beta = DeviceArray.from_host(ctx, np.ascontiguousarray(res.LFC if lfc is None else lfc, dtype=np.float64))
        d_disp = DeviceArray.from_host(ctx, np.ascontiguousarray(res.dispersions, dtype=np.float64))
        d_out = DeviceArray(ctx, (3 * G,), np.float64)
        ctx.call("dsq_dev_wald", None, self.ldn, _vp(d_sf.ptr), _vp(self.d_Xt.ptr), D.ldx, N, G, P, _vp(d_disp.ptr),
                 _vp(d_beta.ptr), _vp(ridge.ctypes.data), _vp(contrast.ctypes.data), c_double(np.log(2) * lfc_null),
                 ALT[alt_hypothesis], _vp(d_out.ptr), _vp(d_out.ptr + 8 * G), _vp(d_out.ptr + 16 * G))
        o = d_out.to_host()
        pv, st, se = o[:G].copy(), o[G:2 * G].copy(), o[2 * G:].copy()
        z = np.asarray(res.new_all_zeroes, dtype=bool)
        if z.any():  # ds.py:357-360
            se[z], st[z], pv[z] = 0.0, 0.0, 1.0
        return pv, st, se

    def close(self):
        """Release the pooled device buffers."""
        if getattr(self, "_side_pending", False) or getattr(self, "_lfc_forked", False):
            self.ctx.call("dsq_side_abort")
            self._side_pending = False
            self._lfc_forked = False
        for _cap, ptr in self._pool_free + self._pool_used:
            self.ctx.free(ptr)
        self._pool_free, self._pool_used = [], []
        if getattr(self, "_mix_ys_cache", None) is not None:
            self._mix_ys_cache[1].free(); self._mix_ys_cache[2].free()
            self._mix_ys_cache = None
        if getattr(self, "_mix", None):
            self.ctx.lib.dsq_mix_destroy(_vp(self._mix))
            self._mix = None
        self._pinned.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ lazy N x G layers
    def layer(self, name):
        """Fetch an N x G layer ("mu_LFC", "hat_diagonals", "cooks") to the host (NaN for zero genes)."""
        d = self.layers[name]
        if d is None:  # mu / hat diagonals of the LFC fit were consumed in its epilogue: rebuild from beta
            d_y, d_sf, d_b, d_d, Gn = self.layers["_fit"]
            D = self.design
            self.layers["mu_LFC"], self.layers["hat_diagonals"] = self._dmat(Gn), self._dmat(Gn)
            self.ctx.call("dsq_dev_irls_layers", _vp(d_y.ptr), self.ldn, _vp(d_sf.ptr), _vp(self.d_Xt.ptr), D.ldx,
                          self.N, Gn, self.P, _vp(d_d.ptr), _vp(d_b.ptr), c_double(self.min_mu),
                          _vp(self.layers["mu_LFC"].ptr), _vp(self.layers["hat_diagonals"].ptr))
            d = self.layers[name]
        nzi = self.layers["nz_idx"]
        if name == "cooks" and self.layers.get("_cooks_ld"):  # slot order on the device: back to sample order here
            cld = self.layers["_cooks_ld"]
            rows = self.ctx.d2h_rows(d.ptr, len(nzi), cld, cld)[:, self._slot_of]
        else:
            rows = self.ctx.d2h_rows(d.ptr, len(nzi), self.N, self.ldn)
        out = np.full((self.N, self.G), np.nan)
        out[:, nzi] = rows.T
        return out


def deseq2(counts, design_matrix, contrast=None, *, device=0, ctx=None, lfc_null=0.0, alt_hypothesis=None,
           **kw) -> DeseqResult:
    """One-shot convenience wrapper: upload, run, return the result vectors."""
    pipe = DeseqPipeline(counts, design_matrix, ctx=ctx, device=device, **kw)
    return pipe.deseq2(contrast=contrast, lfc_null=lfc_null, alt_hypothesis=alt_hypothesis)
