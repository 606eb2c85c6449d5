"""AnnData-free façade with the reference's user-level surface.

``DeseqDataSet`` / ``DeseqStats`` mirror the parts of ``pydeseq2.dds.DeseqDataSet`` and
``pydeseq2.ds.DeseqStats`` a typical analysis touches — constructor arguments, ``deseq2()``,
``summary()``, ``lfc_shrink()``, and the field names written by the path (``obs["size_factors"]``,
``var["dispersions"]``, ``varm["LFC"]``, ``uns["trend_coeffs"]``, ``results_df`` …) —
on top of the device pipeline, without ``anndata`` or ``formulaic``.  Designs are formulas of metadata
columns (``"~group + condition"``, ``"~group*condition"``; object/category columns are treatment-coded
with the first sorted level as reference, numeric columns enter as they are) or an explicit design matrix.
"""
from __future__ import annotations

import copy as _copy
import re
import warnings

import numpy as np
import pandas as pd

from . import summary as _summary
from ._design import DesignPack
from ._lib import Context
from .pipeline import DeseqPipeline


def build_design(metadata: pd.DataFrame, design, ref_level=None) -> pd.DataFrame:
    """Design matrix for a formula of metadata columns: main effects (same columns and names ``formulaic``
    produces: ``Intercept``, ``factor[T.level]`` …, continuous covariates under their own name) and
    interactions ``a:b`` / ``a*b`` (products of the main-effect columns, named ``a[T.x]:b[T.y]``, after all
    main effects).
    ``ref_level = [factor, level]`` makes ``level`` the reference of ``factor`` instead of its first
    sorted level."""
    if isinstance(design, pd.DataFrame):
        return design.astype(float)
    if not isinstance(design, str):
        X = np.asarray(design, dtype=float)
        return pd.DataFrame(X, index=metadata.index, columns=[f"x{j}" for j in range(X.shape[1])])
    rhs = design.strip()
    if not rhs.startswith("~"):
        raise ValueError("design must be a formula starting with '~' (e.g. '~condition') or a matrix")
    terms = []
    for t in (s.strip() for s in rhs[1:].split("+")):
        if not t:
            continue
        if "*" in t:  # a*b = a + b + a:b
            parts = [s.strip() for s in t.split("*")]
            terms.extend(parts)
            terms.append(":".join(parts))
        else:
            terms.append(t)
    # main effects first, then interactions by degree (the order formulaic uses), duplicates dropped
    seen, ordered = set(), []
    for t in sorted(terms, key=lambda s: s.count(":")):
        if t not in seen:
            seen.add(t)
            ordered.append(t)

    def expand(name):
        """Columns of one variable: treatment-coded levels of a factor or the numeric covariate itself."""
        if not re.fullmatch(r"[A-Za-z_][A-Za-z0-9_.]*", name):
            raise NotImplementedError(f"unsupported design term {name!r}: terms are metadata columns, their "
                                      f"interactions (a:b, a*b), 0/1 for the intercept")
        if name not in metadata.columns:
            raise KeyError(f"design term {name!r} is not a metadata column")
        col = metadata[name]
        if col.isna().any():
            raise ValueError("NaNs are not allowed in the design factors.")
        if col.dtype.kind in "OUSb" or str(col.dtype) == "category":
            levels = sorted(col.astype(str).unique())
            if ref_level is not None and ref_level[0] == name:
                if str(ref_level[1]) not in levels:
                    raise KeyError(f"ref_level: {ref_level[1]!r} is not a level of {name!r}")
                levels.remove(str(ref_level[1]))
                levels.insert(0, str(ref_level[1]))
            return [(f"{name}[T.{lv}]", (col.astype(str) == lv).to_numpy().astype(float)) for lv in levels[1:]]
        return [(name, col.to_numpy().astype(float))]

    cols = {}
    intercept = True
    for t in ordered:
        if t == "1":
            continue
        if t in ("0", "-1"):
            intercept = False
            continue
        combos = [("", np.ones(len(metadata)))]
        for var in t.split(":"):
            combos = [((a + ":" if a else "") + b, va * vb) for a, va in combos for b, vb in expand(var.strip())]
        for name, v in combos:
            cols[name] = v
    out = pd.DataFrame(cols, index=metadata.index)
    if intercept:
        out.insert(0, "Intercept", 1.0)
    return out


def check_counts(counts) -> None:
    """Non-negative integers only, no NaNs (what ``utils.test_valid_counts`` enforces, utils.py:110-133; same
    messages).  Runs on the host before anything is uploaded."""
    arr = counts.to_numpy() if isinstance(counts, pd.DataFrame) else np.asarray(counts)
    if not np.issubdtype(arr.dtype, np.number):
        if isinstance(counts, pd.DataFrame) and counts.isna().any().any():
            raise ValueError("NaNs are not allowed in the count matrix.")
        raise ValueError("The count matrix should only contain numbers.")
    if arr.dtype.kind == "f":
        if np.isnan(arr).any():
            raise ValueError("NaNs are not allowed in the count matrix.")
        if (arr % 1 != 0).any():
            raise ValueError("The count matrix should only contain integers.")
    if (arr < 0).any():
        raise ValueError("The count matrix should only contain non-negative values.")


class DeseqDataSet:
    """Counts + metadata + design, fitted on the GPU (cf. ``pydeseq2.dds.DeseqDataSet``, dds.py:206-340).

    The fit runs as ONE open pass of the device pipeline (``DeseqPipeline.begin_step`` / ``advance``): ``deseq2()`` runs
    it through, the stage-wise methods ``fit_size_factors`` … ``refit`` (dds.py:584-1110) advance it one stage at a
    time with the reference's lazy prerequisite chaining (a stage whose input field is missing runs the stage that
    writes it first) and publish the reference's fields after each stage.  The pipeline itself is created on first use,
    so constructing, slicing (``dds[:, genes]``), copying and pickling a data set never touch the GPU; a pickled or
    copied data set carries its host fields and rebuilds the device state when the next stage is asked for.
    """

    _PIPE_KEYS = ("min_mu", "min_disp", "max_disp", "refit_cooks", "min_replicates", "beta_tol", "fit_type",
                  "size_factors_fit_type")

    def __init__(self, *, counts: pd.DataFrame = None, metadata: pd.DataFrame = None, adata=None, design="~condition",
                 design_factors=None,
                 continuous_factors=None, ref_level=None, refit_cooks=True,
                 min_mu=0.5, min_disp=1e-8, max_disp=10.0, beta_tol=1e-8, min_replicates=7, fit_type="parametric",
                 size_factors_fit_type="ratio", control_genes=None, device=0, ctx: Context | None = None, quiet=True,
                 n_cpus=None, inference=None, low_memory=False):
        # n_cpus: accepted for signature compatibility (dds.py:206-229; the engine is the GPU pipeline).  inference: an
        # object of the plug-in interface - a HipInference lends its device context, anything else is ignored.
        # design_factors (+ continuous_factors): the reference's older way to spell an additive design.
        pre_var = None
        if adata is not None:
            # an AnnData-like object (dds.py:231-249; anything with X / obs / var / obs_names / var_names): its counts and
            # metadata are taken, fields it already carries in .var stay until a stage overwrites them
            if counts is not None or metadata is not None:
                raise ValueError("adata was provided; do not pass counts or metadata.")
            xa = adata.X.toarray() if hasattr(adata.X, "toarray") else np.asarray(adata.X)
            counts = pd.DataFrame(xa, index=adata.obs_names, columns=adata.var_names)
            metadata = pd.DataFrame(adata.obs)
            pre_var = pd.DataFrame(adata.var) if getattr(adata, "var", None) is not None else None
        elif counts is None or metadata is None:
            raise ValueError("Either adata or both counts and metadata arguments must be provided.")
        if design_factors is not None:
            fac = [design_factors] if isinstance(design_factors, str) else list(design_factors)
            design = "~" + " + ".join(fac)
            for cf in continuous_factors or []:
                metadata = metadata.copy()
                metadata[cf] = metadata[cf].astype(float)
        named = isinstance(counts, pd.DataFrame)
        if not named:
            counts = pd.DataFrame(np.asarray(counts))
        if counts.shape[0] != metadata.shape[0]:
            raise ValueError("counts (samples x genes) and metadata disagree on the number of samples")
        check_counts(counts)
        same = set(counts.index) == set(metadata.index)
        if named and not same:  # AnnData refuses this in the reference (tests/test_edge_cases.py::test_indexes)
            raise ValueError("The count matrix and the metadata should have the same sample index.")
        self.obs = metadata.loc[counts.index].copy() if same else metadata.copy()
        if isinstance(design, pd.DataFrame) and not (len(design) == len(self.obs)
                                                       and (design.index == self.obs.index).all()):
            raise ValueError("The design matrix and the metadata should have the same sample index.")
        self.obs_names, self.var_names = counts.index, counts.columns
        self.X = counts.to_numpy()
        if self.X.size and np.issubdtype(self.X.dtype, np.number) and float(self.X.max()) >= 2.0 ** 31:
            raise ValueError("The count matrix should only contain non-negative integers below 2^31.")
        self.design = design
        self._ref_level = ref_level
        dm = build_design(self.obs, design, ref_level)
        self.obsm = _LazyObsm(self, {"design_matrix": dm})
        DesignPack(dm.to_numpy(), min_replicates)  # (host-only: refuses a design wider than the engine takes, before any GPU work)
        if np.linalg.matrix_rank(dm.to_numpy()) < dm.shape[1]:  # dds.py:1550-1563
            warnings.warn("The design matrix is not full rank, so the model cannot be fitted, but some operations "
                          "like design-free VST remain possible. To perform differential expression analysis, "
                          "please remove the design variables that are linear combinations of others.",
                          UserWarning, stacklevel=2)
        self.var = pd.DataFrame(index=self.var_names)
        if pre_var is not None and len(pre_var.columns):
            self.var = pre_var.set_axis(self.var_names).copy()
        self.varm, self.layers, self.uns = {}, _LazyLayers(self), {}
        self.refit_cooks, self.fit_type, self.quiet = bool(refit_cooks), fit_type, quiet
        self.min_mu, self.min_disp, self.max_disp = min_mu, min_disp, max(max_disp, self.X.shape[0])  # dds.py:312
        self.beta_tol, self.min_replicates = beta_tol, min_replicates
        self.size_factors_fit_type = size_factors_fit_type
        self.low_memory = bool(low_memory)
        self.control_genes = control_genes
        self._control_genes = self._control_index(control_genes)
        self.inference = inference
        if ctx is None and inference is not None and isinstance(getattr(inference, "ctx", None), Context):
            ctx = inference.ctx
        self._ctx, self._device = ctx, device
        self._pipe_obj, self._step, self._res = None, None, None
        self._sf_published = None

    # ------------------------------------------------------------------ container surface (AnnData's, as far as the path uses it)
    @property
    def n_obs(self):
        return self.X.shape[0]

    @property
    def n_vars(self):
        return self.X.shape[1]

    @property
    def shape(self):
        return self.X.shape

    def _control_index(self, control_genes):
        """names, integer positions or a boolean mask (dds.py:640-650) -> what the pipeline takes."""
        if control_genes is None:
            return None
        cg = np.asarray(control_genes)
        if cg.dtype.kind in "OUS":
            idx = self.var_names.get_indexer(cg)
            if (idx < 0).any():
                raise KeyError("control_genes: unknown gene name")
            return idx
        return cg

    def _index(self, key, names, n):
        """One axis of ``dds[rows, cols]``: slice, names, integer positions or a boolean mask -> integer positions."""
        if isinstance(key, slice):
            return np.arange(n)[key]
        if isinstance(key, (str, bytes)):
            key = [key]
        k = np.asarray(key.to_numpy() if hasattr(key, "to_numpy") else key)
        if k.dtype == bool:
            if k.shape[0] != n:
                raise IndexError("boolean index of the wrong length")
            return np.nonzero(k)[0]
        if k.dtype.kind in "OUS":
            idx = names.get_indexer(k)
            if (idx < 0).any():
                raise KeyError(f"unknown names: {list(k[idx < 0])[:5]}")
            return idx
        return np.atleast_1d(k).astype(int)

    def __getitem__(self, key):
        """``dds[:, genes]`` / ``dds[samples]`` / ``dds[samples, genes]`` (names, positions, masks, slices): a new data set
        on the sub-matrix with the fields sliced along (dds.py:868-873, 1330; AnnData semantics).  Layers that still live
        on the device are fetched through the parent when read."""
        if not isinstance(key, tuple):
            key = (key, slice(None))
        rows = self._index(key[0], self.obs_names, self.n_obs)
        cols = self._index(key[1], self.var_names, self.n_vars)
        new = object.__new__(type(self))
        new.__dict__.update({k: v for k, v in self.__dict__.items()
                             if k not in ("obs", "obsm", "var", "varm", "layers", "uns", "X", "_pipe_obj", "_step", "_res")})
        new.X = self.X[np.ix_(rows, cols)]
        new.obs, new.var = self.obs.iloc[rows].copy(), self.var.iloc[cols].copy()
        new.obs_names, new.var_names = self.obs_names[rows], self.var_names[cols]
        new.uns = dict(self.uns)
        new.varm = {k: (v.iloc[cols].copy() if hasattr(v, "iloc") else np.asarray(v)[cols]) for k, v in self.varm.items()}
        new.obsm = _LazyObsm(new, {"design_matrix": self.obsm["design_matrix"].iloc[rows].copy()})
        for k in self.obsm.available():
            if k != "design_matrix":  # N x (genes with counts): sliced by sample only
                new.obsm.bind_parent(k, self, rows, None)
        new.layers = _LazyLayers(new)
        for k in self.layers.available():
            new.layers.bind_parent(k, self, rows, cols)
        new._control_genes = None if self._control_genes is None else np.nonzero(
            np.isin(cols, np.arange(self.n_vars)[self._control_genes]))[0]
        new._pipe_obj, new._step, new._res, new._sf_published = None, None, None, None
        return new

    def copy(self):
        """A data set of its own: host fields copied (layers that live on the device are fetched first, unless
        ``low_memory``), device state not shared - the copy rebuilds it when a stage is asked for."""
        new = object.__new__(type(self))
        new.__dict__.update(self._host_state())
        new._ctx = self._ctx
        return new

    def _host_state(self):
        d = {k: v for k, v in self.__dict__.items() if k not in ("_pipe_obj", "_step", "_ctx", "layers", "obsm", "inference")}
        d = _copy.deepcopy(d)
        d["layers"] = self.layers.materialised_copy(skip=self.low_memory)
        d["obsm"] = self.obsm.materialised_copy(skip=self.low_memory)
        d["inference"] = None
        d["_pipe_obj"], d["_step"], d["_ctx"] = None, None, None
        return d

    def __getstate__(self):
        """Pickling (examples/plot_step_by_step.py:175-177; the reference needs to_picklable_anndata, dds.py:1112-1138):
        the host fields travel, the device context and the open pass do not."""
        return self._host_state()

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.layers._dds = self
        self.obsm._dds = self

    def to_picklable_anndata(self):
        """The reference's name for "something that pickles" (dds.py:1112-1138): here the data set itself does."""
        return self.copy()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def close(self):
        """Release the device state (the host fields stay)."""
        p = self.__dict__.get("_pipe_obj")
        if p is not None:
            p.close()
        self._pipe_obj, self._step = None, None

    # ------------------------------------------------------------------ the device pipeline behind the fields
    @property
    def _pipe(self) -> DeseqPipeline:
        if self._pipe_obj is None:
            self._pipe_obj = DeseqPipeline(
                self.X, self.obsm["design_matrix"].to_numpy(), ctx=self._ctx, device=self._device, min_mu=self.min_mu,
                min_disp=self.min_disp, max_disp=self.max_disp, refit_cooks=self.refit_cooks,
                min_replicates=self.min_replicates, beta_tol=self.beta_tol, fit_type=self.fit_type,
                size_factors_fit_type=self.size_factors_fit_type, control_genes=self._control_genes,
                keep_cooks=not self.low_memory)
            self._ctx = self._pipe_obj.ctx
            self._step = None
        p = self._pipe_obj
        p.fit_type, p.refit_cooks = self.fit_type, bool(self.refit_cooks)  # (attributes a user may change between calls)
        return p

    def _given_size_factors(self):
        """Size factors a pass must START from: those in ``obs`` when the user put them there (dds.py:724-726 only fits
        them when the column is missing), None when they are the ones the engine published itself."""
        if "size_factors" not in self.obs:
            return None
        sf = np.asarray(self.obs["size_factors"], dtype=float)
        return sf

    def _advance(self, upto, refit_size_factors=False):
        """Bring the open pass to stage ``upto`` (opening a new one when there is none, when another use of the pipeline
        has recycled its buffers, or when the size factors are to be fitted again), stage by stage."""
        pipe = self._pipe
        st = self._step
        order = pipe.STAGES
        if (st is None or refit_size_factors or not pipe.step_alive(st) or st.done == "finish"
                or order.index(st.done) > order.index(upto)):
            given = None if refit_size_factors else self._given_size_factors()
            st = self._step = pipe.begin_step(size_factors=given)
        if upto in ("lfc", "refit", "finish") and order.index(st.done) < order.index("lfc") and "dispersions" in self.var:
            pipe.advance(st, "map")
            mine = np.asarray(self.var["dispersions"], dtype=float)
            pub = getattr(self, "_disp_published", None)
            if pub is None or not np.array_equal(mine, pub, equal_nan=True):  # edited between the stages: honour it
                pipe.set_dispersions(st, mine)
        pipe.advance(st, upto)
        return st

    # ------------------------------------------------------------------ the pipeline
    def deseq2(self, fit_type=None):
        """Size factors, dispersions, LFCs, Cook's outliers and their refit (dds.py:516-562)."""
        if fit_type is not None:
            self.fit_type = fit_type
        pipe = self._pipe
        self._step = None
        r = pipe.deseq2()
        self._publish_all(r)
        return self

    def _publish_all(self, r):
        # (the pipeline's result vectors are views of ITS page-locked slabs: the data set keeps copies of its own)
        self._res = type(r)(**{k: (np.array(v) if isinstance(v, np.ndarray) else v)
                               for k, v in ((f, getattr(r, f)) for f in r.__dataclass_fields__)})
        r = self._res
        v, cols = self.var, self.obsm["design_matrix"].columns
        self._set_size_factors(r.size_factors)
        v["non_zero"], v["_normed_means"] = np.array(r.non_zero), np.array(r.normed_means)
        self.non_zero_idx = np.arange(self.n_vars)[np.asarray(r.non_zero, bool)]
        v["_MoM_dispersions"], v["genewise_dispersions"] = np.array(r.mom_dispersions), np.array(r.genewise_dispersions)
        v["_genewise_converged"], v["fitted_dispersions"] = np.array(r.genewise_converged), np.array(r.fitted_dispersions)
        v["MAP_dispersions"], v["_MAP_converged"] = np.array(r.MAP_dispersions), np.array(r.MAP_converged)
        v["dispersions"], v["_outlier_genes"] = np.array(r.dispersions), np.array(r.outlier_genes)
        self._disp_published = np.array(r.dispersions)
        v["_LFC_converged"] = np.array(r.LFC_converged)
        if self._pipe_obj is None or self._pipe_obj.refit_cooks or getattr(self._step, "force_refit", False):
            v["replaced"], v["refitted"] = np.array(r.replaced), np.array(r.refitted)  # dds.py:1317-1326: written by refit()
        v["_pvalue_cooks_outlier"] = np.array(r.cooks_outlier)
        self.new_all_zeroes_genes = self.var_names[np.asarray(r.new_all_zeroes, bool)]
        self.varm["LFC"] = pd.DataFrame(np.array(r.LFC), index=self.var_names, columns=cols)
        self.uns["disp_function_type"] = r.disp_function_type
        if r.trend_coeffs is not None:
            self.uns["trend_coeffs"] = pd.Series(np.array(r.trend_coeffs), index=["a0", "a1"])
        if r.mean_disp is not None:
            self.uns["mean_disp"] = r.mean_disp
        self.uns["_squared_logres"], self.uns["prior_disp_var"] = r.squared_logres, r.prior_disp_var
        lay, obm = self.layers, self.obsm
        lay.offer("normed_counts", "_mu_LFC", "_hat_diagonals")
        obm.offer("_mu_LFC", "_hat_diagonals")
        if not self.low_memory:  # dds.py:1103-1106: the Cook's layers do not survive cooks_outlier() in low-memory mode
            lay.offer("cooks", "_mu_hat")
            if "refitted" in v and np.asarray(r.refitted, bool).any():
                lay.offer("replace_cooks")
        else:  # dds.py:1032-1034
            obm.withdraw("_mu_LFC", "_hat_diagonals")
            lay.withdraw("_mu_LFC", "_hat_diagonals", "_mu_hat")

    def _set_size_factors(self, sf):
        self.obs["size_factors"] = np.array(sf, dtype=float)
        self._sf_published = np.array(sf, dtype=float)

    def fit_size_factors(self, fit_type=None, control_genes=None):
        """Size factors only (dds.py:584-711): ``"ratio"`` (median of ratios), ``"poscounts"`` or ``"iterative"``;
        ``control_genes`` given here override the data set's (dds.py:616-619).  Writes ``obs["size_factors"]``,
        ``var["_normed_means"]`` and offers ``layers["normed_counts"]``."""
        if fit_type is None:
            fit_type = self.size_factors_fit_type
        if fit_type not in ("ratio", "poscounts", "iterative"):
            raise ValueError("fit_type: 'ratio', 'poscounts' or 'iterative'")
        pipe = self._pipe
        if control_genes is not None:
            cg = self._control_index(control_genes)
            m = np.zeros(self.n_vars, dtype=np.uint8)
            m[np.asarray(cg)] = 1
            pipe._control_mask = m
        old, pipe.size_factors_fit_type = pipe.size_factors_fit_type, fit_type
        try:
            st = self._advance("size_factors", refit_size_factors=True)
        finally:
            pipe.size_factors_fit_type = old
        self._set_size_factors(st.r.size_factors)
        sf = self._sf_published
        self.var["_normed_means"] = ((1.0 / sf) @ self.X) / self.n_obs  # (= normed_counts.mean(0), dds.py:708)
        self.layers.offer("normed_counts")
        return self

    def _fit_iterate_size_factors(self, niter: int = 10, quant: float = 0.95):
        """The ``iterative`` mode by its reference name (dds.py:1460-1548; tests/test_pydeseq2.py:344-364)."""
        if (niter, quant) != (10, 0.95):
            raise NotImplementedError("iterative size factors: niter = 10, quant = 0.95 (the reference's defaults)")
        return self.fit_size_factors("iterative")

    def _need(self, where, key, stage_fn):
        """The reference's lazy chaining (dds.py:725, 812, 849, 892, 944, 992): a stage whose input is not there yet runs
        the stage that writes it."""
        if key not in where:
            stage_fn()

    def fit_genewise_dispersions(self, vst=False):
        """Genewise dispersions (dds.py:713-797): MoM start, mu_hat (linear model or IRLS), L-BFGS-B per gene."""
        self._need(self.obs, "size_factors", lambda: self.fit_size_factors(fit_type=self.size_factors_fit_type))
        st = self._advance("genewise")
        r = self._pipe.publish(st, ("nm", "mom", "gw", "gconv"))
        if self._sf_published is None or not np.array_equal(self._sf_published, np.asarray(self.obs["size_factors"], float)):
            self._sf_published = np.asarray(self.obs["size_factors"], dtype=float).copy()
        v = self.var
        v["non_zero"] = np.array(r.non_zero)
        self.non_zero_idx = np.arange(self.n_vars)[np.asarray(r.non_zero, bool)]
        v["_normed_means"], v["_MoM_dispersions"] = r.normed_means, r.mom_dispersions
        v["vst_genewise_dispersions" if vst else "genewise_dispersions"] = r.genewise_dispersions
        v["_genewise_converged"] = r.genewise_converged
        self.layers.offer("normed_counts", "_vst_mu_hat" if vst else "_mu_hat")
        return self

    def fit_dispersion_trend(self, vst=False):
        """Dispersion trend (dds.py:799-838): parametric gamma GLM with the mean-based fallback, or ``"mean"``."""
        name = "vst_genewise_dispersions" if vst else "genewise_dispersions"
        self._need(self.var, name, lambda: self.fit_genewise_dispersions(vst))
        if vst:
            old, self.fit_type = self.fit_type, getattr(self, "vst_fit_type", self.fit_type)
        try:
            st = self._advance("trend")
        finally:
            if vst:
                self.fit_type = old
        r = self._pipe.publish(st, ("fit",))
        if r.disp_function_type == "parametric":
            self.uns["vst_trend_coeffs" if vst else "trend_coeffs"] = pd.Series(np.array(r.trend_coeffs), index=["a0", "a1"])
        else:
            self.uns["mean_disp"] = r.mean_disp
        if vst:
            if r.disp_function_type != "parametric":
                self.vst_fit_type = "mean"
            if r.disp_function_type != "parametric":
                self.var["fitted_dispersions"] = r.fitted_dispersions
            return self
        self.uns["disp_function_type"] = r.disp_function_type
        self.var["fitted_dispersions"] = r.fitted_dispersions
        return self

    # ---- contrasts from conditions (dds.py:339-347, 564-582: thin wrappers over `formulaic_contrasts`, which is not a
    # dependency here; the same vectors from the façade's own design builder)
    @property
    def variables(self):
        """Names of the metadata columns the design formula uses (dds.py:339-347)."""
        if not isinstance(self.design, str):
            raise ValueError("Retrieving variables is only possible if the model was initialized using a formula.")
        out = []
        for t in self.design.strip().lstrip("~").replace("*", "+").replace(":", "+").split("+"):
            t = t.strip()
            if t and t not in ("0", "1", "-1") and t not in out:
                out.append(t)
        return out

    def cond(self, **kwargs):
        """The design row of a condition: ``dds.cond(condition="B", group="X")`` - variables that are not named sit at
        their reference level (categorical) or at 0 (numeric).  ``dds.cond(...) - dds.cond(...)`` is a contrast vector for
        ``DeseqStats`` (dds.py:564-578)."""
        if not isinstance(self.design, str):
            raise ValueError("cond() needs a design formula.")
        unknown = [k for k in kwargs if k not in self.variables]
        if unknown:
            raise ValueError(f"Variables {unknown} are not part of the design formula.")
        row = {}
        for v in self.variables:
            col = self.obs[v]
            categorical = col.dtype.kind in "OUSb" or str(col.dtype) == "category"
            if v in kwargs:
                val = kwargs[v]
                if categorical and str(val) not in set(col.astype(str)):
                    raise ValueError(f"'{val}' is not a level of '{v}'.")
            elif categorical:
                levels = sorted(col.astype(str).unique())
                named = [c for c in self.obsm["design_matrix"].columns if c.startswith(f"{v}[T.")]
                ref = [lv for lv in levels if f"{v}[T.{lv}]" not in named]
                val = ref[0] if ref else levels[0]
            else:
                val = 0.0
            row[v] = val
        # one more row appended to the metadata so that the factor levels (and their coding) are those of the data set
        probe = pd.concat([self.obs[self.variables], pd.DataFrame([row], index=["__cond__"])])
        for v in self.variables:
            if self.obs[v].dtype.kind not in "OUSb" and str(self.obs[v].dtype) != "category":
                probe[v] = probe[v].astype(float)
        cols = list(self.obsm["design_matrix"].columns)
        dm = build_design(probe, self.design, getattr(self, "_ref_level", None))
        dm = dm.reindex(columns=cols, fill_value=0.0)
        return dm.loc["__cond__"].to_numpy(dtype=float)

    def contrast(self, column, baseline, group_to_compare):
        """Contrast vector of a pairwise comparison within one design variable (dds.py:580-582)."""
        return self.cond(**{column: group_to_compare}) - self.cond(**{column: baseline})

    def disp_function(self, x):
        """The fitted trend at normalised means ``x`` (dds.py:833-838)."""
        x = np.asarray(x, dtype=float)
        if self.uns["disp_function_type"] == "parametric":
            c = np.asarray(self.uns["trend_coeffs"], dtype=float)
            return c[0] + c[1] / x
        return np.full_like(x, self.uns["mean_disp"])

    def fit_dispersion_prior(self):
        """Prior variance of the log dispersions around the trend (dds.py:840-884).  The device pass fits trend and prior
        in one call, so this publishes what that call found."""
        self._need(self.var, "fitted_dispersions", self.fit_dispersion_trend)
        st = self._advance("trend")
        self.uns["_squared_logres"], self.uns["prior_disp_var"] = st.r.squared_logres, st.r.prior_disp_var
        return self

    def fit_MAP_dispersions(self):
        """MAP dispersions and the final ``dispersions`` (dds.py:886-935)."""
        self._need(self.uns, "prior_disp_var", self.fit_dispersion_prior)
        st = self._advance("map")
        r = self._pipe.publish(st, ("map", "mconv", "disp", "outl", "gw"))
        v = self.var
        v["MAP_dispersions"], v["_MAP_converged"] = r.MAP_dispersions, r.MAP_converged
        v["dispersions"], v["_outlier_genes"] = r.dispersions, r.outlier_genes
        self._disp_published = np.array(r.dispersions)
        if self.low_memory:  # dds.py:933-935
            self.layers.withdraw("_mu_hat")
        return self

    def fit_LFC(self):
        """Log fold changes by IRLS (dds.py:937-984); the engine's launch also leaves the per-sample half of the Cook's
        stage and the Wald statistics of the default contrast behind."""
        self._need(self.var, "dispersions", self.fit_MAP_dispersions)
        st = self._advance("lfc")
        r = self._pipe.publish(st, ("beta", "lconv"))
        self.varm["LFC"] = pd.DataFrame(r.LFC, index=self.var_names, columns=self.obsm["design_matrix"].columns)
        self.var["_LFC_converged"] = r.LFC_converged
        self.obsm.offer("_mu_LFC", "_hat_diagonals")
        self.layers.offer("_mu_LFC", "_hat_diagonals")
        return self

    def calculate_cooks(self):
        """Cook's distances (dds.py:986-1040): ``layers["cooks"]``, computed in the LFC launch's epilogue."""
        self._need(self.var, "dispersions", self.fit_MAP_dispersions)
        self._advance("lfc")
        if "LFC" not in self.varm:
            self.fit_LFC()
        self.layers.offer("cooks")
        if self.low_memory:  # dds.py:1032-1034
            self.obsm.withdraw("_mu_LFC", "_hat_diagonals")
            self.layers.withdraw("_mu_LFC", "_hat_diagonals")
        return self

    def refit(self):
        """Replace the Cook's outliers and refit the genes that had any (dds.py:1042-1064, 1301-1458)."""
        if "cooks" not in self.layers.available():
            self.calculate_cooks()
        st = self._advance("lfc")
        st.force_refit = True
        self._finish()
        return self

    def _finish(self):
        """Run the open pass to its end and publish every field (the refit's patches, the Cook's outlier mask)."""
        st = self._advance("finish")
        self._publish_all(st.r)
        return st

    def cooks_outlier(self) -> pd.Series:
        """Genes whose p-value is to be masked for a Cook's outlier (dds.py:1066-1110)."""
        if "_pvalue_cooks_outlier" not in self.var:
            if "cooks" not in self.layers.available() and self._res is None:
                self.calculate_cooks()
            self._finish()
        out = pd.Series(np.asarray(self.var["_pvalue_cooks_outlier"], dtype=bool), index=self.var_names)
        if self.low_memory:  # dds.py:1103-1106
            self.layers.withdraw("cooks", "replace_cooks")
        return out

    def _ensure_finished(self):
        """DeseqStats needs the whole fit (LFC, dispersions, Cook's mask): a finished one is taken as it is (a slice or a
        copy carries it in its fields), an open stage-wise pass is run to its end."""
        if self._res is None or self._res.pvalue is None:
            if "LFC" not in self.varm and self._step is None:
                raise AttributeError("Please run deseq2() on the DeseqDataSet first.")
            fitted = {"non_zero", "_normed_means", "dispersions", "_pvalue_cooks_outlier"} <= set(self.var.columns)
            if fitted and "LFC" in self.varm and "size_factors" in self.obs and self._step is None:
                self._res = self._res_from_fields()  # (dds[:, genes] of a fitted data set: its fields ARE the fit)
            else:
                self._finish()
        return self._res

    def _res_from_fields(self):
        """The engine-side result record rebuilt from the data set's own fields (what DeseqStats reads)."""
        from .pipeline import DeseqResult

        v = self.var
        col = lambda k, d=np.nan: (np.asarray(v[k]) if k in v else np.full(self.n_vars, d))  # noqa: E731
        naz = np.asarray(self.var_names.isin(getattr(self, "new_all_zeroes_genes", pd.Index([]))), dtype=bool)
        return DeseqResult(
            size_factors=np.asarray(self.obs["size_factors"], dtype=float), normed_means=col("_normed_means").astype(float),
            non_zero=col("non_zero").astype(bool), mom_dispersions=col("_MoM_dispersions"),
            genewise_dispersions=col("genewise_dispersions"), genewise_converged=col("_genewise_converged"),
            trend_coeffs=(np.asarray(self.uns["trend_coeffs"], dtype=float) if "trend_coeffs" in self.uns else None),
            disp_function_type=self.uns.get("disp_function_type", "parametric"), mean_disp=self.uns.get("mean_disp"),
            fitted_dispersions=col("fitted_dispersions"), squared_logres=self.uns.get("_squared_logres"),
            prior_disp_var=self.uns.get("prior_disp_var"), MAP_dispersions=col("MAP_dispersions"),
            MAP_converged=col("_MAP_converged"), outlier_genes=col("_outlier_genes", False).astype(bool),
            dispersions=col("dispersions").astype(float), LFC=self.varm["LFC"].to_numpy(dtype=float),
            LFC_converged=col("_LFC_converged"), replaced=col("replaced", False).astype(bool),
            refitted=col("refitted", False).astype(bool), new_all_zeroes=naz,
            cooks_outlier=col("_pvalue_cooks_outlier", False).astype(bool),
            pvalue=np.full(self.n_vars, np.nan), stat=np.full(self.n_vars, np.nan), lfcSE=np.full(self.n_vars, np.nan))

    def vst(self, use_design: bool = False, fit_type=None):
        """Variance stabilising transformation into ``layers["vst_counts"]`` (dds.py:349-514): dispersions
        and trend fitted with an intercept-only design unless ``use_design``."""
        self.vst_fit_type = fit_type if fit_type is not None else self.fit_type
        self.vst_fit(use_design=use_design)
        self.layers["vst_counts"] = self.vst_transform()
        return self.layers["vst_counts"]

    def vst_fit(self, use_design: bool = False):
        """Fit the transformation (dds.py:384-438): size factors, genewise dispersions and trend; keeps the
        training log geometric means so that new samples can be transformed later."""
        if not hasattr(self, "vst_fit_type"):
            self.vst_fit_type = self.fit_type
        if self.vst_fit_type not in ("parametric", "mean"):
            raise NotImplementedError(f"Found fit_type '{self.vst_fit_type}'. Expected 'parametric' or 'mean'.")
        p0 = self._pipe
        X = self.obsm["design_matrix"].to_numpy() if use_design else np.ones((self.n_obs, 1))
        pipe = p0 if use_design else DeseqPipeline(self.X, X, ctx=p0.ctx, min_mu=p0.min_mu, min_disp=p0.min_disp,
                                                    max_disp=p0.max_disp, beta_tol=p0.beta_tol, fit_type=self.vst_fit_type,
                                                    size_factors_fit_type=p0.size_factors_fit_type,
                                                    control_genes=self._control_genes)
        # The reference tests `"size_factors" not in self.obsm` (dds.py:404) - size factors live in .obs, so the test is
        # always true: vst_fit() ALWAYS refits them with size_factors_fit_type and overwrites obs["size_factors"];
        # fit_size_factors falls back to the data set's control_genes (dds.py:628-631, set in __init__, dds.py:319),
        # so the control-gene mask stays in force.  Mirrored as is.
        old_ft, pipe.fit_type = pipe.fit_type, self.vst_fit_type
        try:
            r = pipe.deseq2(stop_after_trend=True)
        finally:
            pipe.fit_type = old_ft
            if pipe is not p0:
                pipe.close()
        if pipe is p0:
            self._step = None  # (the device layers of an earlier fit were recycled by this run: rebuilt when read)
        self._set_size_factors(r.size_factors)
        self.layers.offer("normed_counts")
        self.var["vst_genewise_dispersions"] = r.genewise_dispersions
        if r.disp_function_type == "parametric":
            self.uns["vst_trend_coeffs"] = pd.Series(r.trend_coeffs, index=["a0", "a1"])
            self._vst_params = dict(trend_coeffs=r.trend_coeffs)
        else:
            self.vst_fit_type = "mean"
            self._vst_params = dict(mean_disp=r.mean_disp)
        with np.errstate(divide="ignore"):  # preprocessing.deseq2_norm_fit on the training counts
            self.logmeans = np.log(self.X).mean(0)
        self.filtered_genes = ~np.isinf(self.logmeans)
        return self

    def vst_transform(self, counts=None) -> np.ndarray:
        """Apply the fitted transformation (dds.py:440-514) to the dataset's own counts, or to new samples
        (``counts``: samples x genes), whose size factors come from the TRAINING log geometric means."""
        if "size_factors" not in self.obs or not hasattr(self, "_vst_params"):
            raise RuntimeError("The vst_fit method should be called prior to vst_transform.")
        if counts is None:
            return self._pipe.vst_transform(self.obs["size_factors"].to_numpy(), **self._vst_params)
        return self._pipe.vst_transform_new(np.asarray(counts), self.logmeans, self.filtered_genes, **self._vst_params)

    @property
    def non_zero_genes(self):
        return self.var_names[np.asarray(self.var["non_zero"], dtype=bool)]


class _Lazy(dict):
    """N x G (layers) or N x genes-with-counts (obsm) matrices that stay on the device until they are read: a stage
    OFFERS a key, reading it fetches (and keeps) the host copy, ``withdraw`` removes it (``low_memory``, dds.py:228).
    An offered matrix whose device copy has been recycled by a later pass is rebuilt by running the (deterministic) pass
    again up to the stage that writes it - milliseconds on the device - so an offer never goes stale."""

    def __init__(self, dds, init=None):
        super().__init__(init or {})
        self._dds = dds
        self._offered = set()
        self._parents = {}

    def offer(self, *keys):
        for k in keys:
            if not dict.__contains__(self, k):
                self._offered.add(k)

    def withdraw(self, *keys):
        for k in keys:
            self._offered.discard(k)
            self._parents.pop(k, None)
            if dict.__contains__(self, k):
                dict.__delitem__(self, k)

    def bind_parent(self, key, parent, rows, cols):
        self._parents[key] = (parent, rows, cols)

    def available(self):
        return set(dict.keys(self)) | self._offered | set(self._parents)

    def __contains__(self, key):
        return key in self.available()

    def keys(self):
        return self.available()

    def __delitem__(self, key):
        if key not in self.available():
            raise KeyError(key)
        self.withdraw(key)

    def __missing__(self, key):
        if key in self._parents:
            parent, rows, cols = self._parents[key]
            full = type(self).__getitem__(getattr(parent, self._attr), key)
            val = full[rows] if cols is None else full[np.ix_(rows, cols)]
        elif key in self._offered:
            val = self._fetch(key)
            self._offered.discard(key)
        else:
            raise KeyError(key)
        dict.__setitem__(self, key, val)
        return val

    def materialised_copy(self, skip=False):
        """A detached copy for copy() / pickling: what was offered is fetched first (unless ``skip``)."""
        new = type(self)(None)
        for k in list(self.available()):
            if dict.__contains__(self, k) or not skip:
                try:
                    v = self[k]
                except Exception:  # noqa: BLE001 - the device copy is gone (another pass recycled it): not part of the state
                    continue
                dict.__setitem__(new, k, v.copy() if hasattr(v, "copy") else v)
        return new


class _LazyLayers(_Lazy):
    """``normed_counts``, ``_mu_hat``, ``_mu_LFC``, ``_hat_diagonals``, ``cooks``, ``replace_cooks``: N x G, NaN columns for
    the genes without counts."""

    _attr = "layers"

    def _fetch(self, key):
        dds = self._dds
        if key == "normed_counts":
            return dds.X / np.asarray(dds.obs["size_factors"], dtype=float)[:, None]
        if key in ("_mu_hat", "_vst_mu_hat"):
            st = dds._step
            if st is None or not dds._pipe.step_alive(st) or getattr(st, "mh", None) is None:
                # deseq2() in one go keeps no open pass: mu_hat is a function of the fit's inputs, rebuild the stage
                st = dds._advance("genewise")
            return dds._pipe.mu_hat_host(st)
        if key == "replace_cooks":  # dds.py:1449-1458: the Cook's distances with the replaceable samples of the refitted genes zeroed
            ck = np.array(self["cooks"])
            rep = np.asarray(dds._pipe.design.replaceable, dtype=bool)
            for j in np.nonzero(np.asarray(dds.var["refitted"], dtype=bool))[0]:
                ck[rep, j] = 0.0
            return ck
        name = {"_mu_LFC": "mu_LFC", "_hat_diagonals": "hat_diagonals", "cooks": "cooks"}.get(key)
        if name is None:
            raise KeyError(key)
        pipe = dds._pipe
        if name not in pipe.layers:  # recycled by a later pass (or the pipeline is new: a copy, an unpickled data set)
            dds._advance("lfc")
        return pipe.layer(name)


class _LazyObsm(_Lazy):
    """``design_matrix`` plus the reference's ``_mu_LFC`` / ``_hat_diagonals`` (dds.py:975-976): N x (genes with counts)."""

    _attr = "obsm"

    def _fetch(self, key):
        dds = self._dds
        full = dds.layers[key]
        return np.ascontiguousarray(full[:, np.asarray(dds.var["non_zero"], dtype=bool)])


class DeseqStats:
    """Wald tests, adjusted p-values and LFC shrinkage (cf. ``pydeseq2.ds.DeseqStats``, ds.py:110-447)."""

    def __init__(self, dds: DeseqDataSet, contrast, alpha=0.05, cooks_filter=True, independent_filter=True,
                 prior_LFC_var=None, lfc_null=0.0, alt_hypothesis=None, inference=None, quiet=True, n_cpus=None):
        finish = getattr(dds, "_ensure_finished", None)
        if finish is not None:
            finish()  # (a stage-wise pass left open after fit_LFC() / calculate_cooks() is run to its end)
        elif dds._res is None:
            raise AttributeError("Please run deseq2() on the DeseqDataSet first.")
        if getattr(dds, "refit_cooks", False) and "replaced" not in dds.var:  # ds.py:208-216
            raise AttributeError("dds has 'refit_cooks' set to True but Cooks outliers have not been refitted. Please run "
                                 "'dds.refit()' first or set 'dds.refit_cooks' to False.")
        self.dds, self.alpha, self.prior_LFC_var = dds, alpha, prior_LFC_var
        self.cooks_filter, self.independent_filter = cooks_filter, independent_filter
        self.lfc_null, self.alt_hypothesis, self.quiet = lfc_null, alt_hypothesis, quiet
        self.design_matrix = dds.obsm["design_matrix"]
        self.LFC = dds.varm["LFC"].copy()
        self.base_mean = dds.var["_normed_means"].copy()
        self.contrast = contrast
        self.contrast_vector = self._contrast_vector(contrast)
        self.shrunk_LFCs = False

    def _contrast_vector(self, contrast) -> np.ndarray:
        # errors as in the reference (ds.py:172-188, tests/test_edge_cases.py::test_contrast): IndexError for a
        # list that is too short, ValueError for unknown factors / levels and for a vector of the wrong length
        cols = list(self.design_matrix.columns)
        if contrast is None:
            raise ValueError('Default contrasts are no longer supported. The "contrast" argument must be provided.')
        if isinstance(contrast, np.ndarray) or not isinstance(contrast[0], str):
            v = np.asarray(contrast, dtype=float)
            if v.shape[0] != len(cols):
                raise ValueError("The contrast vector must have the same length as the design matrix.")
            return v
        factor, tested, ref = str(contrast[0]), str(contrast[1]), str(contrast[2])
        if factor not in self.dds.obs.columns:
            raise ValueError(f"The contrast variable ('{factor}') should be one of the design factors.")
        levels = sorted(self.dds.obs[factor].astype(str).unique())
        if tested not in levels or ref not in levels:
            raise ValueError(f"The contrast levels ({tested}, {ref}) should be levels of '{factor}': {levels}.")
        v = np.zeros(len(cols))
        for lv, sign in ((tested, 1.0), (ref, -1.0)):
            name = f"{factor}[T.{lv}]"
            if name in cols:  # the reference level of the factor has no column: it is the intercept
                v[cols.index(name)] = sign
        return v

    def run_wald_test(self):
        """Wald test of the CURRENT coefficients (``self.LFC``: shrunk ones after ``lfc_shrink``), ds.py:303-360."""
        r = self.dds._res
        P = self.LFC.shape[1]
        if self.prior_LFC_var is not None:  # ds.py:326-329
            ridge = np.diag(1.0 / np.asarray(self.prior_LFC_var, dtype=float) ** 2)
        else:
            ridge = np.diag(np.repeat(1e-6, P))
        pv, st, se = self.dds._pipe.wald(r, self.contrast_vector, self.lfc_null, self.alt_hypothesis,
                                         lfc=self.LFC.to_numpy(), ridge=ridge)
        idx = self.dds.var_names
        self.p_values, self.statistics, self.SE = pd.Series(pv, index=idx), pd.Series(st, index=idx), pd.Series(se, index=idx)
        self._wald_key = (self.lfc_null, self.alt_hypothesis)
        if hasattr(self, "padj"):
            del self.padj

    def summary(self, **kwargs) -> pd.DataFrame:
        """Wald test, Cook's filtering, adjusted p-values; returns and stores ``results_df`` (ds.py:219-299).
        The Wald test is (re)run only when there are no p-values yet or ``lfc_null`` / ``alt_hypothesis`` change
        (ds.py:255-264), so a summary after ``lfc_shrink`` keeps the shrunk coefficients with their standard errors."""
        self.lfc_null = kwargs.get("lfc_null", self.lfc_null)
        self.alt_hypothesis = kwargs.get("alt_hypothesis", self.alt_hypothesis)
        if not hasattr(self, "p_values") or getattr(self, "_wald_key", None) != (self.lfc_null, self.alt_hypothesis):
            self.run_wald_test()
        if not hasattr(self, "padj"):
            if self.cooks_filter:
                self._cooks_filtering()
            if self.independent_filter:
                self._independent_filtering()
            else:
                self._p_value_adjustment()
        df = pd.DataFrame(index=self.dds.var_names)
        df["baseMean"] = self.base_mean
        df["log2FoldChange"] = self.LFC.to_numpy() @ self.contrast_vector / np.log(2)
        df["lfcSE"] = self.SE / np.log(2)
        df["stat"], df["pvalue"], df["padj"] = self.statistics, self.p_values, self.padj
        self.results_df = df
        return df

    def plot_MA(self, log: bool = True, save_path=None, **kwargs):
        """MA plot (ds.py:449-484).  Plotting is outside this engine's scope: the reference's precondition is kept (an
        AttributeError before ``summary()``), the figure itself needs matplotlib and is a scatter of ``results_df``."""
        if not hasattr(self, "results_df"):
            raise AttributeError("Trying to make an MA plot but p-values were not computed yet. "
                                 "Please run the summary() method first.")
        try:
            import matplotlib.pyplot as plt
        except ImportError as e:  # pragma: no cover - matplotlib is not a dependency of the engine
            raise NotImplementedError("plot_MA needs matplotlib, which this engine does not depend on") from e
        df = self.results_df
        sig = (df["padj"] < self.alpha).to_numpy()
        fig, ax = plt.subplots()
        ax.scatter(df["baseMean"], df["log2FoldChange"], c=np.where(sig, "red", "grey"), s=kwargs.pop("s", 8), **kwargs)
        if log:
            ax.set_xscale("log")
        ax.set_xlabel("mean of normalized counts")
        ax.set_ylabel("log2 fold change")
        if save_path is not None:
            fig.savefig(save_path, bbox_inches="tight")
        return ax

    def _cooks_filtering(self):
        """p-values of the Cook's outlier genes -> NaN (ds.py:544-550)."""
        if not hasattr(self, "p_values"):
            self.run_wald_test()
        pv = self.p_values.to_numpy().copy()
        pv[self.dds.cooks_outlier().to_numpy()] = np.nan
        self.p_values = pd.Series(pv, index=self.dds.var_names)

    def _adjust(self, independent_filter):
        if not hasattr(self, "p_values"):
            self.run_wald_test()
        padj, self._padj_info = _summary.adjusted_pvalues(self.dds._pipe.ctx, self.base_mean.to_numpy(),
                                                          self.p_values.to_numpy(), self.alpha, independent_filter)
        self.padj = pd.Series(padj, index=self.dds.var_names)

    def _independent_filtering(self):
        """Adjusted p-values with the baseMean cut-off that maximises the rejections (ds.py:486-527), on the device."""
        self._adjust(True)

    def _p_value_adjustment(self):
        """Benjamini-Hochberg over all genes with a p-value (ds.py:529-542)."""
        self._adjust(False)

    def __getstate__(self):
        """Pickles with its data set (examples/plot_step_by_step.py:243-246); device buffers are not part of the state."""
        return dict(self.__dict__)

    def lfc_shrink(self, coeff: str, adapt: bool = True) -> pd.DataFrame:
        """apeGLM shrinkage of one LFC column, p-values unchanged (ds.py:363-447)."""
        if coeff not in self.LFC.columns:
            raise KeyError(f"The coeff argument '{coeff}' should be one the LFC columns. "
                           f"The available LFC coeffs are {self.LFC.columns[1:]}.")
        j = self.LFC.columns.get_loc(coeff)
        if not hasattr(self, "SE"):
            self.run_wald_test()
        r = self.dds._res
        work = type(r)(**{k: getattr(r, k) for k in r.__dataclass_fields__})
        work.LFC, work.lfcSE = self.LFC.to_numpy().copy(), self.SE.to_numpy()
        work.size_factors = np.asarray(self.dds.obs["size_factors"], dtype=float)
        work.dispersions = np.asarray(self.dds.var["dispersions"], dtype=float)
        lfc, se, conv, self.prior_scale = _summary.lfc_shrink(self.dds._pipe, work, j, adapt=adapt)
        self.LFC.iloc[:, j] = lfc
        self.SE = pd.Series(se, index=self.dds.var_names)
        self._LFC_shrink_converged = pd.Series(conv, index=self.dds.var_names)
        self.shrunk_LFCs = True
        if hasattr(self, "results_df"):
            self.results_df["log2FoldChange"] = self.LFC.iloc[:, j] / np.log(2)
            self.results_df["lfcSE"] = self.SE / np.log(2)
            return self.results_df
        return None
