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

import re
import warnings

import numpy as np
import pandas as pd

from . import summary as _summary
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
    """Counts + metadata + design, fitted on the GPU (cf. ``pydeseq2.dds.DeseqDataSet``, dds.py:206-340)."""

    def __init__(self, *, counts: pd.DataFrame, metadata: pd.DataFrame, design="~condition", design_factors=None,
                 continuous_factors=None, ref_level=None, refit_cooks=True,
                 min_mu=0.5, min_disp=1e-8, max_disp=10.0, beta_tol=1e-8, min_replicates=7, fit_type="parametric",
                 size_factors_fit_type="ratio", control_genes=None, device=0, ctx: Context | None = None, quiet=True,
                 n_cpus=None, inference=None, low_memory=False):
        # n_cpus / inference / low_memory: accepted for signature compatibility (dds.py:206-229); the engine
        # is the GPU pipeline.  design_factors (+ continuous_factors): the reference's older way to spell
        # an additive design.
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
        self.n_obs, self.n_vars = self.X.shape
        self.design = design
        dm = build_design(self.obs, design, ref_level)
        self.obsm = {"design_matrix": dm}
        if np.linalg.matrix_rank(dm.to_numpy()) < dm.shape[1]:  # dds.py:1550-1563
            warnings.warn("The design matrix is not full rank, so the model cannot be fitted, but some operations "
                          "like design-free VST remain possible. To perform differential expression analysis, "
                          "please remove the design variables that are linear combinations of others.",
                          UserWarning, stacklevel=2)
        self.var = pd.DataFrame(index=self.var_names)
        self.varm, self.layers, self.uns = {}, _LazyLayers(self), {}
        self.refit_cooks, self.fit_type, self.quiet = refit_cooks, fit_type, quiet
        if control_genes is not None:  # names, integer positions or a boolean mask (dds.py:640-650)
            cg = np.asarray(control_genes)
            control_genes = self.var_names.get_indexer(cg) if cg.dtype.kind in "OUS" else cg
            if cg.dtype.kind in "OUS" and (np.asarray(control_genes) < 0).any():
                raise KeyError("control_genes: unknown gene name")
        self._control_genes = control_genes
        self._pipe = DeseqPipeline(self.X, dm.to_numpy(), ctx=ctx, device=device, min_mu=min_mu, min_disp=min_disp,
                                   max_disp=max_disp, refit_cooks=refit_cooks, min_replicates=min_replicates,
                                   beta_tol=beta_tol, fit_type=fit_type, size_factors_fit_type=size_factors_fit_type,
                                   control_genes=control_genes)
        self._res = None

    # ------------------------------------------------------------------ the pipeline
    def deseq2(self):
        """Size factors, dispersions, LFCs, Cook's outliers and their refit (dds.py:516-562)."""
        r = self._res = self._pipe.deseq2()
        v, cols = self.var, self.obsm["design_matrix"].columns
        self.obs["size_factors"] = r.size_factors
        v["non_zero"], v["_normed_means"] = r.non_zero, r.normed_means
        v["_MoM_dispersions"], v["genewise_dispersions"] = r.mom_dispersions, r.genewise_dispersions
        v["_genewise_converged"], v["fitted_dispersions"] = r.genewise_converged, r.fitted_dispersions
        v["MAP_dispersions"], v["_MAP_converged"] = r.MAP_dispersions, r.MAP_converged
        v["dispersions"], v["_outlier_genes"] = r.dispersions, r.outlier_genes
        v["_LFC_converged"], v["replaced"], v["refitted"] = r.LFC_converged, r.replaced, r.refitted
        v["_pvalue_cooks_outlier"] = r.cooks_outlier
        self.varm["LFC"] = pd.DataFrame(r.LFC, index=self.var_names, columns=cols)
        self.uns["disp_function_type"] = r.disp_function_type
        if r.trend_coeffs is not None:
            self.uns["trend_coeffs"] = pd.Series(r.trend_coeffs, index=["a0", "a1"])
        if r.mean_disp is not None:
            self.uns["mean_disp"] = r.mean_disp
        self.uns["_squared_logres"], self.uns["prior_disp_var"] = r.squared_logres, r.prior_disp_var
        return self

    def fit_size_factors(self, fit_type=None):
        """Size factors only (dds.py:600-708): ``"ratio"`` (median of ratios) or ``"poscounts"``."""
        if fit_type is not None:
            if fit_type not in ("ratio", "poscounts", "iterative"):
                raise ValueError("fit_type: 'ratio', 'poscounts' or 'iterative'")
            self._pipe.size_factors_fit_type = fit_type
        r = self._pipe.deseq2(stop_after_size_factors=True)
        self.obs["size_factors"] = r.size_factors
        return self

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
            self.layers.clear()  # the device layers of an earlier deseq2() were recycled by this run
        self.obs["size_factors"] = r.size_factors
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

    def cooks_outlier(self) -> pd.Series:
        return pd.Series(np.asarray(self._res.cooks_outlier, dtype=bool), index=self.var_names)

    @property
    def non_zero_genes(self):
        return self.var_names[np.asarray(self.var["non_zero"], dtype=bool)]


class _LazyLayers(dict):
    """N x G layers stay on the device until asked for (``normed_counts``, ``_mu_LFC``, ``_hat_diagonals``,
    ``cooks``)."""

    def __init__(self, dds):
        super().__init__()
        self._dds = dds

    def __missing__(self, key):
        dds = self._dds
        if dds._res is None:
            raise KeyError(key)
        if key == "normed_counts":
            val = dds.X / np.asarray(dds.obs["size_factors"])[:, None]
        else:
            name = {"_mu_LFC": "mu_LFC", "_hat_diagonals": "hat_diagonals", "cooks": "cooks"}.get(key)
            if name is None:
                raise KeyError(key)
            val = dds._pipe.layer(name)
        self[key] = val
        return val


class DeseqStats:
    """Wald tests, adjusted p-values and LFC shrinkage (cf. ``pydeseq2.ds.DeseqStats``, ds.py:110-447)."""

    def __init__(self, dds: DeseqDataSet, contrast, alpha=0.05, cooks_filter=True, independent_filter=True,
                 prior_LFC_var=None, lfc_null=0.0, alt_hypothesis=None, inference=None, quiet=True, n_cpus=None):
        if dds._res is None:
            raise AttributeError("Please run deseq2() on the DeseqDataSet first.")
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
            pv = self.p_values.to_numpy().copy()
            if self.cooks_filter:
                pv[self.dds.cooks_outlier().to_numpy()] = np.nan
            self.p_values = pd.Series(pv, index=self.dds.var_names)
            padj, self._padj_info = _summary.adjusted_pvalues(self.dds._pipe.ctx, self.base_mean.to_numpy(), pv,
                                                              self.alpha, self.independent_filter)
            self.padj = pd.Series(padj, index=self.dds.var_names)
        df = pd.DataFrame(index=self.dds.var_names)
        df["baseMean"] = self.base_mean
        df["log2FoldChange"] = self.LFC.to_numpy() @ self.contrast_vector / np.log(2)
        df["lfcSE"] = self.SE / np.log(2)
        df["stat"], df["pvalue"], df["padj"] = self.statistics, self.p_values, self.padj
        self.results_df = df
        return df

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
