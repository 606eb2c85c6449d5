"""ctypes binding of libdeseq_hip.so (C ABI in include/deseq_hip.h).

The shared library is the product: there is NO CPU fallback.  If it is missing or no
MI355X is visible, loading / context creation raises and every op fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSQ_LIB", os.path.join(_HERE, "libdeseq_hip.so"))  # DSQ_LIB: developer A/B builds

DSQ_MAX_P = 48
_DEBUG = bool(os.environ.get("DSQ_DEBUG"))
SAMPLE_MAJOR, GENE_MAJOR = 0, 1
I32, I64 = 0, 1
ALT = {None: 0, "greaterAbs": 1, "lessAbs": 2, "greater": 3, "less": 4}

c_void_p, c_int, c_double, c_size_t = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_vp = c_void_p
HOOK_FN = C.CFUNCTYPE(None, c_void_p)  # dsq_hook_fn (include/deseq_hip.h)


class DsqCells(C.Structure):
    """dsq_cells of include/deseq_hip.h: the design's distinct rows (device pointers)."""

    _fields_ = [("d_cell_of", C.c_void_p), ("d_Xc", C.c_void_p), ("d_XX", C.c_void_p), ("n_cells", C.c_int)]


class DsqError(RuntimeError):
    """A libdeseq_hip call failed (HIP error, bad argument, out of memory)."""


_lib = None


def load():
    """Load libdeseq_hip.so and declare every prototype of include/deseq_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DsqError(
            f"{LIB_PATH} not found: build it with `make -C pydeseq2_amd/csrc` "
            "(or __graft_entry__.build()).  pydeseq2_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)

    def proto(name, *argtypes, res=c_int):
        fn = getattr(lib, name)
        fn.argtypes = list(argtypes)
        fn.restype = res
        return fn

    proto("dsq_create", c_int, C.POINTER(_vp))
    proto("dsq_destroy", _vp, res=None)
    proto("dsq_last_error", _vp, res=C.c_char_p)
    proto("dsq_device_info", _vp, C.c_char_p, c_int, C.POINTER(c_int), C.POINTER(c_size_t), C.c_char_p, c_int)
    proto("dsq_sync", _vp)
    proto("dsq_debug_pending_error", res=C.c_char_p)
    proto("dsq_timer_start", _vp)
    proto("dsq_timer_stop", _vp, C.POINTER(C.c_float))
    proto("dsq_last_alpha_kernel", _vp, C.POINTER(C.c_float), C.POINTER(c_int))
    proto("dsq_malloc", _vp, c_size_t, C.POINTER(_vp))
    proto("dsq_free", _vp, _vp)
    proto("dsq_memset", _vp, _vp, c_int, c_size_t)
    proto("dsq_h2d", _vp, _vp, _vp, c_size_t)
    proto("dsq_d2h", _vp, _vp, _vp, c_size_t)
    proto("dsq_h2d_2d", _vp, _vp, c_size_t, _vp, c_size_t, c_size_t, c_size_t)
    proto("dsq_d2h_2d", _vp, _vp, c_size_t, _vp, c_size_t, c_size_t, c_size_t)
    # Inference level
    proto("dsq_inf_lin_reg_mu", _vp, _vp, c_int, c_int, _vp, _vp, c_int, c_int, c_int, c_double, _vp)
    # (the ...2 entry points carry the trailing `optimizer`; the unsuffixed ones are the pre-ABI-5 signatures: L-BFGS-B)
    proto("dsq_inf_irls2", _vp, _vp, c_int, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_double, c_double,
          c_double, c_double, c_int, _vp, _vp, _vp, _vp, c_int)
    proto("dsq_inf_irls", _vp, _vp, c_int, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_double, c_double,
          c_double, c_double, c_int, _vp, _vp, _vp, _vp)
    proto("dsq_inf_alpha_mle2", _vp, _vp, c_int, c_int, _vp, _vp, c_int, _vp, c_int, c_int, c_int, c_double,
          c_double, c_double, c_int, c_int, _vp, _vp, c_int)
    proto("dsq_inf_alpha_mle", _vp, _vp, c_int, c_int, _vp, _vp, c_int, _vp, c_int, c_int, c_int, c_double,
          c_double, c_double, c_int, c_int, _vp, _vp)
    proto("dsq_abi_version", res=c_int)
    proto("dsq_plugin_cache_config", _vp, c_int, C.c_longlong)
    proto("dsq_plugin_cache_clear", _vp)
    proto("dsq_plugin_cache_stats", _vp, C.POINTER(c_double), c_int)
    proto("dsq_comm_info", _vp, C.POINTER(c_int), C.POINTER(c_int))
    proto("dsq_mix_bind", _vp, _vp, _vp, _vp)
    proto("dsq_mix_bind2", _vp, _vp, _vp, _vp, c_int)
    proto("dsq_dev_mix_counts_to_slots", _vp, _vp, c_int, c_int, _vp, _vp, _vp)
    proto("dsq_dev_mix_mu_slots", _vp, _vp, _vp, _vp, c_int, _vp)
    proto("dsq_host_sync_count", res=C.c_ulonglong)
    proto("dsq_dev_pack2", _vp, _vp, _vp, c_int, c_int, _vp)
    proto("dsq_dev_unzip2", _vp, _vp, c_int, c_int, _vp, _vp)
    proto("dsq_plugin_digest_host", _vp, c_int, c_int, c_int, c_int, c_int, C.POINTER(C.c_ulonglong))
    proto("dsq_inf_wald_test", _vp, _vp, _vp, _vp, _vp, c_int, _vp, _vp, c_double, c_int, c_int, c_int,
          c_int, _vp, _vp, _vp)
    proto("dsq_inf_fit_rough_dispersions", _vp, _vp, c_int, _vp, c_int, c_int, c_int, _vp)
    proto("dsq_inf_fit_moments_dispersions", _vp, _vp, c_int, _vp, c_int, c_int, _vp)
    proto("dsq_inf_fit_moments_dispersions2", _vp, _vp, c_int, _vp, c_int, c_int, _vp, _vp)
    proto("dsq_inf_dispersion_trend_gamma_glm", _vp, _vp, _vp, c_int, _vp, _vp, C.POINTER(c_int))
    proto("dsq_inf_grid_fit_alpha", _vp, _vp, c_int, c_int, _vp, _vp, c_int, c_int, c_int, c_int, c_double, c_double,
          _vp)
    proto("dsq_inf_grid_fit_beta", _vp, _vp, c_int, c_int, _vp, _vp, _vp, c_int, c_int, c_double, c_int, c_double,
          c_double, _vp)
    proto("dsq_dev_trend_loss_grad", _vp, _vp, _vp, _vp, c_int, c_double, c_double, C.POINTER(c_double),
          C.POINTER(c_double))
    proto("dsq_dev_trend_fit", _vp, _vp, _vp, c_int, c_double, c_double, _vp, C.POINTER(c_double),
          C.POINTER(c_int), C.POINTER(c_int))
    proto("dsq_dev_prior_mad", _vp, _vp, _vp, c_int, c_double, c_double, _vp, C.POINTER(c_double))
    # device-resident stages
    proto("dsq_dev_counts_to_gene_major", _vp, _vp, c_int, c_int, c_int, c_int, _vp, c_int, C.POINTER(c_int))
    proto("dsq_dev_f64_to_gene_major", _vp, _vp, c_int, c_int, c_int, _vp, c_int)
    proto("dsq_dev_logmeans", _vp, _vp, c_int, c_int, c_int, _vp, _vp)
    proto("dsq_dev_size_factors", _vp, _vp, c_int, c_int, c_int, _vp, _vp, _vp, _vp)
    proto("dsq_dev_size_factors_new", _vp, _vp, c_int, c_int, c_int, _vp, _vp, _vp, _vp)
    proto("dsq_dev_mom", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_double, c_double,
          _vp, _vp, _vp, _vp)
    proto("dsq_dev_lin_mu", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_double, _vp)
    proto("dsq_dev_trend_eval", _vp, _vp, c_int, c_double, c_double, _vp)
    proto("dsq_dev_trend_prior", _vp, _vp, _vp, c_int, c_double, c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
    proto("dsq_dev_select_dispersions", _vp, _vp, _vp, _vp, c_int, c_double, c_double, c_double, _vp, _vp)
    proto("dsq_dev_scatter_rows_f64", _vp, _vp, _vp, c_int, c_int, _vp)
    proto("dsq_dev_sf_keys_compact", _vp, _vp, c_int, c_int, c_int, _vp, _vp, _vp, _vp, C.POINTER(c_int))
    proto("dsq_prior_mad_work_doubles", c_int, res=c_size_t)
    proto("dsq_size_factors_work_doubles", c_int, c_int, res=c_size_t)
    proto("dsq_dev_mom_lin_mu", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_double, c_double,
          c_double, _vp, _vp, _vp)
    proto("dsq_dev_mom_raw", _vp, _vp, c_int, _vp, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_double, c_double,
          _vp, _vp)
    proto("dsq_dev_nll_const", _vp, _vp, c_int, c_int, c_int, _vp, _vp)
    proto("dsq_dev_nll_scaled", _vp, _vp, _vp, c_int, c_int, c_int, _vp, _vp, _vp, _vp)
    proto("dsq_dev_logmeans_poscounts", _vp, _vp, c_int, c_int, c_int, _vp, _vp)
    proto("dsq_dev_vst", _vp, _vp, c_int, c_int, c_int, _vp, c_int, c_double, c_double, _vp)
    proto("dsq_inf_lfc_shrink_nbinom_glm2", _vp, _vp, c_int, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_double,
          c_double, c_int, _vp, _vp, _vp, c_int)
    proto("dsq_inf_lfc_shrink_nbinom_glm", _vp, _vp, c_int, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_double,
          c_double, c_int, _vp, _vp, _vp)
    proto("dsq_dev_lfc_shrink", _vp, _vp, c_int, _vp, _vp, c_int, c_int, c_int, c_int, _vp, c_double, c_double,
          c_int, _vp, _vp, _vp)
    proto("dsq_dev_lfc_shrink2", _vp, _vp, c_int, _vp, _vp, c_int, c_int, c_int, c_int, _vp, c_double, c_double,
          c_int, _vp, _vp, _vp, _vp)
    proto("dsq_dev_lfc_shrink3", _vp, _vp, c_int, _vp, _vp, c_int, c_int, c_int, c_int, _vp, c_double, c_double,
          c_int, _vp, _vp, _vp, _vp, c_int)
    proto("dsq_dev_padj_prepare", _vp, _vp, _vp, c_int, c_double, _vp, _vp, _vp, _vp, C.POINTER(c_int))
    proto("dsq_dev_padj_finish", _vp, _vp, _vp, _vp, c_int, c_int, c_int, _vp)
    proto("dsq_d2d", _vp, _vp, _vp, c_size_t)
    cells_p = C.POINTER(DsqCells)
    proto("dsq_dev_mom_lin_coef", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_double, c_double,
          c_double, _vp, _vp, _vp, _vp)
    proto("dsq_dev_alpha_mle2", _vp, _vp, _vp, c_int, _vp, c_int, c_int, c_int, c_int, _vp, c_double, c_double,
          c_double, c_int, c_int, _vp, _vp, _vp, _vp, c_int, cells_p, _vp, _vp, c_double)
    proto("dsq_dev_alpha_mle3", _vp, _vp, _vp, c_int, _vp, c_int, c_int, c_int, c_int, _vp, c_double, c_double,
          c_double, c_int, c_int, _vp, _vp, _vp, _vp, c_int, cells_p, _vp, _vp, c_double, _vp, c_int, _vp, c_int, _vp)
    proto("dsq_dev_alpha_mle4", _vp, _vp, _vp, c_int, _vp, c_int, c_int, c_int, c_int, _vp, c_double, c_double,
          c_double, c_int, c_int, _vp, _vp, _vp, _vp, c_int, cells_p, _vp, _vp, c_double, _vp, c_int, _vp, c_int, _vp,
          _vp, _vp)
    proto("dsq_mix_create", _vp, _vp, c_int, c_int, C.POINTER(_vp))
    proto("dsq_mix_destroy", _vp, res=None)
    proto("dsq_mix_launch_count")
    proto("dsq_mix_slots", _vp, _vp)
    proto("dsq_mix_takes_irls", _vp, c_int)
    proto("dsq_mix_info", _vp, C.POINTER(c_int), C.POINTER(c_int), C.POINTER(c_int))
    proto("dsq_alpha_rows_eligible", c_int, c_int, c_int)
    proto("dsq_alpha_needs_mu", c_int, c_int, c_int)
    proto("dsq_dev_cell_mu", _vp, _vp, cells_p, c_int, c_int, _vp)
    proto("dsq_dev_alpha_row_split", _vp, _vp, c_int, c_int, c_int, _vp)
    proto("dsq_dev_robust_disp", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_int, _vp)
    proto("dsq_dev_robust_disp2", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_int, c_int, _vp)
    proto("dsq_dev_lfc_fit", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_int, _vp, c_double,
          c_double, c_double, c_double, c_int, _vp, _vp, _vp, _vp, _vp, cells_p, _vp, _vp, c_double, _vp, _vp, _vp,
          _vp, _vp, _vp, _vp, c_double, c_int, _vp, _vp, _vp)
    proto("dsq_dev_lfc_fit2", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_int, _vp, c_double,
          c_double, c_double, c_double, c_int, _vp, _vp, _vp, _vp, _vp, cells_p, _vp, _vp, c_double, _vp, _vp, _vp,
          _vp, _vp, _vp, _vp, c_double, c_int, _vp, _vp, _vp, _vp, c_int)
    proto("dsq_dev_irls_layers", _vp, _vp, c_int, _vp, _vp, c_int, c_int, c_int, c_int, _vp, _vp, c_double, _vp, _vp)
    proto("dsq_side_begin", _vp)
    proto("dsq_side_end", _vp)
    proto("dsq_side_wait", _vp)
    proto("dsq_side_abort", _vp)
    proto("dsq_set_deferred", _vp, c_int)
    proto("dsq_lfc_fork_begin", _vp)
    proto("dsq_lfc_fork_end", _vp)
    proto("dsq_lfc_set_part", _vp, _vp, c_int, c_int)
    proto("dsq_lfc_takes_parts", c_int, c_int, cells_p, _vp, c_int)
    proto("dsq_dev_select_dispersions_part", _vp, _vp, _vp, _vp, c_int, c_double, c_double, c_double, _vp, _vp, _vp, _vp, _vp,
          c_int, c_int)
    proto("dsq_alpha_set_late_flags", _vp, _vp)
    proto("dsq_lfc_prepare", _vp, _vp, c_int, _vp, _vp, c_int)
    proto("dsq_irls_order_hint", _vp, _vp, c_int)
    proto("dsq_set_alpha_hook", _vp, _vp, _vp)  # (fn: a HOOK_FN cast to void*, or None)
    proto("dsq_upload_counts_i32", _vp, _vp, c_int, c_size_t, _vp, C.POINTER(c_int))
    proto("dsq_host_alloc", _vp, c_size_t, C.POINTER(_vp))
    proto("dsq_host_free", _vp, _vp)
    proto("dsq_d2h_async", _vp, _vp, _vp, c_size_t)
    proto("dsq_h2d_async", _vp, _vp, _vp, c_size_t)
    proto("dsq_dev_alpha_mle", _vp, _vp, _vp, c_int, _vp, c_int, c_int, c_int, c_int, _vp, c_double,
          c_double, c_double, c_int, c_int, _vp, _vp, _vp, _vp, c_int)
    proto("dsq_dev_irls", _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_int, c_int, c_int, _vp, c_double,
          c_double, c_double, c_double, c_int, _vp, _vp, _vp, _vp, _vp)
    proto("dsq_dev_cooks", _vp, _vp, c_int, _vp, _vp, _vp, _vp, _vp, c_int, c_int, c_int, _vp, c_int, c_int,
          c_int, c_double, _vp, _vp, _vp, _vp, _vp, _vp)
    proto("dsq_dev_replace_outliers", _vp, _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_double, _vp, _vp)
    proto("dsq_dev_replace_outliers2", _vp, _vp, _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_double, _vp, _vp, c_int, _vp)
    proto("dsq_dev_wald", _vp, _vp, c_int, _vp, _vp, c_int, c_int, c_int, c_int, _vp, _vp, _vp, _vp,
          c_double, c_int, _vp, _vp, _vp)
    proto("dsq_dev_gather_rows_f64", _vp, _vp, c_int, _vp, c_int, c_int, _vp)
    proto("dsq_dev_gather_rows_i32", _vp, _vp, c_int, _vp, c_int, c_int, _vp)
    proto("dsq_comm_unique_id", _vp, C.c_char_p, c_int)
    proto("dsq_comm_init", _vp, C.c_char_p, c_int, c_int)
    proto("dsq_comm_destroy", _vp)
    proto("dsq_comm_allreduce_sum", _vp, _vp, c_size_t, c_int)
    proto("dsq_comm_allgather", _vp, _vp, _vp, c_size_t)
    proto("dsq_dev_sf_keys", _vp, _vp, c_int, c_int, c_int, _vp, _vp, _vp)
    proto("dsq_dev_sf_count", _vp, _vp, c_int, c_int, _vp)
    proto("dsq_dev_sf_init", _vp, _vp, c_int, _vp, _vp)
    proto("dsq_dev_sf_hist", _vp, _vp, c_int, c_int, _vp, c_int, _vp)
    proto("dsq_dev_sf_pick", _vp, _vp, c_int, c_int, _vp, _vp)
    proto("dsq_dev_sf_finish", _vp, _vp, _vp, c_int, _vp)
    _lib = lib
    return lib


EXPORTS = [
    "dsq_create", "dsq_destroy", "dsq_last_error", "dsq_device_info", "dsq_sync", "dsq_debug_pending_error", "dsq_timer_start",
    "dsq_timer_stop", "dsq_last_alpha_kernel", "dsq_malloc", "dsq_free", "dsq_memset", "dsq_h2d", "dsq_d2h", "dsq_h2d_2d",
    "dsq_d2h_2d", "dsq_inf_lin_reg_mu", "dsq_inf_irls", "dsq_inf_alpha_mle", "dsq_inf_wald_test",
    "dsq_inf_fit_rough_dispersions", "dsq_inf_fit_moments_dispersions", "dsq_dev_trend_loss_grad", "dsq_dev_trend_fit", "dsq_dev_prior_mad",
    "dsq_dev_counts_to_gene_major", "dsq_dev_f64_to_gene_major", "dsq_dev_logmeans",
    "dsq_dev_size_factors", "dsq_dev_mom", "dsq_dev_lin_mu", "dsq_dev_alpha_mle", "dsq_dev_irls",
    "dsq_dev_cooks", "dsq_dev_replace_outliers", "dsq_dev_wald", "dsq_dev_gather_rows_f64",
    "dsq_dev_gather_rows_i32", "dsq_comm_unique_id", "dsq_comm_init", "dsq_comm_destroy",
    "dsq_comm_allreduce_sum", "dsq_comm_allgather", "dsq_dev_sf_keys", "dsq_dev_sf_count", "dsq_dev_sf_init",
    "dsq_dev_sf_hist", "dsq_dev_sf_pick", "dsq_dev_sf_finish", "dsq_dev_trend_eval", "dsq_dev_trend_prior", "dsq_dev_select_dispersions",
    "dsq_dev_scatter_rows_f64", "dsq_d2d", "dsq_dev_mom_lin_mu", "dsq_dev_sf_keys_compact", "dsq_prior_mad_work_doubles", "dsq_size_factors_work_doubles", "dsq_dev_mom_raw", "dsq_dev_nll_const", "dsq_dev_nll_scaled", "dsq_dev_logmeans_poscounts", "dsq_dev_vst", "dsq_inf_lfc_shrink_nbinom_glm", "dsq_dev_lfc_shrink", "dsq_dev_lfc_shrink2", "dsq_dev_lfc_shrink3", "dsq_dev_padj_prepare", "dsq_dev_padj_finish", "dsq_host_alloc", "dsq_host_free", "dsq_d2h_async", "dsq_h2d_async",
    "dsq_dev_size_factors_new", "dsq_dev_mom_lin_coef", "dsq_dev_alpha_mle2", "dsq_dev_robust_disp", "dsq_dev_robust_disp2", "dsq_dev_lfc_fit", "dsq_dev_lfc_fit2", "dsq_dev_irls_layers",
    "dsq_side_begin", "dsq_side_end", "dsq_side_wait", "dsq_side_abort", "dsq_set_deferred", "dsq_irls_order_hint", "dsq_set_alpha_hook", "dsq_dev_alpha_mle3", "dsq_dev_alpha_mle4", "dsq_mix_create", "dsq_mix_destroy", "dsq_mix_info", "dsq_mix_slots", "dsq_mix_takes_irls", "dsq_dev_replace_outliers2", "dsq_mix_launch_count", "dsq_alpha_rows_eligible", "dsq_alpha_needs_mu", "dsq_dev_cell_mu", "dsq_dev_alpha_row_split",
    "dsq_upload_counts_i32", "dsq_inf_dispersion_trend_gamma_glm", "dsq_inf_grid_fit_alpha", "dsq_inf_grid_fit_beta",
    "dsq_inf_irls2", "dsq_inf_alpha_mle2", "dsq_inf_lfc_shrink_nbinom_glm2", "dsq_inf_fit_moments_dispersions2",
    "dsq_abi_version", "dsq_plugin_cache_config", "dsq_plugin_cache_clear", "dsq_plugin_cache_stats",
    "dsq_plugin_digest_host", "dsq_comm_info", "dsq_host_sync_count", "dsq_dev_pack2", "dsq_dev_unzip2",
    "dsq_mix_bind", "dsq_mix_bind2", "dsq_dev_mix_counts_to_slots", "dsq_dev_mix_mu_slots",
    "dsq_lfc_fork_begin", "dsq_lfc_fork_end", "dsq_lfc_set_part", "dsq_lfc_takes_parts", "dsq_dev_select_dispersions_part", "dsq_lfc_prepare", "dsq_alpha_set_late_flags",
]


def ptr(a):
    """Raw pointer of a numpy array (None -> NULL) or pass a device pointer (int) through."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a


class Context:
    """One HIP device context + stream (dsq_ctx).  Raises DsqError if no GPU is usable."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _vp()
        rc = self.lib.dsq_create(int(device), C.byref(h))
        if rc != 0 or not h.value:
            raise DsqError(
                f"dsq_create(device={device}) failed (rc={rc}): no usable MI355X / HIP runtime. "
                "pydeseq2_amd runs on the GPU only."
            )
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            for bufs in self.__dict__.pop("_padj_buffers", {}).values():  # cached work buffers (summary.py)
                for b in bufs:
                    b.free()
            self.lib.dsq_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.h, *args)
        if _DEBUG:
            pend = self.lib.dsq_debug_pending_error().decode()
            if pend:
                print(f"[dsq debug] pending HIP error after {name}: {pend}", flush=True)
        if rc != 0:
            msg = self.lib.dsq_last_error(self.h).decode(errors="replace")
            if rc == -2:
                raise ValueError(msg)
            raise DsqError(f"{name} failed (rc={rc}): {msg}")

    # ---- info / timing
    def device_info(self):
        name, arch = C.create_string_buffer(256), C.create_string_buffer(256)
        cu, mem = c_int(), c_size_t()
        self.call("dsq_device_info", name, 256, C.byref(cu), C.byref(mem), arch, 256)
        return dict(name=name.value.decode(), arch=arch.value.decode(), cu_count=cu.value, mem_bytes=mem.value)

    def sync(self):
        self.call("dsq_sync")

    def timer_start(self):
        self.call("dsq_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self.call("dsq_timer_stop", C.byref(ms))
        return float(ms.value)

    # ---- memory
    def malloc(self, nbytes: int) -> int:
        p = _vp()
        self.call("dsq_malloc", c_size_t(int(nbytes)), C.byref(p))
        return p.value

    def free(self, dptr):
        if dptr:
            self.call("dsq_free", _vp(dptr))

    def memset(self, dptr, value, nbytes):
        self.call("dsq_memset", _vp(dptr), int(value), c_size_t(int(nbytes)))

    def h2d(self, dptr, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.call("dsq_h2d", _vp(dptr), _vp(arr.ctypes.data), c_size_t(arr.nbytes))

    def d2h(self, arr: np.ndarray, dptr):
        assert arr.flags.c_contiguous
        self.call("dsq_d2h", _vp(arr.ctypes.data), _vp(dptr), c_size_t(arr.nbytes))
        return arr

    def d2h_rows(self, dptr, rows, cols, ld, dtype=np.float64):
        """Pitched device matrix [rows][ld] -> contiguous host [rows][cols]."""
        out = np.empty((rows, cols), dtype=dtype)
        isz = out.itemsize
        self.call("dsq_d2h_2d", _vp(out.ctypes.data), c_size_t(cols * isz), _vp(dptr), c_size_t(ld * isz),
                  c_size_t(cols * isz), c_size_t(rows))
        return out

    def h2d_rows(self, dptr, arr: np.ndarray, ld):
        arr = np.ascontiguousarray(arr)
        rows, cols = arr.shape
        isz = arr.itemsize
        self.call("dsq_h2d_2d", _vp(dptr), c_size_t(ld * isz), _vp(arr.ctypes.data), c_size_t(cols * isz),
                  c_size_t(cols * isz), c_size_t(rows))


class DeviceArray:
    """Owning handle of a device allocation (1-D or pitched 2-D)."""

    def __init__(self, ctx: Context, shape, dtype, ld=None):
        self.ctx = ctx
        self.shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        self.dtype = np.dtype(dtype)
        self.ld = ld if ld is not None else (self.shape[-1] if len(self.shape) > 1 else None)
        n = self.shape[0] * (self.ld if len(self.shape) > 1 else 1)
        self.nbytes = int(n) * self.dtype.itemsize
        self.ptr = ctx.malloc(self.nbytes)

    @classmethod
    def from_host(cls, ctx, arr, ld=None):
        arr = np.ascontiguousarray(arr)
        self = cls(ctx, arr.shape, arr.dtype, ld)
        if arr.ndim == 2 and self.ld != arr.shape[1]:
            ctx.h2d_rows(self.ptr, arr, self.ld)
        else:
            ctx.h2d(self.ptr, arr)
        return self

    def to_host(self):
        if len(self.shape) == 2 and self.ld != self.shape[1]:
            return self.ctx.d2h_rows(self.ptr, self.shape[0], self.shape[1], self.ld, self.dtype)
        out = np.empty(self.shape, dtype=self.dtype)
        return self.ctx.d2h(out, self.ptr)

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PinnedPool:
    """Free list of page-locked host buffers of one pipeline (results may outlive the pipeline)."""

    def __init__(self, ctx):
        self.ctx, self.free, self.closed = ctx, [], False

    def take(self, nbytes):
        best = None
        for k, (cap, ptr) in enumerate(self.free):  # best fit; a small request must not eat the big slab
            if nbytes <= cap <= 2 * nbytes + 65536 and (best is None or cap < self.free[best][0]):
                best = k
        if best is not None:
            cap, ptr = self.free.pop(best)
            return _PinnedSlab(self, cap, ptr)
        p = _vp()
        self.ctx.call("dsq_host_alloc", c_size_t(int(nbytes)), C.byref(p))
        return _PinnedSlab(self, int(nbytes), p.value)

    def take_free(self, nbytes):
        """A pooled buffer of at least nbytes (best fit, as take), or None - never allocates."""
        best = None
        for k, (cap, ptr) in enumerate(self.free):
            if nbytes <= cap <= 2 * nbytes + 65536 and (best is None or cap < self.free[best][0]):
                best = k
        if best is None:
            return None
        cap, ptr = self.free.pop(best)
        return _PinnedSlab(self, cap, ptr)

    def release(self, cap, ptr):
        if self.closed:
            self.ctx.call("dsq_host_free", _vp(ptr))
        else:
            self.free.append((cap, ptr))

    def close(self):
        self.closed = True
        while self.free:
            _cap, ptr = self.free.pop()
            self.ctx.call("dsq_host_free", _vp(ptr))


class _PinnedSlab:
    """Page-locked host buffer; numpy views keep it alive, the last one returns it to the pool."""

    def __init__(self, pool, cap, ptr):
        self._pool, self.cap, self.ptr = pool, cap, ptr

    def view(self, offset, count, dtype):
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(self.ptr + offset)
        buf._slab = self  # numpy array -> ctypes buffer -> slab
        return np.frombuffer(buf, dtype=dtype, count=count)

    def __del__(self):
        try:
            self._pool.release(self.cap, self.ptr)
        except Exception:
            pass
