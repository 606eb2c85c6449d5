"""CPU-side checks of the drop-in boundary: libdeseq_hip.so loads and exports every symbol
include/deseq_hip.h declares; the Python mirror of the Inference interface has the reference's
method names / signatures; the product fails loudly without a GPU (no CPU fallback)."""
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "deseq_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dsq_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from pydeseq2_amd import _lib

    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names


def test_inference_interface_matches_reference():
    """Method names and parameter names of pydeseq2.inference.Inference (inference.py:13-362)."""
    from pydeseq2_amd import HipInference

    want = {
        "lin_reg_mu": ["counts", "size_factors", "design_matrix", "min_mu"],
        "irls": ["counts", "size_factors", "design_matrix", "disp", "min_mu", "beta_tol", "min_beta",
                 "max_beta", "optimizer", "maxiter"],
        "alpha_mle": ["counts", "design_matrix", "mu", "alpha_hat", "min_disp", "max_disp", "prior_disp_var",
                      "cr_reg", "prior_reg", "optimizer"],
        "wald_test": ["design_matrix", "disp", "lfc", "mu", "ridge_factor", "contrast", "lfc_null",
                      "alt_hypothesis"],
        "fit_rough_dispersions": ["normed_counts", "design_matrix"],
        "fit_moments_dispersions": ["normed_counts", "size_factors"],
        "dispersion_trend_gamma_glm": ["covariates", "targets"],
        "lfc_shrink_nbinom_glm": ["design_matrix", "counts", "size", "offset", "prior_no_shrink_scale",
                                  "prior_scale", "optimizer", "shrink_index"],
    }
    for name, params in want.items():
        sig = inspect.signature(getattr(HipInference, name))
        assert list(sig.parameters)[1:] == params, name
    assert isinstance(HipInference.n_cpus, property)
    # where the reference tree is present (the build container; not the GPU box): the table above IS the abstract class -
    # every abstract method, its parameter names in order, and the defaults the reference declares
    ref_file = "/root/reference/pydeseq2/inference.py"
    if os.path.exists(ref_file):
        import importlib.util

        spec = importlib.util.spec_from_file_location("_ref_inference_abc", ref_file)  # (numpy / pandas only)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        ref = mod.Inference
        assert sorted(ref.__abstractmethods__) == sorted(want)
        for name in ref.__abstractmethods__:
            rs, hs = inspect.signature(getattr(ref, name)), inspect.signature(getattr(HipInference, name))
            assert list(rs.parameters) == list(hs.parameters), name
            for k, rp in rs.parameters.items():
                if rp.default is not inspect.Parameter.empty:
                    assert hs.parameters[k].default == rp.default, (name, k)


def test_fails_loudly_without_gpu():
    from pydeseq2_amd import _lib

    try:
        ctx = _lib.Context(0)
    except _lib.DsqError as e:
        assert "GPU" in str(e) or "MI355X" in str(e)
    else:  # a GPU is present: the context must be a real gfx950 device
        assert "gfx" in ctx.device_info()["arch"]


def test_host_mean_trend_matches_oracle():
    import numpy as np

    from oracle import nbglm_oracle as orc
    from pydeseq2_amd import trend
    from tests.helpers import load_kat

    gw = np.clip(load_kat("p2")["gw_alpha"], 1e-8, 40)
    assert trend.mean_trend(gw, 1e-8) == orc.mean_trend(gw, 1e-8)


def test_plugin_cache_digest_is_layout_dtype_and_thread_independent():
    """The content digest the Inference-level entry points identify a host matrix by (include/deseq_hip.h,
    csrc/dsq_plugin_cache.h): the same for the C-order and the F-order copy of a matrix, for int32 and int64 counts, for
    any number of hashing threads - and different after ANY single-element change."""
    import ctypes as C

    import numpy as np

    from pydeseq2_amd import _lib

    lib = _lib.load()

    def dg(a, threads=4):
        et = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.float64): 2}[a.dtype]
        lay = 0 if a.flags.c_contiguous else 1
        assert a.flags.c_contiguous or a.flags.f_contiguous
        out = (C.c_ulonglong * 2)()
        assert lib.dsq_plugin_digest_host(a.ctypes.data, et, lay, a.shape[0], a.shape[1], threads, out) == 0
        return (int(out[0]), int(out[1]))

    rng = np.random.default_rng(5)
    y = rng.poisson(30, (700, 900)).astype(np.int64)  # (above the single-thread threshold)
    base = dg(y)
    assert dg(np.asfortranarray(y)) == base
    assert dg(y.astype(np.int32)) == base and dg(np.asfortranarray(y.astype(np.int32))) == base
    assert dg(y, 1) == base and dg(y, 7) == base and dg(np.asfortranarray(y), 13) == base
    for (n, g) in ((0, 0), (699, 899), (123, 456)):
        z = y.copy()
        z[n, g] += 1
        assert dg(z) != base
    z = y.copy()
    z[3, 4], z[4, 3] = y[4, 3], y[3, 4]  # the same multiset of values at other positions
    assert (z == y).all() or dg(z) != base
    assert dg(y.T.copy()) != base  # (another shape)
    m = rng.normal(size=(300, 1100)) + 5.0
    bm = dg(m)
    assert dg(np.asfortranarray(m)) == bm and dg(m, 1) == bm
    m2 = m.copy()
    m2[17, 1000] = np.nextafter(m2[17, 1000], np.inf)  # one ulp
    assert dg(m2) != bm


def test_every_environment_knob_is_registered():
    """tools/knobs.py holds the registry of the DSQ_* switches (kind, default, effect): a knob read anywhere in the package
    that is missing there - or one registered that nothing reads - fails here, and so does a stale table in README.md."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "knobs.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
