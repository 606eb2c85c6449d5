"""Time the UNMODIFIED reference kernels on a slice of the benchmark configurations (build container only).

The GPU box has no /root/reference, so bench.py's in-run CPU baseline is the oracle ("port").  This script is the
honest substitute SURVEY 8(d) asks for: the reference's own ``DefaultInference`` (default_inference.py:14-264 - one
joblib task per gene calling utils.fit_alpha_mle / irls_solver / wald_test ...) is imported through the same 3-line
shim as tests/golden/make_golden.py and driven, in dds.py / ds.py call order, by the oracle's orchestration
(``orc.deseq2(..., inference=DefaultInference())``; pydeseq2.dds itself needs anndata, which is not installed).  The
oracle's own kernels are timed on the same slice and the same cores, which gives the factor by which bench.py's
"port" baseline differs from the reference's kernels.

    python tests/tests/tools/cpu_reference_slice.py [genes]     ->  profiles/cpu_reference_slice.json
"""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from bench import CONFIGS, synth_fast  # noqa: E402
from make_golden import _import_reference  # noqa: E402
from oracle import nbglm_oracle as orc  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    genes = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    _, _, _, di = _import_reference()
    cores = os.cpu_count() or 1
    out = {"_note": "unmodified reference kernels (pydeseq2 v0.5.3 DefaultInference, joblib loky) under the oracle's "
                    "restatement of the dds.py / ds.py orchestration, timed in the build container; 'port' = the "
                    "oracle's own kernels on the same slice and cores (what bench.py times on the GPU box)",
           "_host": {"cpu": cpu_model(), "cores": cores}}
    out["_round"] = 5
    for cfg in ("c2", "c3", "c4", "c5"):
        _, N, design = CONFIGS[cfg][:3]
        g_cfg = genes if cfg != "c5" else max(genes // 4, 200)  # (5000 samples: a quarter of the genes, the same seconds)
        if cfg == "c5":
            from pydeseq2_amd.synth import synth_counts_block

            counts, X = synth_counts_block(g_cfg, N, design, 0)
        else:
            counts, X = synth_fast(g_cfg, N, design, seed=0)
        inf = di.DefaultInference(n_cpus=cores)
        small = counts[:, :64]
        orc.deseq2(small, X, inference=inf, keep_layers=False)   # loky workers up, imports done
        orc.deseq2(small, X, n_jobs=cores, keep_layers=False)
        t0 = time.perf_counter()
        ref = orc.deseq2(counts, X, inference=inf, keep_layers=False)
        t_ref = time.perf_counter() - t0
        t0 = time.perf_counter()
        port = orc.deseq2(counts, X, n_jobs=cores, keep_layers=False)
        t_port = time.perf_counter() - t0
        ok = ~np.isnan(ref.dispersions)
        agree = float(np.max(np.abs(port.dispersions[ok] - ref.dispersions[ok]) / ref.dispersions[ok]))
        genes_cfg = counts.shape[1]
        out[cfg] = {
            "kind": "reference", "value": round(genes_cfg / t_ref, 1), "unit": "genes/s", "cores": cores,
            "seconds": round(t_ref, 2),
            "sample": f"{genes_cfg} genes x {N} samples, design {design} (p={X.shape[1]}), same generator as bench.py",
            "port_value_same_slice_same_cores": round(genes_cfg / t_port, 1),
            "outputs_bit_identical_port_vs_reference": bool(all(
                np.array_equal(getattr(port, k), getattr(ref, k), equal_nan=True)
                for k in ("dispersions", "LFC", "pvalue", "stat", "lfcSE", "genewise_dispersions", "MAP_dispersions"))),
            "port_over_reference": round(t_ref / t_port, 2),
            "max_rel_dispersion_difference_port_vs_reference": agree,
        }
        print(cfg, out[cfg], flush=True)
    path = os.path.join(ROOT, "profiles", "cpu_reference_slice.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
