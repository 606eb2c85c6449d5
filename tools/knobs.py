#!/usr/bin/env python3
"""Registry of the DSQ_* environment knobs.

    python tools/knobs.py            -> rewrites the table between the markers in README.md
    python tools/knobs.py --check    -> exit 1 when a knob read in the source is not registered here (or the reverse),
                                        or when README.md's table is stale (tests/test_capi_exports.py runs this)

Every knob is a developer / measurement switch: the product's behaviour is the default.  kind: "A/B" = selects an older or
alternative kernel route for same-box comparisons (results stay within the documented tolerances), "tuning" = a numeric
threshold, "debug", "bench" = read by bench.py only, "plumbing".
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = {
    # name: (kind, default, what it does)
    "DSQ_LIB": ("plumbing", "pydeseq2_amd/libdeseq_hip.so", "path of the shared library to load (A/B builds from `make variant`)"),
    "DSQ_DEBUG": ("debug", "off", "print every C-ABI call that fails with its arguments"),
    "DSQ_DEBUG_ROWS": ("debug", "off", "print launch geometry and occupancy of the row / mixed-design kernels"),
    "DSQ_NO_CELL_PATH": ("A/B", "off", "ignore the design-cell structure: every design on the general kernels"),
    "DSQ_NO_ALPHA_ROWS": ("A/B", "off", "dispersion fits of <= 4-cell designs on the one-gene-per-wavefront kernel instead of k_alpha_rows"),
    "DSQ_NO_ALPHA_ROWSC": ("A/B", "off", "dispersion fits of 5..32-cell designs off k_alpha_rows_c"),
    "DSQ_NO_ALPHA_WG": ("A/B", "off", "continuation of parked dispersion fits on a wavefront per gene instead of k_alpha_wg"),
    "DSQ_WG_512": ("A/B", "off", "k_alpha_wg with 512 threads also for rows of <= 1024 samples"),
    "DSQ_NO_ALPHA_MIX": ("A/B", "off", "mixed (categorical + continuous) designs on the general kernels instead of k_alpha_mix / k_irls_mix"),
    "DSQ_NO_IRLS_MIX": ("A/B", "off", "only the IRLS of mixed designs on the general kernel"),
    "DSQ_MIX_FORCE": ("A/B", "off", "tests: accept mixed designs whose slot padding exceeds the waste limit"),
    "DSQ_MIX_WG_CONT": ("A/B", "1", "0: continuation of parked mixed-design fits on a wavefront per gene instead of a workgroup per gene"),
    "DSQ_NO_TWO_PHASE": ("A/B", "off", "dispersion fits in one launch (no parking of long line searches)"),
    "DSQ_ALPHA_EVAL_CAP": ("tuning", "0 (kernel default)", "evaluations a dispersion fit runs in the main launch before it is parked"),
    "DSQ_NO_ROW_WAVE": ("A/B", "off", "IRLS of many-column designs off the sixteen-lane row kernel"),
    "DSQ_ROW_MIN_P": ("tuning", "5", "narrowest design the sixteen-lane IRLS kernel takes"),
    "DSQ_IRLS_ROW_MIN_G": ("tuning", "kernel default", "fewest genes for which the sixteen-lane IRLS kernel is launched"),
    "DSQ_IRLS_NO_STAGE": ("A/B", "off", "IRLS without the LDS staging of the gene's row"),
    "DSQ_NO_IRLS_ORDER": ("A/B", "off", "LFC fit in gene order instead of by the iteration counts of the mu_hat fit"),
    "DSQ_NO_ROBUST_LEAN": ("A/B", "off", "robust dispersions always through the LDS-buffered kernel"),
    "DSQ_ROBUST_LEAN_MIN": ("tuning", "kernel default", "smallest cell size for which the lean robust-dispersion kernel is used"),
    "DSQ_NO_SEG_CELLS": ("A/B", "off", "robust dispersions: one design cell per sorting pass"),
    "DSQ_WIDE_MIN_P": ("tuning", "13", "narrowest design the LDS / matrix-core kernel family (dsq_k_wide.hip) takes"),
    "DSQ_WIDE_CELLS": ("A/B", "off", "wide kernels with the per-cell sums"),
    "DSQ_TREND_GRID": ("A/B", "0", "1 / 2: force the multi-workgroup / single-workgroup trend-fit kernel"),
    "DSQ_CU_SPLIT": ("tuning", "32", "compute units reserved for the latency-bound kernels while the side stream runs"),
    "DSQ_CU_SPLIT_MODE": ("tuning", "0", "which units: 0 the first ones of the mask, 1 every (CUs / split)-th"),
    "DSQ_NO_OVERLAP": ("A/B", "off", "no side stream: robust dispersions and the result copy in line"),
    "DSQ_ROBUST_EARLY": ("A/B", "off", "fork the robust dispersions before the genewise fit"),
    "DSQ_ROBUST_LATE": ("A/B", "off", "fork the robust dispersions after the genewise stage"),
    "DSQ_ROBUST_SPLIT": ("tuning", "0.78", "share of the genes whose robust dispersions run under the genewise stage's tail"),
    "DSQ_LFC_OVERLAP": ("A/B", "per design family", "1 / 0: the LFC fit in two launches, the first under the MAP stage's tail / one launch after the stage"),
    "DSQ_LFC_FLAT_PRIORITY": ("A/B", "off", "the forked LFC launch's stream at normal instead of lowest priority"),
    "DSQ_LFC_OVERLAP_MAX_WORK": ("tuning", "1.5e8", "mixed designs: most counts (genes x samples) per device for which the LFC fit is split"),
    "DSQ_LFC_OVERLAP_MIN_GENES": ("tuning", "2048", "fewest genes for which the LFC fit is split into two launches"),
    "DSQ_REPLACE_LEAN": ("A/B", "off", "outlier replacement through the buffer-less kernel (rows beyond a wavefront's LDS) whatever the row length"),
    "DSQ_MAP_WAIT": ("A/B", "per design family", "1 / 0: the MAP launch waits / does not wait for the side stream"),
    "DSQ_UPLOAD_THREADS": ("tuning", "half the cores, <= 16", "host threads narrowing the int64 counts during the upload"),
    "DSQ_UPLOAD_NO_U16": ("A/B", "off", "upload int32 chunks even where the counts fit 16 bits"),
    "DSQ_PLUGIN_CACHE": ("A/B", "1", "0: the plug-in entry points upload every argument on every call"),
    "DSQ_PLUGIN_CACHE_MB": ("tuning", "25 % of HBM", "budget of the plug-in path's device cache"),
    "DSQ_HOST_ALLOC_MALLOC": ("A/B", "off", "page-locked host buffers from hipHostMalloc instead of touched-and-registered ordinary memory"),
    "DSQ_PLUGIN_D2H_THREADS": ("tuning", "half the cores, <= 16", "host threads copying the plug-in's N x G outputs out of the staging buffers into pageable memory; 0: the runtime's own pageable copy"),
    "DSQ_PLUGIN_CACHE_VERIFY": ("debug", "off", "on a cache hit re-upload the argument and compare it with the cached device copy byte for byte"),
    "DSQ_HASH_THREADS": ("tuning", "cores / 2, 32 from 64 cores", "host threads of the plug-in path's content digest"),
    "DSQ_BENCH_SHARE_GPU": ("bench", "off", "all ranks on device 0 (multi-process path on a one-GPU box, host-staged transport)"),
    "DSQ_BENCH_WATCHDOG_S": ("bench", "2400", "seconds after which a bench run that is still going dumps its Python stacks and exits non-zero (0: off)"),
    "DSQ_FORCE_DIST": ("bench", "off", "take the distributed pipeline with one rank"),
    "DSQ_BENCH_NO_PLUGIN": ("bench", "off", "skip the drop-in-path measurement"),
    "DSQ_BENCH_NO_C5_FULL": ("bench", "off", "skip the full-size c5 measurement of the default run"),
}
BEGIN, END = "<!-- knobs:begin (generated by tools/knobs.py) -->", "<!-- knobs:end -->"


def scan():
    found = {}
    for base in ("pydeseq2_amd", "bench.py", "__graft_entry__.py"):
        paths = [os.path.join(ROOT, base)] if base.endswith(".py") else [
            os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(ROOT, base)) for f in fs
            if f.endswith((".py", ".hip", ".h"))]
        for p in paths:
            for i, line in enumerate(open(p, errors="replace"), 1):
                for m in re.finditer(r'(?:getenv\("|environ\.get\("|environ\[")(DSQ_[A-Z0-9_]+)"', line):
                    where = os.path.relpath(p, ROOT)  # (the file, not the line: the table must not go stale with every edit)
                    if where not in found.setdefault(m.group(1), []):
                        found[m.group(1)].append(where)
    return found


def table(found):
    rows = ["| knob | kind | default | effect | read at |", "|---|---|---|---|---|"]
    for k in sorted(KNOBS):
        kind, dflt, what = KNOBS[k]
        sites = found.get(k, [])
        where = ", ".join(f"`{s}`" for s in sites[:2]) + (f" (+{len(sites) - 2})" if len(sites) > 2 else "")
        rows.append(f"| `{k}` | {kind} | {dflt} | {what} | {where} |")
    return "\n".join(rows)


def main():
    found = scan()
    missing = sorted(set(found) - set(KNOBS))
    stale = sorted(set(KNOBS) - set(found))
    if missing or stale:
        print(f"knobs read in the source but not registered: {missing}\nregistered but read nowhere: {stale}", file=sys.stderr)
        sys.exit(1)
    readme = os.path.join(ROOT, "README.md")
    text = open(readme).read()
    block = f"{BEGIN}\n{table(found)}\n{END}"
    if BEGIN in text:
        new = re.sub(re.escape(BEGIN) + r".*?" + re.escape(END), lambda _m: block, text, flags=re.S)
    else:
        new = text.rstrip("\n") + "\n\n## Environment knobs\n\nAll of them are developer / measurement switches (the product is the default); the table is generated from the\nsource by `tools/knobs.py`, and `tests/test_capi_exports.py` fails when a knob is read that is not registered there.\n\n" + block + "\n"
    if "--check" in sys.argv:
        if new != text:
            print("README.md: the knob table is stale - run python tools/knobs.py", file=sys.stderr)
            sys.exit(1)
        return
    open(readme, "w").write(new)
    print(f"{len(KNOBS)} knobs")


if __name__ == "__main__":
    main()
