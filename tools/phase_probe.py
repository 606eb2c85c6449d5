"""Developer aid: per-phase cycle breakdown of the dispersion kernel (k_alpha).

Build the instrumented library here (cross-compile), run this script on the GPU box:
    make -C pydeseq2_amd/csrc phase            -> build/libdeseq_hip_phase.so
    gpurun -- 'DSQ_LIB=build/libdeseq_hip_phase.so python tools/phase_probe.py c3'
Phases: 0 staging/epilogue, 1 NLL constant, 2 eval head (exp/log/lgamma(a)/memo table), 3 sample loop,
4 cross-lane reductions, 5 eval tail (Cholesky, prior), 6 L-BFGS-B machine, 7 after the loop.
"""
import ctypes as C
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pydeseq2_amd  # noqa: E402
from pydeseq2_amd._lib import Context, load  # noqa: E402

NAMES = ["stage/epilogue", "nll const", "eval head", "sample loop", "reductions", "eval tail", "lbfgsb machine", "post"]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    G, N, design = bench.CONFIGS[cfg]
    G = int(os.environ.get("PROBE_GENES", G))
    ctx = Context(0)
    counts, X = bench.synth_fast(G, N, design, seed=2)
    pipe = pydeseq2_amd.DeseqPipeline(counts, X, ctx=ctx)
    pipe.deseq2()
    lib = load()
    buf = (C.c_ulonglong * 16)()
    lib.dsq_debug_phase_read(buf, 1)
    pipe.deseq2()
    ctx.sync()
    lib.dsq_debug_phase_read(buf, 1)
    tot = float(sum(buf[:8]))
    print(f"{cfg}: {G} x {N}; cycles summed over waves and both launches (MLE + MAP)")
    for k, n in enumerate(NAMES):
        print(f"  {k} {n:16s} {buf[k] / 1e6:12.1f} Mcycles  {100.0 * buf[k] / tot:6.2f} %")
    if hasattr(lib, "dsq_debug_phase_read_irls"):
        lib.dsq_debug_phase_read_irls(buf, 1)
        pipe.deseq2()
        ctx.sync()
        lib.dsq_debug_phase_read_irls(buf, 1)
        tot = float(sum(buf[:9]))
        names = ["stage/epilogue", "init", "sweep pre (cell tables)", "sweep sample loop", "sweep reduce/rebuild",
                 "deviance, loop control", "finish (hat, cooks, wald)", "Cholesky", "solve"]
        print("k_irls (all launches of one step):")
        for k, n in enumerate(names):
            print(f"  {k} {n:26s} {buf[k] / 1e6:12.1f} Mcycles  {100.0 * buf[k] / tot:6.2f} %")


if __name__ == "__main__":
    main()
