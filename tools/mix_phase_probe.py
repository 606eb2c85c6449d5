"""Developer aid: per-phase cycle breakdown of k_alpha_mix (build/libdeseq_hip_mixph.so: make -C pydeseq2_amd/csrc mixph).
    DSQ_LIB=build/libdeseq_hip_mixph.so python tools/mix_phase_probe.py [genes]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pydeseq2_amd  # noqa: E402
from pydeseq2_amd._lib import Context, load  # noqa: E402

NAMES = ["queue / exit", "-", "stage: gather, mu_hat row", "stage: tail counts", "eval head (tail sums)", "sample loop",
         "folds + reductions", "p x p algebra", "optimiser step", "result / park"]


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 7500
    _, N, design = bench.CONFIGS["c5"]
    ctx = Context(0)
    counts, X = bench.synth_fast(G, N, design, seed=bench.SEEDS["c5"])
    pipe = pydeseq2_amd.DeseqPipeline(counts, X, ctx=ctx)
    pipe.deseq2()
    lib = load()
    buf = (C.c_ulonglong * 12)()
    lib.dsq_debug_mix_phase_read(buf, 1)
    pipe.deseq2()
    ctx.sync()
    lib.dsq_debug_mix_phase_read(buf, 1)
    tot = float(sum(buf[:10]))
    print(f"c5: {G} x {N}; cycles (clock64) summed over wavefronts and all launches of one step")
    for k, n in enumerate(NAMES):
        print(f"  {k} {n:24s} {buf[k] / 1e6:12.1f} M  {100.0 * buf[k] / tot:6.2f} %")
    if buf[11]:
        print(f"  wavefronts {buf[11]}, mean lifetime {tot / buf[11] / 1e3:.1f} k cycles, longest {buf[10] / 1e3:.1f} k")


if __name__ == "__main__":
    main()
