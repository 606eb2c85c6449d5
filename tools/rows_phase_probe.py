"""Developer aid: per-phase cycle breakdown of k_alpha_rows (build/libdeseq_hip_rowsph.so, -DDSQ_ROWS_PHASES).
Phases: 0 setup/exit, 1 refill (queue, staging of new genes), 2 eval head + tail-count sums, 3 sample loop,
4 reductions + p x p algebra, 5 L-BFGS-B machine + results."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pydeseq2_amd  # noqa: E402
from pydeseq2_amd._lib import Context, load  # noqa: E402

NAMES = ["setup/exit", "refill+staging", "head+tail sums", "sample loop", "reduce+algebra", "machine+results", "-", "-"]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    G, N, design = bench.CONFIGS[cfg]
    ctx = Context(0)
    counts, X = bench.synth_fast(G, N, design, seed=bench.SEEDS[cfg])
    pipe = pydeseq2_amd.DeseqPipeline(counts, X, ctx=ctx)
    pipe.deseq2()
    lib = load()
    buf = (C.c_ulonglong * 8)()
    lib.dsq_debug_rows_phase_read(buf, 1)
    pipe.deseq2()
    ctx.sync()
    lib.dsq_debug_rows_phase_read(buf, 1)
    tot = float(sum(buf[:6]))
    print(f"{cfg}: {G} x {N}; cycles (clock64) summed over waves and both launches (MLE + MAP)")
    for k, n in enumerate(NAMES[:6]):
        print(f"  {k} {n:18s} {buf[k] / 1e6:12.1f} M  {100.0 * buf[k] / tot:6.2f} %")
    if buf[7]:
        print(f"  waves {buf[7]}, mean lifetime {tot / buf[7] / 1e3:.1f} k, longest {buf[6] / 1e3:.1f} k "
              f"(mean / longest = {tot / buf[7] / buf[6]:.3f}: the share of the kernel's duration an average wave is busy)")


if __name__ == "__main__":
    main()
