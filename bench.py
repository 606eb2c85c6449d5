"""bench.py — genes/sec of the end-to-end deseq2() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4]

A "step" is one pass of the whole path (size factors -> MoM -> mu_hat -> genewise alpha ->
trend/prior -> MAP alpha -> IRLS LFC -> Cook's (+ refit) -> Wald) over one synthetic count
matrix that is already resident in HBM when the timed region starts (the upload and the
one-off int64 -> int32 gene-major transposition happen in DeseqPipeline.__init__, outside
it; the per-gene result vectors ARE copied back to the host inside it).

Multi-GPU (driver launches `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`): genes shard across ranks (every rank owns `genes` genes x all samples:
weak scaling); the two cross-gene steps are exchanged in DistDeseqPipeline (size-factor
medians by a distributed radix select whose per-sample digit histograms are all-reduced,
trend/prior on all-gathered per-gene vectors).  torch.distributed is used only as the
rendezvous / collective transport of this harness.

Prints ONE JSON line on rank 0 (contract in the task brief) with two extra objects:
  roofline     — dominant kernel (dispersion MLE/MAP, k_alpha): algorithmic bytes per launch
                 (12*N bytes per gene: int32 counts + fp64 mu_hat) / mean launch
                 duration measured with HIP events on the launch stream, vs 8 TB/s HBM peak
  cpu_baseline — the oracle (numpy/scipy restatement of the reference, joblib over host cores)
                 timed on a bounded gene sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (genes, samples, design)            BASELINE.json configs[1..3]
    "c2": (20000, 200, "2level"),
    "c3": (60000, 1000, "2level"),
    "c4": (60000, 500, "3factor"),
    "c5": (60000, 5000, "mixed"),  # not a default bench line: host generation alone takes minutes
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (256 CU x 128 flop/clk x 2.4 GHz)


def synth_fast(G, N, design, seed):
    """SURVEY §8(d) generator (pydeseq2_amd/synth.py)."""
    from pydeseq2_amd.synth import synth_counts

    return synth_counts(G, N, design, seed)


def cpu_baseline(counts, X, n_sample, n_jobs):
    """Oracle deseq2()+Wald on the first n_sample genes, all host cores (kind = 'port')."""
    from oracle import nbglm_oracle as orc

    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        orc.deseq2(np.ascontiguousarray(counts[:, : 2 * n_jobs]), X, n_jobs=n_jobs, keep_layers=False)  # warm the pool
        sub = np.ascontiguousarray(counts[:, :n_sample])
        t = time.perf_counter()
        orc.deseq2(sub, X, n_jobs=n_jobs, keep_layers=False)
        dt = time.perf_counter() - t
    return sub.shape[1] / dt, dt


class GlooHostComm:
    """Harness-only fallback transport (same allreduce_sum / allgather interface as RcclComm): stages the
    small exchange buffers (<= 2 MB) through the host and torch.distributed/gloo.  Used only if the RCCL
    communicator cannot be brought up on some rank; the JSON line then says so."""

    def __init__(self, ctx, dist, rank, world):
        self.ctx, self.dist, self.rank, self.world = ctx, dist, rank, world

    def allreduce_sum(self, darr):
        import torch

        n = darr.nbytes // darr.dtype.itemsize
        host = np.empty(n, dtype=darr.dtype)
        self.ctx.d2h(host, darr.ptr)
        t = torch.from_numpy(host.view(np.int32) if host.dtype == np.uint32 else host)
        self.dist.all_reduce(t)
        self.ctx.h2d(darr.ptr, host)
        return darr

    def allgather(self, dsend, drecv):
        import torch

        host = np.empty(dsend.nbytes // 8, dtype=np.float64)
        self.ctx.d2h(host, dsend.ptr)
        out = [torch.empty(len(host), dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(out, torch.from_numpy(host))
        self.ctx.h2d(drecv.ptr, np.concatenate([o.numpy() for o in out]))
        return drecv

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--genes", type=int, default=0, help="override genes per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # started by hand without a launcher: re-run under torch.distributed.run, one rank per GPU
        # (the driver launches the ranks itself and never takes this branch)
        import socket
        import subprocess

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    G, N, design = CONFIGS[args.config]
    if args.genes:
        G = args.genes

    dist = None
    if world > 1:
        # control plane of this harness only (barrier, max-over-ranks, unique-id broadcast);
        # the data-path collectives are RCCL calls made by the product through its C ABI
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)

    import pydeseq2_amd
    from pydeseq2_amd._lib import Context

    ctx = Context(local_rank)
    info = ctx.device_info()
    t_gen = time.perf_counter()
    counts, X = synth_fast(G, N, design, seed=1000 * rank + {"c2": 1, "c3": 2, "c4": 3, "c5": 4}[args.config])
    t_gen = time.perf_counter() - t_gen

    if world > 1 or os.environ.get("DSQ_FORCE_DIST"):
        from pydeseq2_amd.distributed import DistDeseqPipeline, RcclComm

        transport = "rccl"
        try:
            box = [RcclComm.unique_id(ctx) if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(box, src=0)
            comm = RcclComm(ctx, box[0], rank, world)
            ok = 1
        except Exception as e:  # noqa: BLE001 - any bring-up failure falls back, loudly
            print(f"[bench] rank {rank}: RCCL bring-up failed: {e}", file=sys.stderr)
            comm, ok = None, 0
        if dist is not None:
            import torch

            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                comm, transport = GlooHostComm(ctx, dist, rank, world), "gloo-host-fallback (RCCL bring-up failed)"
        elif comm is None:
            raise RuntimeError("RCCL communicator could not be created")
        pipe = DistDeseqPipeline(counts, X, comm=comm, ctx=ctx, keep_cooks=True)
    else:
        transport = None
        pipe = pydeseq2_amd.DeseqPipeline(counts, X, ctx=ctx, keep_cooks=True)

    def barrier():
        ctx.sync()  # hipStreamSynchronize on the engine's stream (it owns all device work)
        if dist is not None:
            dist.barrier()
            ctx.sync()

    for _ in range(args.warmup):
        res = pipe.deseq2()
    # The timed region runs the production path (no per-stage synchronisation).  The dispersion
    # kernel's launch durations are still measured live in it: HIP events recorded on the engine's
    # stream around every k_alpha launch, read after the synchronisation the launch ends with anyway.
    pipe.time_kernels = False
    pipe.kernel_log = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = pipe.deseq2()
    ctx.sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch

        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    barrier()
    klog_timed = pipe.kernel_log
    # one extra, untimed step with per-stage event timing (synchronises after every stage)
    pipe.time_kernels, pipe.collect_nfev, pipe.kernel_log = True, True, {}
    res_prof = pipe.deseq2(profile=True)
    klog_prof, pipe.time_kernels, pipe.collect_nfev = pipe.kernel_log, False, False
    # extras outside `value` (a failure here must not cost the bench line): summary tail (SURVEY 8(f)-1:
    # Cook's filter, independent filtering + BH) and apeGLM LFC shrinkage of the tested coefficient (8(f)-2)
    extras = {}
    try:
        from pydeseq2_amd.summary import lfc_shrink
        from pydeseq2_amd.summary import summary as summary_tail

        cvec = np.zeros(X.shape[1])
        cvec[-1] = 1.0
        summary_tail(res_prof, cvec, ctx=ctx)
        ctx.sync()
        t_sum = time.perf_counter()
        for _ in range(3):
            sres = summary_tail(res_prof, cvec, ctx=ctx)
        ctx.sync()
        t_sum = (time.perf_counter() - t_sum) / 3
        extras["summary_tail"] = {"ms": round(t_sum * 1e3, 3), "rejections_at_0.05": int(np.nansum(sres["padj"] < 0.05)),
                                  "cutoff_index": int(sres["info"]["j"]),
                                  "note": "padj with independent filtering on the device (not part of value)"}
        lfc_shrink(pipe, res_prof, X.shape[1] - 1)
        ctx.sync()
        t_shr = time.perf_counter()
        shr = lfc_shrink(pipe, res_prof, X.shape[1] - 1)
        ctx.sync()
        t_shr = time.perf_counter() - t_shr
        extras["lfc_shrink"] = {"ms": round(t_shr * 1e3, 3), "prior_scale": round(float(shr[3]), 6),
                                "converged_fraction": round(float(np.nanmean(shr[2])), 5),
                                "note": "apeGLM MAP LFC of the last coefficient, all genes (not part of value)"}
    except Exception as e:  # noqa: BLE001
        extras["extras_error"] = repr(e)
    barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = world * G / (dt / args.steps)

    # ---- roofline of the dominant kernel (both dispersion launches use k_alpha)
    klog = klog_timed
    # every k_alpha launch of the timed region (2 per step on all genes + 2 tiny ones on the genes
    # refitted after outlier replacement), so that avg_launch_ms is directly comparable with the
    # per-kernel average of `rocprofv3 --kernel-trace --stats` of the same command
    launches = list(klog.get("k_alpha", []))
    mean_ms = float(np.mean([ms for ms, _ in launches]))
    genes_per_launch = float(np.mean([g for _, g in launches]))
    alg_bytes = genes_per_launch * 12.0 * N + genes_per_launch * 17.0
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
    big = [(ms, g) for ms, g in launches if g > 0.5 * G]
    stage_ms = {k: round(float(np.sum([ms for ms, _ in v])), 3) for k, v in klog_prof.items()
                if k not in ("k_alpha", "grid_fallback_genes", "nfev")}
    # companion bound: the fit is fp64-ALU work, ~250 flop per sample and evaluation
    # (lgamma + digamma differences, 3 logs, Cox-Reid sums); evaluations counted by the kernel itself
    nfev_full = [e for e, g in klog_prof.get("nfev", []) if g > 0.5 * G]
    evals = float(np.mean(nfev_full)) if nfev_full else None
    full_ms = float(np.mean([ms for ms, _ in big])) if big else None
    valu = None
    if evals and full_ms:
        tflops = evals * N * 250.0 / (full_ms * 1e-3) / 1e12
        valu = {"bound": "fp64_valu", "achieved": round(tflops, 2), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tflops / FP64_VALU_PEAK_TFLOPS, 4), "evaluations_per_gene": round(evals / G, 2),
                "flop_per_sample_eval": 250}
    n_fallback = float(np.sum([x for x, _ in klog.get("grid_fallback_genes", [])])) / args.steps
    roofline = {
        "bound": "hbm", "kernel": "k_alpha (dispersion MLE/MAP, one gene per wavefront)",
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(mean_ms, 4),
        "launches_timed": len(launches),
        "full_launch_ms": round(float(np.mean([ms for ms, _ in big])), 4) if big else None,
        "pipeline_algorithmic_GBps": round(G * 104.0 * N / (dt / args.steps) / 1e9, 2),
        "kernel_ms_per_step": stage_ms, "grid_fallback_genes_per_step": n_fallback,
        "companion": valu,
    }
    traffic_file = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
    if os.path.exists(traffic_file):
        try:
            roofline["traffic"] = json.load(open(traffic_file)).get("k_alpha_hbm_bytes_per_launch")
        except Exception:
            pass

    # ---- CPU baseline on a bounded sample of the same workload
    cpu = None
    if not args.no_cpu_baseline:
        cores = os.cpu_count() or 1
        n_jobs = min(cores, 64)
        n_sample = args.cpu_sample or {"c2": 20000, "c3": 8000, "c4": 8000, "c5": 1500}[args.config]
        n_sample = min(n_sample, G)
        try:
            v, secs = cpu_baseline(counts, X, n_sample, n_jobs)
            cpu = {"value": round(v, 1), "unit": "genes/s", "cores": n_jobs, "kind": "port",
                   "sample": f"oracle (numpy/scipy restatement of the reference incl. scipy L-BFGS-B per gene, "
                             f"joblib/loky workers warmed up) on the first {n_sample} genes x {N} samples of the "
                             f"same matrix, {secs:.1f} s"}
        except Exception as e:  # noqa: BLE001 - the GPU measurement above stands on its own
            print(f"[bench] cpu_baseline failed: {e!r}", file=sys.stderr)
            cpu = None

    out = {
        "metric": "genes/sec end-to-end deseq2() (size factors->dispersion->IRLS->Wald)",
        "value": round(value, 1), "unit": "genes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.config}: {G} genes x {N} samples per GPU, design {design} "
                               f"(p={X.shape[1]}), NB counts (SURVEY 8d generator)",
                   "genes_per_gpu": G, "samples": N, "p": int(X.shape[1]), "collectives": transport,
                   "device": info["name"], "arch": info["arch"]},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "stage_wall_ms_profiled_step": {k: round(v * 1e3, 3) for k, v in res_prof.timings.items()},
        "speedup_vs_cpu_baseline": round(value / cpu["value"], 1) if cpu else None,
    }
    out.update(extras)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
