"""World-size-2 CPU tests (gloo) of the gene-sharded multi-GPU protocol: the distributed radix
select for the size-factor medians and the trend-input gathering give the single-process answer."""
import os
import socket

import numpy as np
import pytest

from oracle import nbglm_oracle as orc
from pydeseq2_amd import distributed as D
from tests.dist_numpy_ops import NumpySfOps, f64_keys, key_f64


def test_key_mapping_roundtrip_and_order():
    v = np.array([-np.inf, -3.5, -1e-300, -0.0, 0.0, 1e-300, 2.0, np.inf])
    k = f64_keys(v)
    assert (np.diff(k.astype(np.float64)) >= 0).all()
    assert (key_f64(k) == v).all()


def test_median_protocol_single_rank_matches_numpy():
    counts, X = orc.synth_counts(301, 24, "2level", 3)
    lm, keep = orc.logmeans_and_filter(counts)
    sf_ref = orc.size_factors_ratio(counts)[0]
    ops = NumpySfOps(counts, lm)
    sf = D.median_select_protocol(ops, lambda x: x)
    np.testing.assert_allclose(sf, sf_ref, rtol=1e-14)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts, X = orc.synth_counts(400, 20, "2level", 11)
    shard = np.array_split(np.arange(400), world)[rank]
    mine = counts[:, shard]
    lm, _ = orc.logmeans_and_filter(mine)

    def allreduce(x):
        t = torch.from_numpy(x.astype(np.int64))
        dist.all_reduce(t)
        return t.numpy().astype(x.dtype)

    sf = D.median_select_protocol(NumpySfOps(mine, lm), allreduce)
    # trend inputs: gather padded vectors
    G = len(shard) + 3
    n_gathers = [0]

    def allgather_f64(x):
        n_gathers[0] += 1
        out = [torch.zeros(len(x), dtype=torch.float64) for _ in range(world)]
        dist.all_gather(out, torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)))
        return torch.cat(out).numpy()

    class NumpyTrendOps:  # what DistDeseqPipeline._gather_trend_inputs does with dsq_dev_pack2 / dsq_dev_unzip2
        def pack(self):
            return D.pack_trend_inputs(np.full(len(shard), 0.1 * (rank + 1)), np.arange(len(shard)) + 1.0, G)

        def unzip(self, recv):
            return D.unzip_trend_inputs(recv, world, G)

    gw_all, nm_all = D.trend_gather_protocol(NumpyTrendOps(), allgather_f64)
    assert n_gathers[0] == 1  # ONE collective for both per-gene vectors
    allv = np.stack([gw_all.reshape(world, G), nm_all.reshape(world, G)], axis=1)  # [world][2][G]
    # the two-collective protocol: all-gather of log means, medians of the rank's own samples, all-gather of medians
    class NumpySampleOps:
        def local_logmeans(self):
            return lm

        def medians(self, lm_all):
            n0, n1 = D.sample_block(rank, world, counts.shape[0])
            with np.errstate(divide="ignore", invalid="ignore"):
                ratios = np.log(counts[n0:n1].astype(float)) - lm_all[None, :]
            med = np.median(ratios[:, np.isfinite(lm_all)], axis=1)
            out = np.zeros(-(-counts.shape[0] // world))
            out[: n1 - n0] = np.exp(med)
            return out

        def finish(self, allm):
            pad = -(-counts.shape[0] // world)
            allm = allm.reshape(world, pad)
            return np.concatenate([allm[r, : D.sample_block(r, world, counts.shape[0])[1]
                                             - D.sample_block(r, world, counts.shape[0])[0]] for r in range(world)])

    def allgather(x):
        out = [torch.zeros(len(x), dtype=torch.float64) for _ in range(world)]
        dist.all_gather(out, torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)))
        return torch.cat(out).numpy()

    sf2 = D.sample_shard_protocol(NumpySampleOps(), allgather)
    q.put((rank, sf, allv, sf2))
    dist.destroy_process_group()


def test_world2_gloo_protocol_matches_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    counts, X = orc.synth_counts(400, 20, "2level", 11)
    sf_ref = orc.size_factors_ratio(counts)[0]
    for rank, sf, allv, sf2 in res:
        np.testing.assert_allclose(sf, sf_ref, rtol=1e-14)
        np.testing.assert_allclose(sf2, sf_ref, rtol=1e-14)  # two-collective protocol
        assert np.isnan(allv[:, :, -3:]).all() and not np.isnan(allv[:, :, :-3]).any()
        assert np.allclose(allv[1, 0, :-3], 0.2)


def _tcp_worker(rank, world, port, q):
    """Control plane of the torch-free harness (pydeseq2_amd.distributed.TcpControl) carrying the size-factor
    protocol's histograms and the trend inputs, as HostStagedComm does for device buffers."""
    ctl = D.TcpControl(rank, world, "127.0.0.1", port, timeout=60)
    counts, X = orc.synth_counts(400, 20, "2level", 11)
    shard = np.array_split(np.arange(400), world)[rank]
    mine = counts[:, shard]
    lm, _ = orc.logmeans_and_filter(mine)

    def allreduce(x):
        parts = [np.frombuffer(b, dtype=x.dtype).reshape(x.shape) for b in ctl.allgather_bytes(x.tobytes())]
        return np.sum(parts, axis=0).astype(x.dtype)

    sf = D.median_select_protocol(NumpySfOps(mine, lm), allreduce)
    got = ctl.allgather_bytes(bytes([rank]) * (rank + 1))
    uid = ctl.bcast_bytes(b"u" * 128 if rank == 0 else b"")
    mx = ctl.max_float(1.5 * rank)
    ctl.barrier()
    q.put((rank, sf, got, uid, mx))
    ctl.close()


@pytest.mark.parametrize("world", [2, 3])
def test_tcp_control_plane_carries_the_protocol(world):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tcp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    counts, X = orc.synth_counts(400, 20, "2level", 11)
    sf_ref = orc.size_factors_ratio(counts)[0]
    for rank, sf, got, uid, mx in res:
        np.testing.assert_allclose(sf, sf_ref, rtol=1e-14)
        assert got == [bytes([r]) * (r + 1) for r in range(world)]
        assert uid == b"u" * 128 and mx == 1.5 * (world - 1)


def test_tcp_hub_survives_connections_that_are_not_its_ranks():
    """A rank of another job that probes this hub's port range (reads the greeting, sees another base port, hangs up), a
    port scanner or a health probe must not take rank 0 down: the failed handshake is dropped and the hub keeps
    accepting; a connection that claims a rank outside 1..world-1 or a rank that is already connected is refused."""
    import socket
    import threading
    import time

    port = _free_port()
    out = {}

    def hub():
        try:
            out["ctl"] = D.TcpControl(0, 2, "127.0.0.1", port, timeout=30)
        except BaseException as e:  # noqa: BLE001
            out["err"] = e

    th = threading.Thread(target=hub)
    th.start()

    def connect():
        for _ in range(200):
            for pt in range(port + 1, port + 32):
                try:
                    return socket.create_connection(("127.0.0.1", pt), timeout=1)
                except OSError:
                    continue
            time.sleep(0.02)
        raise AssertionError("hub not reachable")

    c = connect()
    c.close()                                   # a probe that hangs up at once
    c = connect()
    c.recv(12)
    c.close()                                   # another job's rank: reads the greeting, hangs up
    c = connect()
    c.recv(12)
    c.sendall((7).to_bytes(4, "little"))        # a rank this job does not have
    time.sleep(0.1)
    c.close()
    real = D.TcpControl(1, 2, "127.0.0.1", port, timeout=30)
    th.join(timeout=30)
    assert "err" not in out, out.get("err")
    got = {}
    t2 = threading.Thread(target=lambda: got.setdefault("v", out["ctl"].allgather_bytes(b"a")))
    t2.start()
    assert real.allgather_bytes(b"b") == [b"a", b"b"]
    t2.join(timeout=30)
    assert got["v"] == [b"a", b"b"]
    real.close()
    out["ctl"].close()


# ---------------------------------------------------------------------------------------------------------------
# bench.py's multi-GPU plan: which block of which matrix a rank works on, and the hand launcher
@pytest.mark.parametrize("config,genes,world", [("c2", 900, 2), ("c2", 1000, 3), ("c5", 1300, 2), ("c5", 1700, 3)])
def test_bench_strong_scaling_plan_tiles_the_named_matrix(config, genes, world, monkeypatch):
    """--scaling strong (bench.py's default for --gpus > 1): the ranks' gene blocks tile the configuration's matrix
    and each rank's sample block is the matching row block of that same matrix (two-collective size factors).  c5 comes
    from the tiled generator: a rank builds its blocks without the whole matrix."""
    import bench
    from pydeseq2_amd.distributed import sample_block
    from pydeseq2_amd.synth import synth_counts, synth_counts_block

    N0 = bench.CONFIGS[config][1]
    monkeypatch.setitem(bench.CONFIGS, config, (bench.CONFIGS[config][0], min(N0, 260), bench.CONFIGS[config][2]))
    N, design = bench.CONFIGS[config][1:]
    if config == "c5":
        full, X = synth_counts_block(genes, N, design, bench.SEEDS[config])
    else:
        full, X = synth_counts(genes, N, design, bench.SEEDS[config])
    parts, total = [], 0
    for r in range(world):
        counts, Xr, samp, G, G_total, gen = bench.plan_rank_data(config, genes, "strong", r, world)
        assert np.array_equal(Xr, X) and G_total == genes and counts.shape == (N, G)
        n0, n1 = sample_block(r, world, N)
        assert np.array_equal(samp, full[n0:n1])
        parts.append(counts)
        total += G
    assert total == genes and np.array_equal(np.concatenate(parts, axis=1), full)
    # weak: every rank its own matrix of the configuration's size
    c0 = bench.plan_rank_data(config, genes, "weak", 0, world)
    c1 = bench.plan_rank_data(config, genes, "weak", 1, world)
    assert c0[0].shape == c1[0].shape == (N, genes) and c0[4] == genes * world and not np.array_equal(c0[0], c1[0])
    assert c0[2] is None


def test_tiled_generator_blocks_agree_on_overlaps():
    from pydeseq2_amd.synth import synth_counts_block

    full, X = synth_counts_block(1234, 300, "mixed", 9)
    a, _ = synth_counts_block(1234, 300, "mixed", 9, genes=(400, 1100))
    b, _ = synth_counts_block(1234, 300, "mixed", 9, samples=(130, 251))
    c, _ = synth_counts_block(1234, 300, "mixed", 9, genes=(499, 502), samples=(124, 127))
    assert np.array_equal(a, full[:, 400:1100]) and np.array_equal(b, full[130:251])
    assert np.array_equal(c, full[124:127, 499:502])
    assert X.shape == (300, 8) and full.dtype == np.int64 and (full >= 0).all()


def test_bench_hand_launcher_starts_one_process_per_rank(tmp_path):
    """`python bench.py --gpus N` without a launcher: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the
    worst exit code comes back (no torch.distributed.run in between)."""
    import bench

    script = tmp_path / "child.py"
    script.write_text(
        "import os, sys\n"
        "r = os.environ['RANK']\n"
        "open(os.path.join(sys.argv[1], 'rank' + r), 'w').write(' '.join(os.environ[k] for k in "
        "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n"
        "sys.exit(3 if (r == '1' and len(sys.argv) > 2) else 0)\n")
    assert bench.launch_local_ranks(3, [str(tmp_path)], script=str(script)) == 0
    got = [(tmp_path / f"rank{r}").read_text().split() for r in range(3)]
    assert [g[0] for g in got] == ["0", "1", "2"] and all(g[2] == "3" and g[3] == "127.0.0.1" for g in got)
    assert len({g[4] for g in got}) == 1
    assert bench.launch_local_ranks(2, [str(tmp_path), "fail"], script=str(script)) == 3
