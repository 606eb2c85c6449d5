"""Gene-sharded multi-GPU deseq2(): one process per GPU, exchanges over RCCL (xGMI).

Every rank owns a block of genes (all samples).  All per-gene kernels are local; only two steps
of the path need the other ranks' genes:

1. size factors — the per-sample MEDIAN over all genes of log(count) - logmean
   (preprocessing.py:59-102).  Each rank builds order-preserving keys for its genes; the
   radix select runs one 8-bit digit at a time, and the per-sample 256-bin digit histograms
   are summed across ranks with an all-reduce (2*N*256 uint32 = 2 MB at N = 1000, 8 passes).
   Every rank then holds identical size factors; no gene data is exchanged.
2. dispersion trend + prior (dds.py:799-884) — all-gather of (genewise dispersion, normalised
   mean) per gene, then every rank fits the identical trend on the gathered vectors.

The exchange layer is abstract (`allreduce_sum`, `allgather`) so that the same protocol code is
exercised on CPU by tests with a gloo backend; the product binding is `RcclComm`, which calls
librccl on device buffers through the C ABI (no PyTorch in the product path).
"""
from __future__ import annotations

import ctypes as C
import socket
import time

import numpy as np

from . import trend as _trend
from ._lib import Context, DeviceArray
from .pipeline import DeseqPipeline

_vp = C.c_void_p


# ------------------------------------------------------------------ protocol (backend agnostic)
def median_select_protocol(ops, allreduce_sum):
    """Distributed per-sample median: `ops` provides the local passes, `allreduce_sum(x)` returns
    the element-wise sum of x over all ranks (in place is fine).  Mirrors k_row_median."""
    total = allreduce_sum(ops.count())
    ops.init(total)
    for shift in range(56, -8, -8):
        ops.pick(allreduce_sum(ops.hist(shift)), shift)
    return ops.finish(total)


def sample_shard_protocol(ops, allgather):
    """Size factors with TWO collectives ("scalable variant"): every rank also holds a block of
    SAMPLES with the genes of all ranks.  (1) all-gather of the per-gene log means (each rank computes them for
    the genes it owns, all samples) -> every rank has the log means of every gene; (2) the rank takes the medians
    of its own samples over all genes - locally, with the single-GPU kernel - and an all-gather of those N / world
    values gives every rank all size factors.  `ops`: local_logmeans() -> padded vector, medians(all_logmeans) ->
    padded vector of this rank's samples, finish(all_medians) -> size factors [N]; `allgather(x)` concatenates the
    ranks' equally sized vectors in rank order."""
    return ops.finish(allgather(ops.medians(allgather(ops.local_logmeans()))))


def sample_block(rank, world, N):
    """[start, stop) of the samples whose medians `rank` computes in the two-collective protocol."""
    cuts = np.linspace(0, N, world + 1).astype(int)
    return int(cuts[rank]), int(cuts[rank + 1])


def trend_inputs_padded(gw_nz, nm_nz, G):
    """Fixed-size per-rank trend inputs: the Gn non-zero genes, then NaN padding (a NaN mean gives
    a NaN covariate, which the fit drops exactly like the reference drops non-finite covariates,
    dds.py:1225-1231)."""
    gw = np.full(G, np.nan)
    nm = np.full(G, np.nan)
    gw[: len(gw_nz)] = gw_nz
    nm[: len(nm_nz)] = nm_nz
    return gw, nm


def trend_gather_protocol(ops, allgather):
    """Trend / prior inputs of ALL ranks with ONE collective: a rank packs its two per-gene vectors (raw genewise
    dispersions, normalised means), each NaN-padded to the largest shard, into one send buffer [2 * Gpad]; the all-gather
    gives [world][2][Gpad], which `ops.unzip` splits into the two [world * Gpad] vectors the trend and prior kernels
    read.  `ops`: pack() -> send buffer, unzip(recv) -> (gw_all, nm_all)."""
    return ops.unzip(allgather(ops.pack()))


def pack_trend_inputs(gw_nz, nm_nz, G):
    """numpy form of the packing (CPU tests, host-side callers): concatenated trend_inputs_padded vectors."""
    return np.concatenate(trend_inputs_padded(gw_nz, nm_nz, G))


def unzip_trend_inputs(recv, world, G):
    """numpy form of the split: recv [world * 2 * G] -> (gw_all, nm_all), each [world * G] in rank order."""
    v = np.asarray(recv).reshape(world, 2, G)
    return v[:, 0, :].reshape(-1).copy(), v[:, 1, :].reshape(-1).copy()


# ------------------------------------------------------------------ RCCL binding
class _CStdoutToStderr:
    """librccl prints a version banner on C stdout at communicator creation; route it to stderr so
    that programs whose stdout is machine-read (bench.py's single JSON line) stay clean."""

    def __enter__(self):
        import os
        import sys

        sys.stdout.flush()
        self._libc = C.CDLL(None)
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import os

        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class RcclComm:
    """RCCL communicator bound to a dsq context (device buffers in, device buffers out)."""

    def __init__(self, ctx: Context, uid: bytes, rank: int, world: int):
        self.ctx, self.rank, self.world = ctx, rank, world
        assert len(uid) == 128
        with _CStdoutToStderr():
            ctx.call("dsq_comm_init", C.c_char_p(uid), int(rank), int(world))

    @staticmethod
    def unique_id(ctx: Context) -> bytes:
        buf = C.create_string_buffer(128)
        with _CStdoutToStderr():
            ctx.call("dsq_comm_unique_id", buf, 128)
        return buf.raw

    def allreduce_sum(self, darr: DeviceArray):
        dtype = 0 if darr.dtype == np.uint32 else 1
        n = darr.nbytes // darr.dtype.itemsize
        self.ctx.call("dsq_comm_allreduce_sum", _vp(darr.ptr), C.c_size_t(n), dtype)
        return darr

    def allgather(self, dsend: DeviceArray, drecv: DeviceArray):
        self.ctx.call("dsq_comm_allgather", _vp(dsend.ptr), _vp(drecv.ptr), C.c_size_t(dsend.nbytes))
        return drecv

    def info(self):
        """(nranks, rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int(0), C.c_int(0)
        self.ctx.call("dsq_comm_info", C.byref(n), C.byref(r))
        return int(n.value), int(r.value)

    def close(self):
        self.ctx.call("dsq_comm_destroy")


class TcpControl:
    """Torch-free control plane of a one-process-per-GPU job (rank 0 is a star hub on a TCP socket): byte
    all-gather, broadcast, barrier.  Carries the RCCL unique id at start-up, the barriers / max-over-ranks of
    bench.py, and — only if RCCL cannot be brought up on some rank — the small exchange buffers themselves
    (`HostStagedComm`).  The launcher's MASTER_PORT is usually occupied by its own rendezvous store, so the
    hub listens on the first free port of [port + 1, port + 32) and clients find it by its greeting."""

    MAGIC = b"DSQCTL01"

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29500, timeout: float = 180.0):
        self.rank, self.world = rank, world
        self.peers, self.sock = [], None
        if world == 1:
            return
        ports = [port + 1 + k for k in range(31)]
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            for pt in ports:
                try:
                    srv.bind((addr, pt))
                    break
                except OSError:
                    continue
            else:
                raise OSError(f"no free control port in {ports[0]}..{ports[-1]}")
            srv.listen(world)
            srv.settimeout(timeout)
            peers = {}
            deadline = time.time() + timeout
            while len(peers) < world - 1:
                left = deadline - time.time()
                if left <= 0:
                    srv.close()
                    raise TimeoutError(f"control plane: {world - 1 - len(peers)} rank(s) did not connect")
                srv.settimeout(left)
                try:
                    c, _a = srv.accept()
                except socket.timeout:
                    continue
                # A connection that is not one of this job's ranks (a rank of another job probing the port range reads
                # the greeting, sees another base port and hangs up; a port scanner; a health probe) must not take the
                # hub down: its handshake fails, the socket is closed, the hub keeps accepting until the deadline.
                # The handshake is served inline, so it gets a SHORT deadline (a real rank answers the greeting within
                # a round trip; a silent connection may not hold up the ranks waiting in the backlog), and the rank is
                # only admitted once it has the hub's acknowledgement: a rank whose handshake was cut reconnects.
                try:
                    c.settimeout(min(0.5, max(left, 0.1)))
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    # (the job's base port: two jobs with overlapping port ranges do not adopt each other's ranks)
                    c.sendall(self.MAGIC + int(port).to_bytes(4, "little"))
                    r = int.from_bytes(self._recvn(c, 4), "little")
                    if not (1 <= r < world) or r in peers:
                        raise ConnectionError(f"control plane: unexpected rank {r}")
                    c.sendall(b"\x01")
                    c.settimeout(timeout)
                except (OSError, ConnectionError):
                    c.close()
                    continue
                peers[r] = c
            srv.close()
            self.peers = [peers[r] for r in range(1, world)]
        else:
            t0 = time.time()
            while self.sock is None:
                for pt in ports:
                    try:
                        c = socket.create_connection((addr, pt), timeout=2)
                        c.settimeout(5)
                        if self._recvn(c, len(self.MAGIC) + 4) == self.MAGIC + int(port).to_bytes(4, "little"):
                            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            c.sendall(int(rank).to_bytes(4, "little"))
                            if self._recvn(c, 1) == b"\x01":  # admitted (else: the hub dropped this handshake - retry)
                                c.settimeout(timeout)
                                self.sock = c
                                break
                        c.close()
                    except (OSError, ConnectionError):
                        continue
                if self.sock is None:
                    if time.time() - t0 > timeout:
                        raise TimeoutError("control plane: rank 0 not reachable")
                    time.sleep(0.05)

    @staticmethod
    def _recvn(c, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("control plane: peer closed the connection")
            buf += chunk
        return bytes(buf)

    @classmethod
    def _send(cls, c, b):
        c.sendall(len(b).to_bytes(8, "little") + b)

    @classmethod
    def _recv(cls, c):
        return cls._recvn(c, int.from_bytes(cls._recvn(c, 8), "little"))

    def allgather_bytes(self, b: bytes):
        """Every rank contributes `b`; every rank gets the list of all contributions in rank order."""
        if self.world == 1:
            return [b]
        if self.rank == 0:
            parts = [b] + [self._recv(c) for c in self.peers]
            blob = b"".join(len(x).to_bytes(8, "little") + x for x in parts)
            for c in self.peers:
                self._send(c, blob)
            return parts
        self._send(self.sock, b)
        blob, parts, o = self._recv(self.sock), [], 0
        while o < len(blob):
            n = int.from_bytes(blob[o:o + 8], "little")
            parts.append(blob[o + 8:o + 8 + n])
            o += 8 + n
        return parts

    def bcast_bytes(self, b):
        return self.allgather_bytes(b if self.rank == 0 else b"")[0]

    def barrier(self):
        self.allgather_bytes(b"")

    def max_float(self, v: float) -> float:
        return max(float(np.frombuffer(x, dtype=np.float64)[0]) for x in self.allgather_bytes(np.float64(v).tobytes()))

    def close(self):
        for c in self.peers + ([self.sock] if self.sock is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = [], None


class HostStagedComm:
    """Same interface as RcclComm, but the (small) exchange buffers travel device -> host -> TcpControl -> host
    -> device.  A fallback for a node on which RCCL cannot be initialised, and the transport of the CPU tests."""

    def __init__(self, ctx, control: TcpControl):
        self.ctx, self.control, self.rank, self.world = ctx, control, control.rank, control.world

    def allreduce_sum(self, darr):
        n = darr.nbytes // darr.dtype.itemsize
        host = np.empty(n, dtype=darr.dtype)
        self.ctx.d2h(host, darr.ptr)
        parts = [np.frombuffer(x, dtype=darr.dtype) for x in self.control.allgather_bytes(host.tobytes())]
        self.ctx.h2d(darr.ptr, np.sum(parts, axis=0).astype(darr.dtype))
        return darr

    def allgather(self, dsend, drecv):
        host = np.empty(dsend.nbytes // dsend.dtype.itemsize, dtype=dsend.dtype)
        self.ctx.d2h(host, dsend.ptr)
        self.ctx.h2d(drecv.ptr, np.frombuffer(b"".join(self.control.allgather_bytes(host.tobytes())), dtype=dsend.dtype))
        return drecv

    def close(self):
        pass


def bring_up_comm(ctx: Context, control: TcpControl):
    """RCCL communicator over the ranks of `control` (unique id broadcast over the control plane); if any rank
    fails to initialise RCCL every rank falls back to the host-staged transport.  Returns (comm, transport)."""
    uid, err = b"", ""
    if control.rank == 0:
        try:
            uid = RcclComm.unique_id(ctx)
        except Exception as e:  # noqa: BLE001
            err = repr(e)
    uid = control.bcast_bytes(uid)
    comm = None
    if len(uid) == 128:
        try:
            comm = RcclComm(ctx, uid, control.rank, control.world)
        except Exception as e:  # noqa: BLE001
            err = repr(e)
    oks = control.allgather_bytes(b"1" if comm is not None else b"0" + err.encode())
    if all(x == b"1" for x in oks):
        return comm, "rccl"
    why = "; ".join(f"rank {r}: {x[1:].decode(errors='replace')}" for r, x in enumerate(oks) if x != b"1")
    if comm is not None:
        try:
            comm.close()
        except Exception:  # noqa: BLE001
            pass
    return HostStagedComm(ctx, control), f"host-staged tcp fallback (RCCL bring-up failed: {why})"


def exchange_unique_id(ctx: Context, rank: int, world: int, addr: str, port: int, timeout=120.0) -> bytes:
    """Torch-free bootstrap kept for callers that only need the id: rank 0 creates the RCCL unique id and the
    control plane broadcasts it."""
    control = TcpControl(rank, world, addr, port, timeout)
    try:
        return control.bcast_bytes(RcclComm.unique_id(ctx) if rank == 0 else b"")
    finally:
        control.close()


# ------------------------------------------------------------------ device passes of the median
class _DeviceSfOps:
    def __init__(self, pipe: DeseqPipeline, d_lm, d_mask=None):
        self.p, ctx = pipe, pipe.ctx
        N, G = pipe.N, pipe.G
        # keys only for this rank's usable genes (finite logmean, inside the mask if there is one): [N][Gu]
        self.d_keys = pipe._pooled((N * G,), np.uint64)
        d_idx = pipe._pooled((G + 2,), np.int32)
        gu = C.c_int(0)
        ctx.call("dsq_dev_sf_keys_compact", _vp(pipe.d_raw.ptr), pipe._count_type, N, G, _vp(d_lm.ptr),
                 _vp(d_mask.ptr) if d_mask is not None else None, _vp(d_idx.ptr), _vp(self.d_keys.ptr), C.byref(gu))
        self.Gu = int(gu.value)
        self.d_cnt = pipe._pooled((N,), np.uint32)
        self.d_prefix = pipe._pooled((2 * N,), np.uint64)
        self.d_rank = pipe._pooled((2 * N,), np.uint32)
        self.d_hist = pipe._pooled((2 * N * 256,), np.uint32)

    def count(self):
        self.p.ctx.call("dsq_dev_sf_count", _vp(self.d_keys.ptr), self.p.N, self.Gu, _vp(self.d_cnt.ptr))
        return self.d_cnt

    def init(self, total):
        self.p.ctx.call("dsq_dev_sf_init", _vp(total.ptr), self.p.N, _vp(self.d_prefix.ptr), _vp(self.d_rank.ptr))

    def hist(self, shift):
        self.p.ctx.call("dsq_dev_sf_hist", _vp(self.d_keys.ptr), self.p.N, self.Gu, _vp(self.d_prefix.ptr),
                        int(shift), _vp(self.d_hist.ptr))
        return self.d_hist

    def pick(self, hist, shift):
        self.p.ctx.call("dsq_dev_sf_pick", _vp(hist.ptr), self.p.N, int(shift), _vp(self.d_prefix.ptr),
                        _vp(self.d_rank.ptr))

    def finish(self, total):
        d_sf = self.p._pooled((self.p.N,), np.float64)
        self.p.ctx.call("dsq_dev_sf_finish", _vp(self.d_prefix.ptr), _vp(total.ptr), self.p.N, _vp(d_sf.ptr))
        return d_sf


class _DeviceSampleShardOps:
    """Device side of sample_shard_protocol: the rank's sample block [n_r x (world * Gpad)] in the padded global
    gene order (rank-major; padding columns hold zeros and get a log mean of -inf, which excludes them)."""

    def __init__(self, pipe, d_lm):
        self.p, self.d_lm = pipe, d_lm

    def local_logmeans(self):
        p = self.p
        d_send = p._pooled((p.Gpad,), np.float64)
        p.ctx.h2d(d_send.ptr, np.full(p.Gpad, -np.inf)) if p.G < p.Gpad else None
        p.ctx.call("dsq_d2d", _vp(d_send.ptr), _vp(self.d_lm.ptr), C.c_size_t(8 * p.G))
        return d_send

    def medians(self, d_lm_all):
        p = self.p
        n_r, Gall = p._samp_rows, p.Gpad * p.comm.world
        d_part = p._pooled((p._samp_pad,), np.float64)
        p.ctx.memset(d_part.ptr, 0, 8 * p._samp_pad)
        if n_r:
            if p._samp_work is None:
                p._samp_work = DeviceArray(p.ctx, (p.ctx.lib.dsq_size_factors_work_doubles(n_r, Gall),), np.float64)
            p.ctx.call("dsq_dev_size_factors", _vp(p._d_samp.ptr), 0, n_r, Gall, _vp(d_lm_all.ptr), None,
                       _vp(p._samp_work.ptr), _vp(d_part.ptr))
        return d_part

    def finish(self, d_all):
        p = self.p
        W = p.comm.world
        allv = p._down(d_all, W * p._samp_pad).reshape(W, p._samp_pad)
        sf = np.concatenate([allv[r, : sample_block(r, W, p.N)[1] - sample_block(r, W, p.N)[0]] for r in range(W)])
        return p._up(sf)


class DistDeseqPipeline(DeseqPipeline):
    """DeseqPipeline over a gene shard; `comm` provides allreduce_sum / allgather on device arrays.

    ``sample_shard``: optionally the counts of this rank's block of samples (``sample_block(rank, world, N)``) for
    the genes of ALL ranks, [n_r x G_total] in rank order of the gene blocks: the size factors then take two
    collectives (sample_shard_protocol) instead of the 1 + 8 all-reduces of the distributed radix select."""

    def __init__(self, counts, design_matrix, *, comm, sample_shard=None, **kw):
        super().__init__(counts, design_matrix, **kw)
        self.comm = comm
        self._gene_counts = {}
        self._gathered = None
        # ranks may own different numbers of genes: the gathered vectors are padded to the largest shard
        d_g = self._pooled_once((1,), np.float64, float(self.G))
        d_all = DeviceArray(self.ctx, (comm.world,), np.float64)
        comm.allgather(d_g, d_all)
        sizes = d_all.to_host().astype(int)
        self.Gpad = int(sizes.max())
        self._d_samp = None
        if sample_shard is not None and self.size_factors_fit_type == "ratio" and self._control_mask is None:
            samp = np.asarray(sample_shard)
            n0, n1 = sample_block(comm.rank, comm.world, self.N)
            if samp.shape != (n1 - n0, int(sizes.sum())):
                raise ValueError(f"sample_shard must be {(n1 - n0, int(sizes.sum()))}, got {samp.shape}")
            padded = np.zeros((n1 - n0, comm.world * self.Gpad), dtype=np.int32)
            o = 0
            for r, gsz in enumerate(sizes):
                padded[:, r * self.Gpad: r * self.Gpad + gsz] = samp[:, o:o + gsz]
                o += gsz
            self._d_samp = DeviceArray.from_host(self.ctx, padded)
            self._samp_rows = n1 - n0
            self._samp_pad = -(-self.N // comm.world)  # ceil: every rank sends the same number of medians
            self._samp_work = None

    def _pool_reset(self):
        super()._pool_reset()
        self._gathered = None  # the gathered trend inputs lived in the recycled buffers

    def _all_genes_host(self, d_vec, n):
        """The ranks' per-gene vectors concatenated in rank order (iterative size factors: the trimmed mean of the
        dispersions and the objective's quantile / sum run over the genes of all ranks, dds.py:1460-1548).  One
        all-gather of Gpad doubles per call; the ranks' lengths are exchanged once per distinct local length."""
        W, G = self.comm.world, self.Gpad
        if n not in self._gene_counts:
            d_n = self._pooled_once((1,), np.float64, float(n))
            d_all_n = DeviceArray(self.ctx, (W,), np.float64)
            self.comm.allgather(d_n, d_all_n)
            self._gene_counts[n] = d_all_n.to_host().astype(int)
        sizes = self._gene_counts[n]
        if getattr(self, "_agh", None) is None:  # called once per objective evaluation: two persistent buffers
            self._agh = (DeviceArray(self.ctx, (G,), np.float64), DeviceArray(self.ctx, (G * W,), np.float64))
        d_send, d_all = self._agh
        self.ctx.call("dsq_d2d", _vp(d_send.ptr), _vp(d_vec.ptr), C.c_size_t(8 * n))
        self.comm.allgather(d_send, d_all)
        allv = self._down(d_all, G * W).reshape(W, G)
        return np.concatenate([allv[r, : sizes[r]] for r in range(W)])

    def _pooled_once(self, shape, dtype, value):
        arr = DeviceArray(self.ctx, shape, dtype)
        self.ctx.h2d(arr.ptr, np.full(shape, value, dtype=dtype))
        return arr

    def _size_factors(self, d_lm):
        # log means and masks are per gene (local; `control_genes` index this rank's genes); the medians over
        # the genes of all ranks come from the shared radix protocol; every rank ends with the same factors
        if self._d_samp is not None:  # two collectives: log means all-gathered, medians of the rank's own samples
            def gather(d_send):
                d_recv = self._pooled((d_send.nbytes // 8 * self.comm.world,), np.float64)
                return self.comm.allgather(d_send, d_recv)

            return sample_shard_protocol(_DeviceSampleShardOps(self, d_lm), gather)
        d_lm, d_mask = self._sf_inputs(d_lm)
        ops = _DeviceSfOps(self, d_lm, d_mask)
        if self.time_kernels:
            self.ctx.timer_start()
        d_sf = median_select_protocol(ops, self.comm.allreduce_sum)
        if self.time_kernels:
            self.kernel_log.setdefault("size_factors_dist", []).append((self.ctx.timer_stop(), self.G))
        return self._sf_finish(d_sf)

    def _gather_trend_inputs(self, Gn):
        """(raw genewise dispersion, normalised mean) of every rank on the device with ONE all-gather
        (trend_gather_protocol): two [world * Gpad] vectors (Gpad = largest shard), NaN beyond a rank's non-zero genes
        (the trend and prior kernels skip NaNs)."""
        d_gw, d_nm = self._last_gw_dev
        G, W, pipe = self.Gpad, self.comm.world, self

        class _Ops:
            def pack(self):
                d_send = pipe._pooled((2 * G,), np.float64)
                pipe.ctx.call("dsq_dev_pack2", _vp(d_gw.ptr), _vp(d_nm.ptr), int(Gn), G, _vp(d_send.ptr))
                return d_send

            def unzip(self, d_recv):
                d_a, d_b = pipe._pooled((G * W,), np.float64), pipe._pooled((G * W,), np.float64)
                pipe.ctx.call("dsq_dev_unzip2", _vp(d_recv.ptr), W, G, _vp(d_a.ptr), _vp(d_b.ptr))
                return d_a, d_b

        def gather(d_send):
            return self.comm.allgather(d_send, self._pooled((2 * G * W,), np.float64))

        self._gathered = trend_gather_protocol(_Ops(), gather)
        return self._gathered

    def _trend_prior_fused(self, Gn, d_fit):
        """The single-GPU step's fused call (trend fit, fitted values, MAD prior: one host synchronisation) on the
        gathered vectors of all ranks; this rank's own fitted values (d_fit) follow from the coefficients without
        another wait.  A step of the sharded pipeline thus has the same host synchronisations as the single-GPU one and
        three collectives (two for the size factors under the sample-block protocol, one here)."""
        d_gw_all, d_nm_all = self._gather_trend_inputs(Gn)
        n_all = self.Gpad * self.comm.world
        c2, ok, n_outer, sq = (C.c_double * 2)(), C.c_int(0), C.c_int(0), C.c_double()
        d_keep = self._dvec(n_all, np.uint8)
        d_fit_all = self._dvec(n_all)
        d_work = self._dvec(self.ctx.lib.dsq_prior_mad_work_doubles(int(n_all)))
        self.ctx.call("dsq_dev_trend_prior", _vp(d_gw_all.ptr), _vp(d_nm_all.ptr), int(n_all), C.c_double(self.min_disp),
                      C.c_double(self.max_disp), _vp(d_keep.ptr), _vp(d_fit_all.ptr), _vp(d_work.ptr), c2, C.byref(ok),
                      C.byref(n_outer), C.byref(sq))
        if not ok.value:
            return None, None  # (the gathered vectors stay for _mean_trend / _prior)
        self._gathered = None
        d_nm = self._last_gw_dev[1]
        self.ctx.call("dsq_dev_trend_eval", _vp(d_nm.ptr), int(Gn), C.c_double(c2[0]), C.c_double(c2[1]), _vp(d_fit.ptr))
        return np.array([c2[0], c2[1]]), float(sq.value)

    def _fit_trend(self, Gn):
        d_gw_all, d_nm_all = self._gathered if self._gathered is not None else self._gather_trend_inputs(Gn)
        return self._run_trend_kernel(d_gw_all, d_nm_all, self.Gpad * self.comm.world)

    def _mean_trend(self, Gn):
        if self._gathered is None:
            self._gather_trend_inputs(Gn)
        gw_all = self._down(self._gathered[0], self.Gpad * self.comm.world)
        gw_all = np.clip(gw_all[~np.isnan(gw_all)], self.min_disp, self.max_disp)
        return _trend.mean_trend(gw_all, self.min_disp)

    def _prior(self, Gn, d_fit, r):
        """MAD prior over the genes of ALL ranks (dds.py:866-884), on the gathered device vectors."""
        from scipy.special import polygamma

        d_gw_all, d_nm_all = self._gathered if self._gathered is not None else self._gather_trend_inputs(Gn)
        n_all = self.Gpad * self.comm.world
        if r.disp_function_type == "parametric":
            a0, a1 = float(r.trend_coeffs[0]), float(r.trend_coeffs[1])
        else:
            a0, a1 = float(r.mean_disp), 0.0
        d_fit_all = self._pooled((n_all,), np.float64)
        self.ctx.call("dsq_dev_trend_eval", _vp(d_nm_all.ptr), n_all, C.c_double(a0), C.c_double(a1),
                      _vp(d_fit_all.ptr))
        sq = C.c_double()
        d_work = self._pooled((self.ctx.lib.dsq_prior_mad_work_doubles(int(n_all)),), np.float64)
        self._k("prior_mad", n_all, "dsq_dev_prior_mad", _vp(d_gw_all.ptr), _vp(d_fit_all.ptr), n_all,
                C.c_double(self.min_disp), C.c_double(self.max_disp), _vp(d_work.ptr), C.byref(sq))
        self._gathered = None
        sq = float(sq.value)
        return sq, float(np.maximum(sq - polygamma(1, (self.N - self.P) / 2), 0.25))
