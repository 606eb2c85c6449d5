"""numpy mirror of the device passes of the distributed size-factor median (tests only):
same order-preserving keys, same 8-bit radix passes as k_sf_* in csrc/dsq_k_stats.hip."""
import numpy as np


def f64_keys(v):
    b = np.asarray(v, dtype=np.float64).view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    return np.where(neg, ~b, b | np.uint64(1 << 63))


def key_f64(k):
    k = np.asarray(k, dtype=np.uint64)
    pos = (k >> np.uint64(63)).astype(bool)
    b = np.where(pos, k & np.uint64((1 << 63) - 1), ~k)
    return b.view(np.float64)


class NumpySfOps:
    SENT = np.uint64(0xFFFFFFFFFFFFFFFF)

    def __init__(self, counts, logmeans):
        """counts: N x G (this rank's genes); logmeans[G] (-inf for genes with a zero)."""
        N, G = counts.shape
        use = np.isfinite(logmeans)
        with np.errstate(divide="ignore", invalid="ignore"):
            ratios = np.log(counts.astype(float)) - logmeans[None, :]
        self.keys = np.where(use[None, :], f64_keys(np.where(use[None, :], ratios, 0.0)), self.SENT)
        self.N = N

    def count(self):
        return (self.keys != self.SENT).sum(1).astype(np.uint32)

    def init(self, total):
        M = total.astype(np.int64)
        self.prefix = np.zeros((2, self.N), dtype=np.uint64)
        self.rank = np.stack([np.where(M > 0, (M - 1) // 2, 0), M // 2]).astype(np.int64)

    def hist(self, shift):
        h = np.zeros((2, self.N, 256), dtype=np.uint32)
        himask = np.uint64(0) if shift == 56 else np.uint64((0xFFFFFFFFFFFFFFFF << (shift + 8)) & 0xFFFFFFFFFFFFFFFF)
        dig = ((self.keys >> np.uint64(shift)) & np.uint64(0xFF)).astype(np.int64)
        valid = self.keys != self.SENT
        for w in range(2):
            m = valid & ((self.keys & himask) == self.prefix[w][:, None])
            for n in range(self.N):
                h[w, n] = np.bincount(dig[n][m[n]], minlength=256)
        return h

    def pick(self, hist, shift):
        for w in range(2):
            for n in range(self.N):
                c = np.cumsum(hist[w, n].astype(np.int64))
                d = int(np.searchsorted(c, self.rank[w, n], side="right"))
                d = min(d, 255)
                self.rank[w, n] -= c[d - 1] if d > 0 else 0
                self.prefix[w, n] |= np.uint64(d << shift)

    def finish(self, total):
        M = total.astype(np.int64)
        v0, v1 = key_f64(self.prefix[0]), key_f64(self.prefix[1])
        med = np.where((M - 1) // 2 == M // 2, v0, (v0 + v1) / 2.0)
        return np.where(M > 0, np.exp(med), np.nan)
