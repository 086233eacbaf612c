"""A numpy restatement of the shard-engine protocol of minbpe_amd/dist.py, for
the CPU (gloo) tests of the data-parallel path.  TEST INFRASTRUCTURE: it is the
CPU stand-in for one rank's GPU, written independently of the HIP kernels
(plain Python loops over the local stream), and is never used by the product.
"""
import numpy as np
import torch

I64MAX = 0x7FFFFFFFFFFFFFFF


class CpuShard:
    def __init__(self, data: bytes, offsets, weight_exp=None, fail_at=None):
        self.fail_at = fail_at  # inject a rank-local failure into merge() of this iteration
        self.data = data
        self.offsets = offsets
        self.weight_exp = weight_exp  # per chunk: pairs inside it count 2**e times (bpe_load_bytes_weighted)
        self.device = None

    # -- protocol ---------------------------------------------------------------
    def begin(self, num_merges, rank, world):
        self.rank = rank
        self.V = 256 + num_merges
        self.ids = list(self.data)
        n = len(self.ids)
        self.start = [False] * n
        offs = [0] if self.offsets is None else [int(o) for o in self.offsets]
        for o in offs:
            if o < n:
                self.start[o] = True
        self.w = [1] * n  # weight of the chunk each token belongs to
        if self.weight_exp is not None:
            ends = offs[1:] + [n]
            for o, e, k in zip(offs, ends, self.weight_exp):
                for p in range(o, e):
                    self.w[p] = 1 << int(k)
        self.tab = np.zeros((self.V, self.V), dtype=np.int64)
        t256 = np.zeros((256, 256), dtype=np.int32)
        for p in range(n - 1):
            if not self.start[p + 1]:
                t256[self.ids[p], self.ids[p + 1]] += self.w[p]
        self.table = torch.from_numpy(t256.reshape(-1))
        self.delta = torch.zeros(4 * self.V, dtype=torch.int32)
        self.key = torch.zeros(3, dtype=torch.int64)
        self._status = 0
        self.rec = {}

    def table_ready(self):
        self.tab[:256, :256] = self.table.numpy().reshape(256, 256)

    def select(self, i):
        self.key[2] = 0
        if self._status not in (0, -3):  # a failure is sticky: later steps are no-ops
            self.key[0] = self.key[1] = I64MAX
            self.key[2] = -2
            return
        M = int(self.tab.max())
        self._count = M
        if M == 0:
            self._status = -3
            self.key[0] = self.key[1] = I64MAX
            return
        self._status = 0
        tied = set(map(tuple, np.argwhere(self.tab == M).tolist()))
        if len(tied) == 1:
            (a, b), = tied
            self.key[0], self.key[1] = a, b
            return
        w0 = w1 = I64MAX
        for p in range(len(self.ids) - 1):
            if not self.start[p + 1] and (self.ids[p], self.ids[p + 1]) in tied:
                k = ((self.rank << 32) | p) + 1
                w0, w1 = (k << 16) | self.ids[p], (k << 16) | self.ids[p + 1]
                break
        self.key[0], self.key[1] = w0, w1

    def merge(self, i):
        V, Z = self.V, 256 + i
        self.delta.zero_()
        if self._status == 0 and int(self.key[2]) < 0:
            self._status = -7  # a peer failed
        if self._status == 0 and self.fail_at == i:
            self._status = -7  # injected: e.g. a bounded device-side wait that timed out
        if self._status != 0:
            self.rec[i] = ((0, 0), 0, len(self.ids), self._status)
            return
        a, b = int(self.key[0]) & 0xFFFF, int(self.key[1]) & 0xFFFF
        self._pair = (a, b)
        ids, st, wt = self.ids, self.start, self.w
        n = len(ids)
        m = [0] * n
        for p in range(n):
            if ids[p] == a and p + 1 < n and ids[p + 1] == b and not st[p + 1] and not (p and m[p - 1]):
                m[p] = 1
        d = self.delta.numpy()
        out, ost, ow = [], [], []
        # old pairs that lose an element / new pairs that gain the new token
        before = {}
        for p in range(n - 1):
            if not st[p + 1] and (m[p] or (p and m[p - 1]) or m[p + 1] or m[p]):
                before[(ids[p], ids[p + 1])] = before.get((ids[p], ids[p + 1]), 0) + wt[p]
        p = 0
        touched = []
        while p < n:
            if m[p]:
                out.append(Z); ost.append(st[p]); ow.append(wt[p]); touched.append(True); p += 2
            else:
                out.append(ids[p]); ost.append(st[p]); ow.append(wt[p]); touched.append(False); p += 1
        after = {}
        for q in range(len(out) - 1):
            if not ost[q + 1] and (touched[q] or touched[q + 1]):
                after[(out[q], out[q + 1])] = after.get((out[q], out[q + 1]), 0) + ow[q]
        for (x, y), c in before.items():
            if (x, y) == (a, b):
                continue
            if x == b and y == a and a != b:
                # (b,a) can be a destroyed right pair (b,R=a) or left pair (L=b,a): either
                # vector maps to the same table entry; charge decR
                d[1 * V + y] += c
            elif x == b:
                d[1 * V + y] += c
            else:
                assert y == a, (x, y, a, b)
                d[0 * V + x] += c
        for (x, y), c in after.items():
            if x == Z:
                d[3 * V + y] += c
            else:
                assert y == Z
                d[2 * V + x] += c
        self.ids, self.start, self.w = out, ost, ow
        self.rec[i] = ((a, b), self._count, len(out), 0)

    def apply(self, i):
        if self._status != 0:
            return
        V, Z = self.V, 256 + i
        a, b = self._pair
        d = self.delta.numpy().astype(np.int64)
        self.tab[:, a] -= d[0 * V:1 * V]
        self.tab[b, :] -= d[1 * V:2 * V]
        self.tab[:, Z] += d[2 * V:3 * V]
        self.tab[Z, :] += d[3 * V:4 * V]
        self.tab[a, b] = 0
        assert self.tab.min() >= 0

    def poll(self, i):
        return self.rec[i]

    def end(self):
        pass
