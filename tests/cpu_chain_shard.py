"""A numpy / pure-Python restatement of the SHARDED CHAIN STEP of the library (api_rccl.hip: dp_train_loop; k_pool.hip:
k_pool_sel + k_pool_sel_dp; k_chain.hip: k_dp_fold_chain, k_apply_chain) for the CPU (gloo) tests: one rank's engine,
speaking the device's payload layouts word for word --

  table   int32[2 x 65536]       the 256 x 256 byte-pair counts as 16-bit limbs (low halves, then high halves): the SUM
                                 over the ranks cannot wrap; table_ready() joins them and refuses counts of 2^32 or more
  ckey    int64[TIE_CAP + 2]     MIN: [0] = -status, [1] = -1 if this rank cannot locate, [2 + l] = rank << 33 | first
                                 LOCAL position of the l-th pool entry to locate (canonical order), INT64_MAX = none here
  cfold   int32[2 K S + 64]      SUM: pair p's format-B vectors SL at [2p S ..), SR at [(2p + 1) S ..), tail = [2 K S ..):
                                 [p] = its adj, [16] = 1 if this rank's status is raised

-- and the pool's two-phase selection (maintain -> sort by (count, pair) -> levels to locate -> MIN -> keys -> batch).
TEST INFRASTRUCTURE, written independently of the HIP kernels (plain loops over the local stream, the pool rule
restated in tests/test_pool_model.py); never used by the product."""
import numpy as np
import torch

from cpu_shard import CpuShard, I64MAX
from test_pool_model import Pool

TIE_CAP, POS_BITS = 96, 33


class _TabView:
    """dict-like view of the replicated dense table (what Pool.maintain / rebuild read)"""

    def __init__(self, tab):
        self.tab = tab

    def get(self, p, default=0):
        return int(self.tab[p[0], p[1]])

    def items(self):
        xs, ys = np.nonzero(self.tab)
        return [((int(x), int(y)), int(self.tab[x, y])) for x, y in zip(xs, ys)]

    def values(self):
        return [c for _, c in self.items()]


class CpuChainShard(CpuShard):
    """CpuShard (the per-merge protocol: the general iterations a chain step hands back) + chain steps."""

    def __init__(self, data, offsets, kcap=8, capacity=24, depth=4):
        super().__init__(data, offsets)
        self.kcap, self.capacity, self.depth = kcap, capacity, depth

    # -- the first exchange: 16-bit limbs ------------------------------------------------------------------------------
    def begin(self, num_merges, rank, world):
        super().begin(num_merges, rank, world)
        self.world = world
        t = self.table.numpy().astype(np.int64)
        self.table = torch.from_numpy(np.concatenate([t & 0xFFFF, t >> 16]).astype(np.int32))
        self.ckey = torch.zeros(TIE_CAP + 2, dtype=torch.int64)
        self.S = (self.V + 63) // 64 * 64
        self.cfold = torch.zeros(2 * self.kcap * self.S + 64, dtype=torch.int32)
        self.pool = Pool(self.kcap, self.capacity, self.depth, None)
        self.iter = 0
        self.last_batch, self.last_z = [], []

    def table_ready(self):
        t = self.table.numpy().astype(np.int64)
        g = (t[65536:] << 16) + t[:65536]
        if int(g.max()) >> 32:
            raise OverflowError("a byte pair occurs 2^32 times or more in the whole job")
        self.tab[:256, :256] = g.reshape(256, 256)

    # -- K2, first half: the pool as far as replicated state goes; my first occurrences into the MIN payload -------------
    def sel_a(self, left):
        k = self.ckey.numpy()
        k[:] = I64MAX
        k[0], k[1] = -int(self._status != 0), 0
        self.mid = None
        if self._status != 0:
            return
        pool, view = self.pool, _TabView(self.tab)
        if pool.entries:
            pool.maintain(self.last_batch, self.last_z, view)
        self.last_batch, self.last_z = [], []
        if not pool.entries:  # rebuild: every pair at or above the depth-th level of the replicated table
            vals = sorted(set(view.values()), reverse=True)
            if not vals:
                self._status = -3  # the table is empty: max() raises in the reference (F6)
                k[0] = -1
                return
            theta = vals[min(len(vals), pool.depth) - 1]
            pool.entries = [dict(pair=p, c=c, rank=None, epoch=0) for p, c in view.items() if c >= theta]
            pool.theta = theta
            pool.trim()
            pool.rebuilds += 1
        es = sorted(pool.entries, key=lambda e: (-e["c"], e["pair"]))  # (equal counts by pair: the same order on every rank)
        levels, i = [], 0
        while i < len(es):
            j = i
            while j < len(es) and es[j]["c"] == es[i]["c"]:
                j += 1
            levels.append(es[i:j])
            i = j
        kmax = min(self.kcap, left)
        locate, seen = [], 0
        if len(es) <= pool.capacity:
            for lv in levels:
                if seen >= kmax:
                    break
                if len(lv) > 1 and (lv[0]["epoch"] == 0 or any(e["epoch"] != lv[0]["epoch"] for e in lv)):
                    if len(locate) + len(lv) > TIE_CAP:
                        break
                    locate.extend(lv)
                seen += len(lv)
        # my first occurrence of every entry to locate: rank << 33 | position in MY stream
        want = {e["pair"]: l for l, e in enumerate(locate)}
        first = {}
        for p in range(len(self.ids) - 1):
            if not self.start[p + 1]:
                pr = (self.ids[p], self.ids[p + 1])
                if pr in want and pr not in first:
                    first[pr] = p
                    if len(first) == len(want):
                        break
        for pr, l in want.items():
            if pr in first:
                k[2 + l] = (self.rank << POS_BITS) | first[pr]
        self.mid = (levels, locate, kmax)

    # -- K2, second half, after the MIN all-reduce: keys from the reduced words, the batch --------------------------------
    def sel_b(self):
        """-> list of pairs (the batch) | "defer" (the general path's merge) | None (nothing to do: a status is up)"""
        k = self.ckey.numpy()
        if int(k[0]) < 0:
            if self._status == 0:
                self._status = -7  # some rank failed: every rank stops at this merge
            return None
        levels, locate, kmax = self.mid
        pool = self.pool
        if len(pool.entries) > pool.capacity:  # one level larger than the pool: the general path decides
            pool.entries = []
            return "defer"
        if locate:
            pool.epoch += 1
            objection = int(k[1]) < 0
            for l, e in enumerate(locate):
                w = int(k[2 + l])
                e["rank"], e["epoch"] = (w, pool.epoch) if (w != I64MAX and not objection) else (None, 0)
        batch, cnts, used, seen = [], [], set(), 0
        head_same = False
        done = False
        for lv in levels:
            if seen >= kmax or done:
                break
            dirty = len(lv) > 1 and (lv[0]["epoch"] == 0 or any(e["epoch"] != lv[0]["epoch"] for e in lv))
            if dirty:
                break  # a level without an order: the walk stops before it
            for e in sorted(lv, key=lambda e: e["rank"] if len(lv) > 1 else 0):
                a, b = e["pair"]
                if a == b or a in used or b in used or len(batch) >= kmax:
                    head_same = a == b and not batch
                    done = True
                    break
                batch.append(e["pair"])
                cnts.append(e["c"])
                used.update((a, b))
            seen += len(lv)
        if not batch:
            pool.entries = []  # (a == b at the head, or a tie nobody could order: the general path's merge; the pool is void)
            return "defer"
        self._cnts = cnts
        return batch

    # -- K3 + fold: the batch's merges on MY shard, one after the other; each pair's format-B delta into the SUM payload ----
    def merge_batch(self, batch):
        f = self.cfold.numpy()
        f[:] = 0
        S, K = self.S, len(batch)
        tail = f[2 * self.kcap * S:]
        if self._status != 0:
            tail[16] = 1
            return
        z0 = 256 + self.iter
        for p, (a, b) in enumerate(batch):
            Z = z0 + p
            ids, st, wt = self.ids, self.start, self.w
            n = len(ids)
            site = [False] * n
            for q in range(n - 1):
                if ids[q] == a and ids[q + 1] == b and not st[q + 1]:
                    site[q] = True  # (a != b: occurrences cannot overlap)
            SL, SR = f[(2 * p) * S:(2 * p + 1) * S], f[(2 * p + 1) * S:(2 * p + 2) * S]
            out, ost, ow = [], [], []
            q = 0
            while q < n:
                if site[q]:
                    w = wt[q]
                    # left pair (L, a) -> (L, Z), unless a starts a chunk or L is the b of the previous site
                    if q > 0 and not st[q]:
                        if not (q >= 2 and site[q - 2]):
                            SL[ids[q - 1]] += w
                    # right pair (b, R) -> (Z, R), unless R starts a chunk / the stream ends; adj when R starts the next site
                    if q + 2 < n and not st[q + 2]:
                        if site[q + 2]:
                            tail[p] += w
                        else:
                            SR[ids[q + 2]] += w
                    out.append(Z); ost.append(st[q]); ow.append(wt[q]); q += 2
                else:
                    out.append(ids[q]); ost.append(st[q]); ow.append(wt[q]); q += 1
            self.ids, self.start, self.w = out, ost, ow
        self._batch = batch

    # -- table update from the summed payload (k_apply_chain): format B -> the four vectors of format A, pair by pair --------
    def apply_batch(self):
        f = self.cfold.numpy().astype(np.int64)
        S = self.S
        tail = f[2 * self.kcap * S:]
        if int(tail[16]) != 0 or self._status != 0:
            if self._status == 0:
                self._status = -7  # some rank's merge pass failed: nobody applies this step
            return 0
        batch, z0 = self._batch, 256 + self.iter
        V = self.V
        for p, (a, b) in enumerate(batch):
            Z = z0 + p
            SL, SR, adj = f[(2 * p) * S:(2 * p) * S + V], f[(2 * p + 1) * S:(2 * p + 1) * S + V], int(tail[p])
            self.tab[:, a] -= SL
            self.tab[:, Z] += SL
            dr, ir = SR.copy(), SR.copy()
            dr[a] += adj
            ir[Z] += adj
            self.tab[b, :] -= dr
            self.tab[Z, :] += ir
            self.tab[a, b] = 0
            assert self.tab.min() >= 0, (a, b, Z)
            self.rec[self.iter + p] = ((a, b), self._cnts[p], None, 0)
        self.last_batch, self.last_z = list(batch), [z0 + p for p in range(len(batch))]
        self.iter += len(batch)
        return len(batch)


def train_chain_sharded(shard, comm, num_merges):
    """The loop of dp_train_loop on the model: chain steps, and the general iteration (CpuShard's per-merge protocol) for
    the merges a step hands back.  Every rank runs the same control flow from replicated facts only.  -> (pairs, counts,
    summed local lengths are the caller's business), stops where the global table runs empty."""
    shard.begin(num_merges, comm.rank, comm.world)
    comm.sum_(shard.table)
    shard.table_ready()
    pairs, counts = [], []
    steps = generals = 0
    while shard.iter < num_merges:
        shard.sel_a(num_merges - shard.iter)
        comm.min_(shard.ckey)
        batch = shard.sel_b()
        if batch is None:
            break
        if batch == "defer":  # the general path's merge: the per-merge protocol, one merge
            i = shard.iter
            shard.select(i)
            comm.min_(shard.key)
            shard.merge(i)
            comm.sum_(shard.delta)
            shard.apply(i)
            pair, cnt, _len, status = shard.poll(i)
            if status != 0:
                break
            pairs.append(pair)
            counts.append(cnt)
            shard.iter += 1
            generals += 1
            continue
        shard.merge_batch(batch)
        comm.sum_(shard.cfold)
        k = shard.apply_batch()
        if k == 0:
            break
        for p in range(shard.iter - k, shard.iter):
            pairs.append(shard.rec[p][0])
            counts.append(shard.rec[p][1])
        steps += 1
    return dict(pairs=pairs, counts=counts, steps=steps, generals=generals, status=shard._status)
