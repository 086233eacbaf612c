"""CPU model of the POOL selection of the chain steps (minbpe_amd/csrc/kernels/k_pool.hip): the list of
pairs tied at the maximum (tests/test_list_model.py) generalised to EVERY pair whose count is at least a threshold
theta, so that a step's batch walks the levels freely (tests/test_level_model.py pins that batch rule) and a full
selection is needed only when the pool runs empty.

State: the pool = every pair with count >= theta (INVARIANT, asserted here against the true table after every step),
each entry with its count and an order key (rank, epoch): ranks of one epoch are first-occurrence positions taken at
one instant, and stay a valid relative order while the entries are untouched.
  step      sort by count; the levels the walk may reach (up to the cap + the rest of the last level) that hold more
            than one entry and are not CLEAN (all entries of one epoch) are LOCATED afresh: every entry of such a level
            gets its true first-occurrence position and a new epoch.  The batch = the longest prefix of (count
            descending, rank) with a != b in which no pair could chain onto a site of another -- (x, y) with x a second
            token or y a first token of a pair before it -- and no FIRST token comes twice; a second token may be shared
            (round 6; tests/test_level_model.py pins that rule on its own), at most `cap` pairs.  An a == b pair at the
            head is the general path's merge (the pool is void afterwards).
  maintain  after the batch: an entry (x, y) with x the SECOND token of batch pairs p (-> Z_p; several p may end in x) or y
            the FIRST token of one (-> Zy) has its occurrences spread over (x, y), (Z_p, y), (x, Zy), (Z_p, Zy): each of
            those whose count in the updated table is >= theta is an entry; the one whose count EQUALS the entry's old count took over
            every occurrence and inherits its order key, the others have no order (epoch 0).  Every other entry is
            untouched.  No pair outside the pool can reach theta: a created pair (L, Z) / (Z, R) / (Zi, Zj) counts at
            most what (L, a) / (b, R) / (bi, aj) counted before, and that pair was in the pool if it counted >= theta.
  rebuild   pool empty: a full selection gathers every pair >= a new theta.
Checked against the reference semantics (a fresh get_stats dict and max() per merge: base.py:13-22, basic.py:35,
regex.py:51-56) on tie-heavy streams, chunked and not, caps 2..16, pool capacities small enough to overflow."""
import random

import numpy as np
import pytest

from test_list_model import make_stream, merge, stats_in_order, table_of, STREAMS


class Pool:
    def __init__(self, cap, capacity, depth, rng):
        self.cap, self.capacity, self.depth, self.rng = cap, capacity, depth, rng
        self.entries = []   # dicts: pair, c, rank, epoch
        self.theta = None
        self.epoch = 0
        self.rebuilds = self.locates = self.inherited = 0

    def rebuild(self, chunks):
        table = table_of(chunks)
        levels = sorted(set(table.values()), reverse=True)
        theta = levels[min(len(levels), self.depth) - 1]
        self.entries = [dict(pair=p, c=c, rank=None, epoch=0) for p, c in table.items() if c >= theta]
        self.theta = theta
        self.trim()
        self.rebuilds += 1

    def trim(self):
        """more entries than the pool holds: whole levels leave from the bottom, theta rises above them"""
        while len(self.entries) > self.capacity:
            low = min(e["c"] for e in self.entries)
            if all(e["c"] == low for e in self.entries):
                break  # (one level larger than the pool: the device defers to the general path)
            self.entries = [e for e in self.entries if e["c"] > low]
            self.theta = low + 1

    def locate(self, chunks, level):
        pairs, _ = stats_in_order(chunks)
        order = {p: i for i, p in enumerate(pairs)}
        self.epoch += 1
        self.locates += 1
        for e in level:
            e["rank"], e["epoch"] = order[e["pair"]], self.epoch

    def batch(self, chunks, left):
        es = sorted(self.entries, key=lambda e: -e["c"])
        # levels in count order; locate the ones the walk may reach
        levels, i = [], 0
        while i < len(es):
            j = i
            while j < len(es) and es[j]["c"] == es[i]["c"]:
                j += 1
            levels.append(es[i:j])
            i = j
        kmax = min(self.cap, left)
        seen = 0
        for lv in levels:
            if seen >= kmax:
                break
            if len(lv) > 1 and (lv[0]["epoch"] == 0 or any(e["epoch"] != lv[0]["epoch"] for e in lv)):
                self.locate(chunks, lv)
            seen += len(lv)
        batch, cnts, firsts, seconds = [], [], set(), set()
        seen = 0
        for lv in levels:
            if seen >= kmax:
                break
            for e in sorted(lv, key=lambda e: e["rank"] if len(lv) > 1 else 0):
                a, b = e["pair"]
                if a == b or a in firsts or a in seconds or b in firsts or len(batch) >= kmax:
                    return batch, cnts, (e["pair"] if not batch else None)
                batch.append(e["pair"])
                cnts.append(e["c"])
                firsts.add(a)
                seconds.add(b)
            seen += len(lv)
        return batch, cnts, None

    def maintain(self, batch, znew, table):
        ends = {}
        for (a, b), z in zip(batch, znew):
            ends.setdefault(b, []).append(z)  # (several pairs of a batch may end in b)
        starts = {a: z for (a, b), z in zip(batch, znew)}
        assert len(starts) == len(batch)  # (first tokens are distinct)
        taken = set(batch)
        out = []
        for e in self.entries:
            x, y = e["pair"]
            if e["pair"] in taken:
                continue
            if x not in ends and y not in starts:
                assert table.get(e["pair"], 0) == e["c"]  # untouched
                out.append(e)
                continue
            lefts = [x] + ends.get(x, [])
            rights = [y] + ([starts[y]] if y in starts else [])
            cands = [(l, r) for l in lefts for r in rights]
            assert sum(table.get(p, 0) for p in cands) <= e["c"]
            for p in cands:
                c = table.get(p, 0)
                if c < self.theta:
                    continue
                if c == e["c"]:  # took over every occurrence: stands where the entry stood
                    out.append(dict(pair=p, c=c, rank=e["rank"], epoch=e["epoch"]))
                    self.inherited += p != e["pair"]
                else:
                    out.append(dict(pair=p, c=c, rank=None, epoch=0))
        self.entries = out
        self.trim()


CASES = [(name, k, n, cap, capacity, depth)
         for (name, k, n) in STREAMS
         for (cap, capacity, depth) in [(8, 96, 6), (16, 24, 4), (2, 8, 3), (16, 96, 40)]]


@pytest.mark.parametrize("name,k,n,cap,capacity,depth", CASES)
@pytest.mark.parametrize("seed", [11, 12])
def test_pool_steps_are_the_references_merges(name, k, n, cap, capacity, depth, seed):
    chunks = make_stream(name, k, n // 2, 313 * seed + n + cap)
    rng = random.Random(seed)
    pool = Pool(cap, capacity, depth, rng)
    total, next_id, done, steps, multi, shared = 160, 256, 0, 0, 0, 0
    while done < total:
        pairs, counts = stats_in_order(chunks)
        if not len(counts) or counts.max() < 2:
            break
        if not pool.entries:
            pool.rebuild(chunks)
        if len(pool.entries) > pool.capacity:  # one level larger than the pool: the general path's merge
            best = pairs[int(np.argmax(counts))]
            chunks = [merge(c, best, next_id) for c in chunks]
            next_id, done = next_id + 1, done + 1
            pool.entries = []
            continue
        batch, cnts, head_same = pool.batch(chunks, total - done)
        if not batch:  # a == b at the head: the general path's merge, the pool is void
            assert head_same is not None and head_same[0] == head_same[1]
            assert pairs[int(np.argmax(counts))] == head_same
            chunks = [merge(c, head_same, next_id) for c in chunks]
            next_id, done = next_id + 1, done + 1
            pool.entries = []
            continue
        znew = list(range(next_id, next_id + len(batch)))
        for pair, cnt, z in zip(batch, cnts, znew):
            ref_pairs, ref_counts = stats_in_order(chunks)
            j = int(np.argmax(ref_counts))
            assert ref_pairs[j] == pair and int(ref_counts[j]) == cnt, (name, seed, done, pair, cnt, ref_pairs[j])
            chunks = [merge(c, pair, z) for c in chunks]
        next_id += len(batch)
        done += len(batch)
        steps += 1
        multi += len(set(cnts)) > 1
        shared += len({b for _, b in batch}) < len(batch)
        table = table_of(chunks)
        pool.maintain(batch, znew, table)
        # INVARIANT: the pool is exactly the pairs that count theta or more
        want = {p: c for p, c in table.items() if c >= pool.theta}
        have = {e["pair"]: e["c"] for e in pool.entries}
        assert have == want, (name, seed, done, sorted(set(want) ^ set(have))[:4])
    assert done > 30
    if name in ("words", "chunks") and depth >= 6:
        assert pool.rebuilds < steps  # (most steps take their pairs off the pool)
    if name in ("k12", "words", "chunks") and cap >= 8 and depth >= 4:
        assert multi > 0
    SHARED[0] += shared


SHARED = [0]


def test_batches_with_shared_second_tokens_did_occur():
    # (runs after the cases above in file order: the rule the model pins is exercised, not only permitted)
    if SHARED[0] == 0:
        pytest.skip("run together with test_pool_steps_are_the_references_merges")
    assert SHARED[0] > 20, SHARED[0]
