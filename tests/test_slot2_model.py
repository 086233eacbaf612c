"""CPU model of the second slotted form's a != b pass (minbpe_amd/csrc/kernels/k_slots2.hip):
the per-slot rule (a slot rewrites itself from its own words + two words of left context + three
of right context taken from headers, and OWES the whole pair-table update of the sites whose `a`
it owns: delta format B), and the inverted-index candidate rule of the sparse pass.  Plain Python
on tiny slots; checked against a brute-force recount and against the plain merge.  This pins the
DESIGN the HIP kernels implement; the kernels themselves are checked on the GPU (test_gpu_*)."""
import random

import pytest

INV = None  # "no word"


def get_stats(words):
    """pairs of a flagged, weighted stream: words = [(id, flag, weight)]"""
    st = {}
    for (x, _, w), (y, fy, _) in zip(words, words[1:]):
        if not fy:
            st[(x, y)] = st.get((x, y), 0) + w
    return st


def plain_merge(words, a, b, Z):
    out, i = [], 0
    while i < len(words):
        if (i + 1 < len(words) and words[i][0] == a and words[i + 1][0] == b and not words[i + 1][1]):
            out.append((Z, words[i][1], words[i][2]))
            i += 2
        else:
            out.append(words[i])
            i += 1
    return out


def headers(slots):
    """per slot: first three words, last two words (None where missing)"""
    hs = []
    for s in slots:
        hs.append(dict(w=[s[i] if i < len(s) else INV for i in range(3)], len=len(s),
                       l0=s[-2] if len(s) >= 2 else INV, l1=s[-1] if len(s) >= 1 else INV))
    return hs


def context(hs, t):
    halo = []
    for u in range(t + 1, len(hs)):
        for i in range(min(hs[u]["len"], 3)):
            if len(halo) < 3:
                halo.append(hs[u]["w"][i])
        if len(halo) >= 3:
            break
    halo += [INV] * (3 - len(halo))
    p1 = p2 = INV
    got = 0
    for u in range(t - 1, -1, -1):
        if hs[u]["len"] == 0:
            continue
        if got == 0:
            p1 = hs[u]["l1"]
            got = 1
            if hs[u]["len"] >= 2:
                p2 = hs[u]["l0"]
                got = 2
        else:
            p2 = hs[u]["l1"]
            got = 2
        if got == 2:
            break
    return halo, p2, p1


def is_pair(x, y, a, b):
    return x is not INV and y is not INV and x[0] == a and y[0] == b and not y[1]


def slot_pass(slot, halo, p2, p1, a, b, Z, SL, SR, t=None, tprev=None, newpairs=None, tnext=None):
    """what merge_ab_tile does for one slot; returns (new slot, adj, changed, had_site).
    newpairs: receives (owner slot, pair) for every pair the pass creates (the index update)."""
    n = len(slot)
    ext = [p2, p1] + slot + halo  # index q + 2
    W = lambda q: ext[q + 2]
    carry = is_pair(p1, slot[0], a, b) if n else False
    r = [is_pair(W(q), W(q + 1), a, b) for q in range(n)]
    out, adj = [], 0
    for q in range(n):
        prev_r = carry if q == 0 else r[q - 1]
        if prev_r:
            continue  # the `b` of a site
        if r[q]:
            wa = W(q)
            out.append((Z, wa[1], wa[2]))
            wt = wa[2]
            L, LL = W(q - 1), W(q - 2)
            if not wa[1] and L is not INV:
                if not is_pair(LL, L, a, b):
                    SL[L[0]] = SL.get(L[0], 0) + wt
                    if newpairs is not None:
                        newpairs.append((t if q else tprev, (L[0], Z)))
                        if q == 0:  # a boundary pair is registered with BOTH slots it touches
                            newpairs.append((t, (L[0], Z)))
            R, RR = W(q + 2), W(q + 3)
            if R is not INV and not R[1]:
                if is_pair(R, RR, a, b):
                    adj += wt
                else:
                    SR[R[0]] = SR.get(R[0], 0) + wt
                if newpairs is not None:
                    np_ = (Z, Z if is_pair(R, RR, a, b) else R[0])
                    newpairs.append((t, np_))
                    if q + 2 >= n:  # R lives in the next slot: boundary pair
                        newpairs.append((tnext, np_))
        else:
            out.append(W(q))
    return out, adj, (carry or any(r)), any(r)


def owned_pairs(slots):
    """exact index: the pairs each slot holds -- left word in the slot -- plus, for the slot that
    starts with the right word of a boundary pair, that pair too (it is the slot that drops the
    word when the pair is merged)"""
    stream = [(t, w) for t, s in enumerate(slots) for w in s]
    own = [set() for _ in slots]
    for (t, x), (u, y) in zip(stream, stream[1:]):
        if not y[1]:
            own[t].add((x[0], y[0]))
            own[u].add((x[0], y[0]))
    return own


def random_stream(rng, n, k, pflag, wmax):
    words, w = [], 1
    for i in range(n):
        f = (i == 0) or (rng.random() < pflag)
        if f:
            w = 1 << rng.randrange(0, wmax + 1)
        words.append((rng.randrange(k), f, w))
    return words


@pytest.mark.parametrize("seed", range(300))
def test_slot_pass_equals_plain_merge_and_recount(seed):
    rng = random.Random(seed)
    tile = rng.choice([1, 2, 3, 4, 5, 8, 24, 40])
    k = rng.choice([2, 3, 4])
    words = random_stream(rng, rng.randrange(0, 120 if tile < 20 else 400), k, rng.choice([0.0, 0.1, 0.4]), rng.choice([0, 2]))
    # ragged slots, some short, some empty
    slots, i = [], 0
    while i < len(words):
        ln = rng.randrange(0 if tile < 20 else tile - 6, tile + 1)
        slots.append(words[i:i + ln])
        i += ln
    slots += [[] for _ in range(rng.randrange(0, 3))]
    next_id = k
    table = get_stats(words)
    index = owned_pairs(slots)  # exact at "build"; afterwards only additions (a Bloom filter cannot forget)
    gap = False  # sticky, like st->gap: once a short slot has been seen every slot is visited until a re-pack
    for step in range(6):
        stream = [w for s in slots for w in s]
        a, b = rng.randrange(next_id), rng.randrange(next_id)
        if a == b:
            continue  # the a == b pass is a different kernel
        Z = next_id
        hs = headers(slots)
        gap = gap or any(len(s) < 3 for s in slots[:-1])
        SL, SR, adj = {}, {}, 0
        new_slots, newpairs = [], []
        for t, s in enumerate(slots):
            halo, p2, p1 = context(hs, t)
            if not s:
                new_slots.append(s)
                continue
            tprev = max([u for u in range(t) if slots[u]], default=None)
            tnext = min([u for u in range(t + 1, len(slots)) if slots[u]], default=None)
            out, ad, changed, had = slot_pass(s, halo, p2, p1, a, b, Z, SL, SR, t, tprev, newpairs, tnext)
            # the sparse pass must not miss a slot that changes or owes an update: the filter
            # admits the pair (a slot that drops its first word holds the boundary pair too)
            cand = gap or (a, b) in index[t]
            assert cand or not changed, (seed, step, t)
            adj += ad
            new_slots.append(out)
        slots = new_slots
        for owner, pair in newpairs:
            index[owner].add(pair)
        # the index stays a superset of what each slot owns -- as long as slot numbers tell
        # neighbours (no short slots), which is when it is relied upon
        gap = gap or any(len(s) < 3 for s in slots[:-1])
        if not gap:
            for t, own in enumerate(owned_pairs(slots)):
                assert own <= index[t], (seed, step, t, own - index[t])
        got = [w for s in slots for w in s]
        assert got == plain_merge(stream, a, b, Z), (seed, step)
        # format B -> table
        for L, c in SL.items():
            table[(L, a)] = table.get((L, a), 0) - c
            table[(L, Z)] = table.get((L, Z), 0) + c
        for R, c in SR.items():
            table[(b, R)] = table.get((b, R), 0) - c
            table[(Z, R)] = table.get((Z, R), 0) + c
        if adj:
            table[(b, a)] = table.get((b, a), 0) - adj
            table[(Z, Z)] = table.get((Z, Z), 0) + adj
        table[(a, b)] = 0
        assert all(v >= 0 for v in table.values())
        assert {p: c for p, c in table.items() if c} == get_stats(got), (seed, step)
        next_id += 1


# ---------------------------------------------------------------------------
# a == b passes over a candidate list (k_merge_aa with AaArgs::cand, k_index.hip build_cand_list): format A
# charges every destroyed / created pair to its LEFT element, so besides the slots that hold a pair (a,a)
# -- a boundary pair is in the filter of both slots it touches -- the slot BEFORE each of them can owe a
# table update.  The claim: no other slot changes or owes anything.

def aa_mbits(ids, a):
    """m[k] = 1 iff a site (a,a) starts at k under the reference's left-to-right pairing (base.py:25-41)"""
    m, k = [0] * len(ids), 0
    while k + 1 < len(ids):
        if ids[k] == a and ids[k + 1] == a:
            m[k] = 1
            k += 2
        else:
            k += 1
    return m


def aa_slot_effects(ids, m, cuts, Z):
    """per slot: (changed, format-A delta entries) with every position's share charged as tile_rewrite does"""
    n = len(ids)
    owner = []
    for t, (lo, hi) in enumerate(zip(cuts, cuts[1:])):
        owner += [t] * (hi - lo)
    T = len(cuts) - 1
    changed = [False] * T
    delta = [dict() for _ in range(T)]

    def add(t, vec, tok):
        delta[t][(vec, tok)] = delta[t].get((vec, tok), 0) + 1

    for k in range(n):
        t = owner[k]
        mk, mkm1 = m[k], (m[k - 1] if k > 0 else 0)
        mkp1 = m[k + 1] if k + 1 < n else 0
        if mk or mkm1:
            changed[t] = True  # the slot holds a site's first word, or drops its second
        if k + 1 < n and not mk:  # an old pair that is not the site itself
            if mkm1:
                add(t, "decR", ids[k + 1])
            elif mkp1:
                add(t, "decL", ids[k])
        if not mkm1:  # an output element: the pair it forms with the next output element
            q = k + (2 if mk else 1)
            if q < n:
                mq = m[q]
                if mk:
                    add(t, "incR", Z if mq else ids[q])
                elif mq:
                    add(t, "incL", ids[k])
    return changed, delta


@pytest.mark.parametrize("seed", range(40))
def test_aa_pass_candidates_are_the_filter_hits_and_their_predecessors(seed):
    rng = random.Random(9000 + seed)
    k = rng.choice([2, 3, 5])
    n = rng.randrange(40, 400)
    ids = [rng.randrange(k) for _ in range(n)]
    a, Z = 0, 99
    cuts = [0]
    while cuts[-1] < n:  # slots of 3..9 words (no short slots: the kernels visit everything otherwise)
        cuts.append(min(n, cuts[-1] + rng.randrange(3, 10)))
    if cuts[-1] - cuts[-2] < 3 and len(cuts) > 2:
        cuts.pop(-2)
    T = len(cuts) - 1
    owner = []
    for t, (lo, hi) in enumerate(zip(cuts, cuts[1:])):
        owner += [t] * (hi - lo)
    # the index: slot t knows pair (x, y) at positions (p, p + 1) if it owns p -- and p + 1's slot knows it too
    cand = set()
    for p in range(n - 1):
        if ids[p] == a and ids[p + 1] == a:
            cand.add(owner[p])
            cand.add(owner[p + 1])
    visit = cand | {t - 1 for t in cand if t > 0}
    m = aa_mbits(ids, a)
    changed, delta = aa_slot_effects(ids, m, cuts, Z)
    for t in range(T):
        if t not in visit:
            assert not changed[t] and not delta[t], (seed, t, delta[t])
    # ... and the deltas of the visited slots alone are the whole table update (brute-force recount)
    before, after = {}, {}
    for x, y in zip(ids, ids[1:]):
        before[(x, y)] = before.get((x, y), 0) + 1
    out, p = [], 0
    while p < n:
        if m[p]:
            out.append(Z)
            p += 2
        else:
            out.append(ids[p])
            p += 1
    for x, y in zip(out, out[1:]):
        after[(x, y)] = after.get((x, y), 0) + 1
    got = dict(before)
    for t in visit:
        for (vec, tok), c in delta[t].items():
            pair = {"decL": (tok, a), "decR": (a, tok), "incL": (tok, Z), "incR": (Z, tok)}[vec]
            got[pair] = got.get(pair, 0) + (c if vec.startswith("inc") else -c)
    got.pop((a, a), None)  # (the merged pair itself is retired by the row scan, not by the delta vectors)
    after_wo = {p_: c for p_, c in after.items() if p_ != (a, a)}
    assert {p_: c for p_, c in got.items() if c} == after_wo, seed
