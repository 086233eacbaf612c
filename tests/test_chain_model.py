"""CPU model of the chained merges of a lean selection (minbpe_amd/csrc/kernels/k_lean.hip).

The claim the kernels rely on: when several pairs are tied at the maximum count M, order them by first
occurrence and keep the prefix up to the first pair that shares a token with an earlier one (or has a == b).
Then the reference (`max(stats, key=stats.get)` over a freshly built `get_stats` dict: base.py:13-22,
basic.py:35) merges exactly that prefix, one pair per iteration and in that order -- for as long as no merge
of the prefix creates a pair whose count reaches M.  Checked here against the reference semantics restated
with numpy on tie-heavy streams; the GPU tests check the kernels against the oracle."""
import random

import numpy as np
import pytest


def stats_in_order(ids):
    """(pairs in dict insertion order = order of first occurrence, their counts)"""
    keys = (ids[:-1].astype(np.int64) << 32) | ids[1:].astype(np.int64)
    uniq, first, counts = np.unique(keys, return_index=True, return_counts=True)
    order = np.argsort(first, kind="stable")
    return [(int(u >> 32), int(u & 0xFFFFFFFF)) for u in uniq[order]], counts[order]


def reference_choice(ids):
    pairs, counts = stats_in_order(ids)
    i = int(np.argmax(counts))  # first maximum in insertion order = what max(stats, key=stats.get) returns
    return pairs[i], int(counts[i])


def merge(ids, pair, idx):  # base.py:25-41
    out, i, n = [], 0, len(ids)
    a, b = pair
    while i < n:
        if ids[i] == a and i + 1 < n and ids[i + 1] == b:
            out.append(idx)
            i += 2
        else:
            out.append(int(ids[i]))
            i += 1
    return np.array(out, dtype=np.int64)


def chain_of(ids):
    """the tied pairs in order of first occurrence, cut as k_sel_lean cuts them"""
    pairs, counts = stats_in_order(ids)
    M = int(counts.max())
    tied = [p for p, c in zip(pairs, counts) if c == M]
    chain, used = [tied[0]], set(tied[0])
    for a, b in tied[1:]:
        if a == b or a in used or b in used:
            break
        chain.append((a, b))
        used.update((a, b))
    return M, chain


def created_reaches(ids_after, new_id, M):
    """does a pair involving the new token reach the tied count (k_apply_lean's chain_cut)?"""
    pairs, counts = stats_in_order(ids_after)
    return any(c >= M and new_id in p for p, c in zip(pairs, counts))


STREAMS = [("k2", 2, 4000), ("k3", 3, 6000), ("k5", 5, 9000), ("k12", 12, 20000), ("words", 0, 12000)]


@pytest.mark.parametrize("name,k,n", STREAMS)
def test_chain_is_the_references_next_merges(name, k, n):
    rng = random.Random(1234 + n)
    if k:
        ids = np.array([97 + rng.randrange(k) for _ in range(n)], dtype=np.int64)
    else:  # a few "words" repeated: rigid sequences, the usual source of ties in text
        words = [bytes(rng.randrange(97, 123) for _ in range(rng.randrange(2, 7))) for _ in range(40)]
        buf = b" ".join(rng.choice(words) for _ in range(n // 4))
        ids = np.frombuffer(buf, dtype=np.uint8).astype(np.int64)
    next_id = 256
    selections = chained = 0
    while len(ids) >= 2 and selections < 150:
        pairs, counts = stats_in_order(ids)
        if counts.max() < 2:
            break
        M, chain = chain_of(ids)
        selections += 1
        for j, pair in enumerate(chain):
            # the reference, on the stream as it stands, picks exactly the chain's next pair
            got, cnt = reference_choice(ids)
            assert got == pair and cnt == M, (name, selections, j, got, pair)
            if pair[0] == pair[1]:  # (the first pair of a selection may have a == b: the general path's merge)
                assert j == 0
            ids = merge(ids, pair, next_id)
            next_id += 1
            chained += j > 0
            if created_reaches(ids, next_id - 1, M):
                break  # chain_cut: the next iteration selects again
    assert selections > 20
    if k in (5, 12):
        assert chained > 0  # (the model exercised at least one chained merge on the less rigid streams)
