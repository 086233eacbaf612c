"""CPU model of the NEXT batch rule of the chain steps (not in the engine yet): a batch that goes on
below the maximum INTO TIED LEVELS.

Today's rule (k_chain.hip, tests/test_list_model.py): the pairs tied at the maximum M in order of first occurrence,
the longest prefix with a != b and no shared token; if that is the whole list, the levels below while each holds
exactly one pair.  The rule pinned here drops the "exactly one":

    walk the levels of the current table from the top; inside a level take the pairs in order of first occurrence;
    stop for good at the first pair that has a == b or shares a token with a pair already taken (or at the cap);
    a level is left only when ALL of its pairs were taken.

Why it is exact (base.py:13-41, basic.py:31-42, regex.py:49-63): the pairs taken share no token, so their sites do
not overlap and no merge of the batch changes another batch pair's count.  A merge (a, b) -> Z lowers only pairs
(L, a) and (b, R) and creates only pairs with Z, each of which inherits at most the count of the (L, a) / (b, R) it
comes from.  When the walk stands at level c every pair above c is in the batch, so everything else counts at most
c: a created pair reaches c only by taking over EVERY occurrence of a level-c pair that shares a token with the
batch -- and then stands exactly where that pair stood in the order of first occurrence -- and the walk stops at
that place at the latest (shared token).  Everything before that place is untouched by the batch: same counts,
same relative order.  So the reference, merging one pair at a time, takes the same pairs in the same order.

Checked against the reference semantics restated with numpy (a fresh get_stats dict and max() per merge), on
tie-heavy streams, chunked and not, with caps 2..16.  tools/level_model.py replays the 1 GB headline run under this
family of rules (profiles/r4_level_model.json)."""
import random

import numpy as np
import pytest

from test_list_model import merge, stats_in_order


def batch_by_levels(chunks, cap):
    """the batch the rule above takes from the current state: [(a, b), ...] (a lone a == b pair is a batch of one:
    the general path merges it)"""
    pairs, counts = stats_in_order(chunks)
    if not pairs:
        return []
    batch, used = [], set()
    for c in sorted(set(int(x) for x in counts), reverse=True):
        for (a, b), n in zip(pairs, counts):  # dict order = order of first occurrence
            if int(n) != c:
                continue
            if a == b or a in used or b in used:
                return batch if batch else [(a, b)]
            batch.append((a, b))
            used |= {a, b}
            if len(batch) == cap:
                return batch
    return batch


def reference_next(chunks, k, next_id):
    """the next k merges of the reference loop from this state (its pairs, and the chunks after them)"""
    out = []
    for j in range(k):
        pairs, counts = stats_in_order(chunks)
        if not pairs:
            break
        best = pairs[int(np.argmax(counts))]  # first maximum in dict order = max(stats, key=stats.get)
        out.append(best)
        chunks = [merge(c, best, next_id + j) for c in chunks]
    return out, chunks


def streams():
    rng = random.Random(1234)
    for trial in range(120):
        alpha = rng.choice([2, 3, 3, 4, 5, 8])
        n = rng.choice([12, 30, 60, 150, 400])
        if trial % 3 == 0:  # one stream (BasicTokenizer)
            yield [np.array([rng.randrange(alpha) for _ in range(n)], dtype=np.int64)]
        else:  # chunks sharing one table (RegexTokenizer); repeated chunks make whole levels tie
            words = [np.array([rng.randrange(alpha) for _ in range(rng.randint(1, 7))], dtype=np.int64)
                     for _ in range(rng.randint(2, 12))]
            yield [words[rng.randrange(len(words))] for _ in range(max(2, n // 4))]


@pytest.mark.parametrize("cap", [2, 4, 8, 16])
def test_batches_through_tied_levels_are_the_reference_merges(cap):
    deep = 0
    for chunks in streams():
        next_id = 100
        for _ in range(40):
            batch = batch_by_levels(chunks, cap)
            if not batch:
                break
            want, after = reference_next(chunks, len(batch), next_id)
            assert want == batch, (cap, batch, want)
            pairs, counts = stats_in_order(chunks)
            table = dict(zip(pairs, (int(c) for c in counts)))
            if len({table[p] for p in batch}) > 1 and any(
                    sum(1 for q in pairs if table[q] == table[p]) > 1 for p in batch[1:] if table[p] < table[batch[0]]):
                deep += 1  # a batch that entered a level below the maximum with several pairs in it
            chunks, next_id = after, next_id + len(batch)
    assert deep > 50, deep  # the cases this rule is about do occur


def test_a_created_pair_takes_the_place_of_the_pair_it_consumes():
    # (1, 2) x3 at the top; level 2 holds (0, 1), (3, 4), (5, 6) in this order of first occurrence.  (0, 1) shares a token
    # with (1, 2): the walk stops there -- and rightly: after the merge (0, Z) has count 2 and stands in (0, 1)'s place,
    # before (3, 4)
    ids = np.array([0, 1, 2, 9, 3, 4, 9, 5, 6, 8, 0, 1, 2, 7, 3, 4, 7, 5, 6, 8, 1, 2], dtype=np.int64)
    got = batch_by_levels([ids], 8)
    assert got == [(1, 2)]
    want, after = reference_next([ids], 4, 100)
    assert want == [(1, 2), (0, 100), (3, 4), (5, 6)]
    # from the state after that merge the walk takes the whole tied level at once
    _, one = reference_next([ids], 1, 100)
    assert batch_by_levels(one, 8)[:3] == [(0, 100), (3, 4), (5, 6)]


def batch_by_levels_shared_seconds(chunks, cap):
    """round 6's rule (k_pool.hip, pool_finish): as batch_by_levels, but only a pair that could CHAIN onto a site of the
    batch stops the walk -- (x, y) with x a second token or y a first token of a pair taken (and a == b) -- or one whose
    FIRST token is taken already (the merge pass looks a pair up by its first token).  A second token may be shared: sites
    of (a, b) and (c, b) never overlap, neither merge changes the other's count, and the pairs a merge lowers or creates
    are (L, a) / (b, R) and their heirs, which chain onto the batch and stop the walk where they stand."""
    pairs, counts = stats_in_order(chunks)
    if not pairs:
        return []
    batch, firsts, seconds = [], set(), set()
    for c in sorted(set(int(x) for x in counts), reverse=True):
        for (a, b), n in zip(pairs, counts):
            if int(n) != c:
                continue
            if a == b or a in firsts or a in seconds or b in firsts:
                return batch if batch else [(a, b)]
            batch.append((a, b))
            firsts.add(a)
            seconds.add(b)
            if len(batch) == cap:
                return batch
    return batch


@pytest.mark.parametrize("cap", [2, 4, 8, 15])
def test_batches_that_share_second_tokens_are_the_reference_merges(cap):
    shared = 0
    for chunks in streams():
        next_id = 100
        for _ in range(40):
            batch = batch_by_levels_shared_seconds(chunks, cap)
            if not batch:
                break
            want, after = reference_next(chunks, len(batch), next_id)
            assert want == batch, (cap, batch, want)
            shared += len({b for _, b in batch}) < len(batch)
            chunks, next_id = after, next_id + len(batch)
    assert shared > 30, shared  # the cases this rule is about do occur
