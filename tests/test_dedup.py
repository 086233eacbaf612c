"""CPU: chunk de-duplication.  bpe_dedup_chunks against a Python Counter, and the
claim the device path rests on -- training on the distinct chunks in first-appearance order,
weighted by multiplicity (split into powers of two), gives the merges, counts and tie-breaks of
the full chunk list -- checked with a plain-Python weighted loop against the oracle."""
import collections

import numpy as np
import pytest

import oracle
from helpers import split_chunks


def chunks_of(data, offs):
    ends = np.append(offs[1:], len(data)).astype(np.int64)
    return [data[int(a):int(b)] for a, b in zip(offs.astype(np.int64), ends)]


def expected_dedup(chunks):
    cnt = collections.Counter(chunks)
    out = []
    for c in dict.fromkeys(chunks):  # first-appearance order
        m, k = cnt[c], 0
        while m:
            if m & 1:
                out.append((c, k))
            m >>= 1
            k += 1
    return out, len(cnt)


def weighted_train(chunks, weights, num_merges):
    """minbpe's loop (regex.py:49-66) with `+ w` in place of `+ 1`; tests only."""
    ids = [list(c) for c in chunks]
    pairs, counts = [], []
    for i in range(num_merges):
        stats = {}
        for ch, w in zip(ids, weights):
            for p in zip(ch, ch[1:]):
                stats[p] = stats.get(p, 0) + w
        if not stats:
            break
        pair = max(stats, key=stats.get)
        pairs.append(pair)
        counts.append(stats[pair])
        new = []
        for ch in ids:
            out, j = [], 0
            while j < len(ch):
                if j + 1 < len(ch) and ch[j] == pair[0] and ch[j + 1] == pair[1]:
                    out.append(256 + i)
                    j += 2
                else:
                    out.append(ch[j])
                    j += 1
            new.append(out)
        ids = new
    return pairs, counts


def _texts(native):
    rng = np.random.default_rng(11)
    yield native.synth_text(40_000, 61).decode()
    # tiny alphabet: many ties, a == b runs, multiplicities with several set bits
    words = ["".join("ab"[int(x)] for x in rng.integers(0, 2, size=int(L))) for L in rng.integers(1, 7, size=40)]
    yield " ".join(words[int(i)] for i in rng.integers(0, len(words), size=3000))
    yield "aaaa aaaa aaaa aa aa aa aa aa a a a b bb bbb bbb bbb"
    yield "x"
    yield "solo"


@pytest.mark.parametrize("threads", [1, 3, 64])
def test_dedup_matches_counter(native, threads):
    for text in _texts(native):
        data, offs = split_chunks(text)
        d2, o2, w, nd = native.dedup_chunks(data, offs, threads)
        exp, distinct = expected_dedup(chunks_of(data, offs))
        assert [(c, int(k)) for c, k in zip(chunks_of(d2, o2), w)] == exp
        assert nd == distinct
        assert sum(len(c) << int(k) for c, k in zip(chunks_of(d2, o2), w)) == len(data)
    # large enough for the threaded path
    text = native.synth_text(1_500_000, 62).decode()
    data, offs = split_chunks(text)
    assert len(offs) > (1 << 16)
    d2, o2, w, nd = native.dedup_chunks(data, offs, threads)
    exp, distinct = expected_dedup(chunks_of(data, offs))
    assert [(c, int(k)) for c, k in zip(chunks_of(d2, o2), w)] == exp and nd == distinct


def test_dedup_edge_cases(native):
    d2, o2, w, nd = native.dedup_chunks(b"", np.empty(0, np.uint64))
    assert d2 == b"" and len(o2) == 0 and nd == 0
    # empty chunks (duplicate offsets) are chunks too: they carry no pairs
    d2, o2, w, nd = native.dedup_chunks(b"abab", np.array([0, 0, 2, 2], np.uint64))
    assert chunks_of(d2, o2) == [b"", b"ab"] and w.tolist() == [1, 1] and nd == 2
    with pytest.raises(RuntimeError):
        native.dedup_chunks(b"abc", np.array([2, 1], np.uint64))
    # offsets are checked by the thread that counts their range: a descent (or an overrun) far into a long list
    n = 200_000
    data = bytes(n)
    for t in (1, 8):
        bad = np.arange(n, dtype=np.uint64)
        bad[150_000] = 10
        with pytest.raises(RuntimeError):
            native.dedup_chunks(data, bad, t)
        bad = np.arange(n, dtype=np.uint64)
        bad[-1] = n + 5
        with pytest.raises(RuntimeError):
            native.dedup_chunks(data, bad, t)


def test_weighted_distinct_chunks_train_like_the_full_list(native):
    for text in _texts(native):
        data, offs = split_chunks(text)
        exp_pairs, exp_counts, _ = oracle.train(data, 80, offs, raise_on_empty=False)
        d2, o2, w, _ = native.dedup_chunks(data, offs)
        got_pairs, got_counts = weighted_train(chunks_of(d2, o2), [1 << int(k) for k in w], 80)
        assert got_pairs == exp_pairs and got_counts == exp_counts
