"""CPU: the incremental exact trainer (oracle/bpe_fast_oracle.c, `oracle.train_fast`) is pinned to the plain restatement
of the reference's loop (`oracle.train`: get_stats -> max -> merge, base.py:13-41, basic.py:31-42, regex.py:49-63) -- on
tie-heavy random streams with and without chunk boundaries, on runs of one letter (F2), on exhaustion (F6), on the
reference's own fixture text, and on committed full-length digests the plain oracle made (all 31,744 merges of an 8 MB
chunked and a 12 MB one-stream input: tails of hundreds of tied pairs).  That is what lets it make the full-length golden
of the 1 GB one-stream input (`basic1g_f`, tests/golden/gen_fast_golden.py), which the plain loop cannot finish."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from helpers import checkpoint_digests, first_divergence  # noqa: E402


def _same(data, nm, offs=None):
    a = oracle.train(data, nm, offs, raise_on_empty=False)
    b = oracle.train_fast(data, nm, offs, raise_on_empty=False)
    assert a == b, (data[:80], nm, None if offs is None else list(offs[:10]))
    return a


def test_known_answers():
    # the reference's own known-answer test (tests/test_tokenizer.py:80-107) and the tie-break examples of SURVEY F3
    assert _same(b"aaabdaaabac", 3)[0] == [(97, 97), (256, 97), (257, 98)]
    assert _same(b"ab ab ab ab", 2)[0] == [(97, 98), (256, 32)]
    assert _same(bytes([5, 6, 7, 8, 5, 6, 7, 8]), 1)[0] == [(5, 6)]
    assert _same(bytes([7, 8, 5, 6, 7, 8, 5, 6]), 1)[0] == [(7, 8)]


@pytest.mark.parametrize("seed", range(60))
def test_random_tie_heavy_streams_equal_the_plain_loop(seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(10):
        k = int(rng.integers(1, 6))
        n = int(rng.integers(0, 600))
        data = bytes(97 + rng.integers(0, k, size=n).astype(np.uint8))
        nm = int(rng.integers(1, 120))
        offs = None
        if rng.integers(0, 2) and n:
            offs = np.unique(np.concatenate([[0], rng.integers(0, n, size=int(rng.integers(0, 40)))])).astype(np.uint64)
        _same(data, nm, offs)  # (runs to exhaustion more often than not: the statuses agree too)


def test_runs_and_exhaustion():
    for n in range(0, 40):
        _same(b"a" * n, 8)
        _same(b"ab" * n, 8)
        _same(b"aab" * n + b"a" * (n % 3), 12)
    with pytest.raises(oracle.OracleEmptyStats):
        oracle.train_fast(b"ab", 2)
    with pytest.raises(oracle.OracleEmptyStats):
        oracle.train_fast(b"", 1)


def test_reference_fixture_text_256_merges():
    with open(os.path.join(ROOT, "tests", "golden", "taylorswift.txt"), "rb") as f:
        data = f.read()
    pairs, counts, lens = _same(data, 256)
    # the reference's own merges hash for this text (Basic, vocab 512)
    merges = {p: 256 + i for i, p in enumerate(pairs)}
    assert hashlib.sha256(repr(list(merges.items())).encode()).hexdigest()[:16] == "96e771b35363a8bb"
    assert lens[-1] == 78746


@pytest.mark.parametrize("name", ["full8r", "full12b"])
def test_committed_full_length_digests_of_the_plain_oracle(name):
    """all 31,744 merges of inputs the plain oracle spent 9 / 33 minutes on"""
    import minbpe_amd
    from minbpe_amd import _native
    with open(os.path.join(ROOT, "tests", "golden", "big_golden.json")) as f:
        g = json.load(f)[name]
    data = minbpe_amd.synth_text(g["bytes"], g["seed"])
    assert hashlib.sha256(data).hexdigest() == g["data_sha256"]
    offs = _native.split_offsets(data, 4) if g["chunked"] else None
    pairs, counts, lens = oracle.train_fast(data, g["done"], offs)
    assert len(pairs) == g["done"] == 31744
    assert first_divergence(checkpoint_digests(pairs, counts, lens, g["step"]), g["digests"]) is None


def test_basic1g_full_length_golden_is_pinned_to_the_plain_oracle():
    with open(os.path.join(ROOT, "tests", "golden", "big_golden.json")) as f:
        big = json.load(f)
    if "basic1g_f" not in big:
        pytest.skip("tests/golden/gen_fast_golden.py basic1g_f has not been run")
    f_, p = big["basic1g_f"], big["basic1g"]
    assert f_["fast"] and f_["done"] == 31744 and f_["data_sha256"] == p["data_sha256"] and not f_["chunked"]
    assert f_["equals_plain_oracle_first"] == p["done"] == 2048
    assert f_["first"] == p["first"]
