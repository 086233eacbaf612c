"""CPU: pins the oracle (oracle/bpe_oracle.c) against vectors produced by the
reference itself (tests/golden/golden.json <- gen_golden.py) and, when the
reference tree is present (build container only), against the live reference
on tests/taylorswift.txt (hashes of SURVEY.md section 8c)."""
import hashlib
import os
import sys
import types

import pytest

import oracle
from helpers import case_text, data_for


def test_train_cases(golden, native):
    for case in golden["train"]:
        data, offs = data_for(case, native)
        if "sha256" in case:
            assert hashlib.sha256(case_text(case, native).encode()).hexdigest() == case["sha256"]
        nm = case["vocab_size"] - 256
        if case["raises_value_error"]:
            with pytest.raises(oracle.OracleEmptyStats):
                oracle.train(data, nm, offs)
            continue
        pairs, counts, lens = oracle.train(data, nm, offs)
        assert [list(p) for p in pairs] == case["merges"], case["name"]


def test_encode_cases(golden, native):
    from helpers import pattern_of, split_chunks
    for case in golden["train"]:
        for enc in case.get("encode", []):
            merges = [tuple(m) for m in case["merges"]]
            if case["kind"] == "basic":
                data, offs = enc["text"].encode(), None
            else:
                data, offs = split_chunks(enc["text"], pattern_of(case))
            ids, _ = oracle.encode(merges, data, offs)
            assert ids.tolist() == enc["ids"], case["name"]


def test_primitives(golden):
    for prim in golden["primitives"]:
        ids = prim["ids"]
        st = oracle.get_stats(ids)
        assert [[a, b, c] for (a, b), c, _ in st] == prim["stats"]
        if prim["stats"]:
            # first max in dict order == reference's max(stats, key=stats.get)
            best = max(st, key=lambda e: e[1])
            assert list(best[0]) == prim["argmax"]
            assert oracle.merge(ids, prim["argmax"], 1000).tolist() == prim["merged"]
            ms = prim["merged_same"]
            assert oracle.merge(ids, ms["pair"], 1001).tolist() == ms["out"]


def test_wikipedia_known_answer():
    # reference tests/test_tokenizer.py:80-107
    pairs, _, _ = oracle.train(b"aaabdaaabac", 3)
    assert pairs == [(97, 97), (256, 97), (257, 98)]
    ids, _ = oracle.encode(pairs, b"aaabdaaabac")
    assert ids.tolist() == [258, 100, 258, 97, 99]


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_against_live_reference_taylorswift(golden, monkeypatch):
    monkeypatch.setitem(sys.modules, "tiktoken", types.ModuleType("tiktoken"))
    sys.path.insert(0, REF)
    text = open(os.path.join(REF, "tests", "taylorswift.txt"), encoding="utf-8").read()
    g = golden["taylorswift"]
    assert hashlib.sha256(text.encode()).hexdigest().startswith(g["sha256_prefix"])

    def h(obj):
        return hashlib.sha256(repr(obj).encode()).hexdigest()[:16]

    pairs, _, _ = oracle.train(text.encode(), 256)
    merges = {p: 256 + i for i, p in enumerate(pairs)}
    assert h(list(merges.items())) == g["basic512_merges_hash"]
    ids, _ = oracle.encode(pairs, text.encode())
    assert len(ids) == g["basic512_encode_len"] and h(ids.tolist()) == g["basic512_encode_hash"]

    from helpers import split_chunks
    data, offs = split_chunks(text)
    pairs, _, _ = oracle.train(data, 256, offs)
    merges = {p: 256 + i for i, p in enumerate(pairs)}
    assert h(list(merges.items())) == g["regex512_merges_hash"]
    ids, _ = oracle.encode(pairs, data, offs)
    assert len(ids) == g["regex512_encode_len"] and h(ids.tolist()) == g["regex512_encode_hash"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_against_live_reference_random(native, monkeypatch):
    """Differential check: the reference itself (imported from its read-only tree) against the
    oracle on random inputs -- tie-heavy alphabets, runs, multi-byte text, special tokens."""
    import random
    monkeypatch.setitem(sys.modules, "tiktoken", types.ModuleType("tiktoken"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from minbpe import BasicTokenizer as RefBasic, RegexTokenizer as RefRegex
    from helpers import split_chunks
    rng = random.Random(424242)
    texts = []
    for k, n in [(2, 400), (3, 900), (5, 1500), (26, 2500)]:
        texts.append("".join(rng.choice([chr(97 + rng.randrange(k)), " ", chr(97 + rng.randrange(k))]) for _ in range(n)))
    texts.append("aaaa" * 50 + " " + "ab" * 80 + "   \n\n" + "aaa " * 40)
    texts.append(native.synth_text(6000, 99).decode())
    for text in texts:
        for nm in (5, 40):
            # BasicTokenizer (basic.py:20-49, 57-74)
            ref = RefBasic()
            try:
                ref.train(text, 256 + nm)
                ref_pairs = list(ref.merges)
            except ValueError:
                ref_pairs = None
            if ref_pairs is None:
                with pytest.raises(oracle.OracleEmptyStats):
                    oracle.train(text.encode(), nm)
            else:
                pairs, _, _ = oracle.train(text.encode(), nm)
                assert pairs == ref_pairs
                probe = text[: len(text) // 3] + text[::-1][:50]
                assert oracle.encode(pairs, probe.encode())[0].tolist() == ref.encode(probe)
            # RegexTokenizer (regex.py:36-70, 92-121)
            ref = RefRegex()
            data, offs = split_chunks(text)
            try:
                ref.train(text, 256 + nm)
                ref_pairs = list(ref.merges)
            except ValueError:
                ref_pairs = None
            if ref_pairs is None:
                with pytest.raises(oracle.OracleEmptyStats):
                    oracle.train(data, nm, offs)
            else:
                pairs, _, _ = oracle.train(data, nm, offs)
                assert pairs == ref_pairs
                probe = text[: len(text) // 3]
                pd, po = split_chunks(probe)
                assert oracle.encode(pairs, pd, po)[0].tolist() == ref.encode_ordinary(probe)


# ---------------------------------------------------------------------------------------------
# The weighted form of the training loop (orc_train_weighted + orc_dedup): a checker-side shortcut that
# makes the reference's answer computable for the 1 GB headline (all 31,744 merges in minutes instead of
# days).  It must give what the plain loop gives on the un-de-duplicated list of chunks.

def _tie_heavy_texts(native):
    import random
    rng = random.Random(77)
    texts = []
    for k, n in [(2, 3000), (3, 5000), (6, 8000)]:
        words = ["".join(chr(97 + rng.randrange(k)) for _ in range(rng.randrange(1, 6))) for _ in range(30)]
        texts.append(" ".join(rng.choice(words) for _ in range(n)))
    texts.append("aaaa " * 40 + "ab ab  ab\n\n" * 30 + "aaaa" * 9 + " x")
    texts.append(native.synth_text(300_000, 31).decode())
    return texts


def test_weighted_oracle_equals_plain_oracle(native):
    import numpy as np
    from helpers import split_chunks
    for text in _tie_heavy_texts(native):
        data, offs = split_chunks(text)
        d2, o2, wts, first = oracle.dedup(data, offs)
        assert int(wts.sum()) == len(offs) and len(o2) == len(set(
            data[int(offs[i]):int(offs[i + 1]) if i + 1 < len(offs) else len(data)] for i in range(len(offs))))
        assert np.all(np.diff(first.astype(np.int64)) > 0)  # order of first appearance
        nm = 120
        plain = oracle.train(data, nm, offs, raise_on_empty=False)
        weighted = oracle.train(d2, nm, o2, raise_on_empty=False, weights=wts)
        assert plain == weighted  # pairs, counts AND the lengths of the full list


def test_oracle_dedup_equals_library_dedup(native):
    """two independent implementations of the same host step (minbpe_amd/csrc/dedup.cpp is the product's)"""
    import numpy as np
    from helpers import split_chunks
    text = native.synth_text(400_000, 32).decode()
    data, offs = split_chunks(text)
    d2, o2, wts, _ = oracle.dedup(data, offs)
    # the library emits a chunk of multiplicity w once per set bit of w (powers of two)
    ld, lo, lexp, nd = native.dedup_chunks(data, offs)
    assert nd == len(o2)
    ends = np.append(lo[1:], np.uint64(len(ld))).astype(np.int64)
    got = {}
    order = []
    for i in range(len(lo)):
        c = ld[int(lo[i]):int(ends[i])]
        if c not in got:
            got[c] = 0
            order.append(c)
        got[c] += 1 << int(lexp[i])
    oends = np.append(o2[1:], np.uint64(len(d2))).astype(np.int64)
    want = [d2[int(o2[i]):int(oends[i])] for i in range(len(o2))]
    assert order == want and [got[c] for c in order] == [int(w) for w in wts]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_weighted_oracle_against_live_reference(native, monkeypatch):
    """RegexTokenizer.train of the reference itself on the full text vs the weighted oracle on the distinct
    chunks: the same merges (regex.py:41-63)."""
    monkeypatch.setitem(sys.modules, "tiktoken", types.ModuleType("tiktoken"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from minbpe import RegexTokenizer as RefRegex
    from helpers import split_chunks
    for text in _tie_heavy_texts(native)[:4] + [native.synth_text(40_000, 33).decode()]:
        data, offs = split_chunks(text)
        d2, o2, wts, _ = oracle.dedup(data, offs)
        for nm in (8, 60):
            ref = RefRegex()
            try:
                ref.train(text, 256 + nm)
                ref_pairs = list(ref.merges)
            except ValueError:
                ref_pairs = None
            if ref_pairs is None:
                with pytest.raises(oracle.OracleEmptyStats):
                    oracle.train(d2, nm, o2, weights=wts)
            else:
                assert oracle.train(d2, nm, o2, weights=wts)[0] == ref_pairs


def test_big_golden_weighted_entries_continue_the_plain_ones():
    """big_golden.json: the digest after k merges is a hash of the first k (pair, count, length) rows, so the
    weighted entries (all merges) must show the plain oracle's digest wherever both have a checkpoint."""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_golden.json")) as f:
        big = json.load(f)
    for wname, pname in (("regex1g_w", "regex1g"), ("cfg3s_w", "cfg3s")):
        w, p = big[wname], big[pname]
        assert w["weighted"] and w["data_sha256"] == p["data_sha256"] and w["offsets_sha256"] == p["offsets_sha256"]
        pd = dict(map(tuple, p["digests"]))
        common = [(k, d) for k, d in w["digests"] if k in pd]
        assert common and all(pd[k] == d for k, d in common)
        assert w["equals_plain_oracle_first"] == p["done"] and w["done"] == w["merges"]
    assert big["regex1g_w"]["done"] == 31744


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_oracle_encode_with_a_cl100k_sized_table_against_live_reference(native, monkeypatch):
    """oracle.encode with 100,000 ranks -- ids 256 + rank up to 100,255, and a merges dict whose values are
    not consecutive (what GPT4Tokenizer's are) -- against the reference's own _encode_chunk
    (regex.py:92-121) given the same merges dict."""
    import numpy as np
    monkeypatch.setitem(sys.modules, "tiktoken", types.ModuleType("tiktoken"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from minbpe import RegexTokenizer as RefRegex
    from helpers import cl100k_shaped_table, split_chunks
    text = native.synth_text(400_000, 51).decode()
    data, offs = split_chunks(text)
    d2, o2, wts, _ = oracle.dedup(data, offs)
    base = oracle.train(d2, 3000, o2, weights=wts)[0]
    probe = native.synth_text(30_000, 52).decode() + " don't  stop 12345 ünïcödé 😉"
    pd, po = split_chunks(probe)
    for kind in ("rank", "sparse"):
        pairs, mids = cl100k_shaped_table(base, 100_000, 5, kind)
        ref = RefRegex()
        vals = mids if mids is not None else 256 + np.arange(len(pairs))
        ref.merges = {(int(a), int(b)): int(v) for (a, b), v in zip(pairs, vals)}
        assert len(ref.merges) == len(pairs)  # (the table has no repeated pair)
        want = ref.encode_ordinary(probe)
        got, _ = oracle.encode(pairs, pd, po, merge_ids=mids)
        assert got.tolist() == want
        assert max(want) >= 65536 and len(set(want)) > 500


def test_pure_python_restatement_against_the_reference_fixtures(golden, native):
    """oracle/pyref.py (what bench.py times as minbpe's pure-Python path on the GPU host) against the vectors the
    reference generated: primitives (get_stats order and counts, max()'s tie-break, merge incl. a == a runs), every
    BasicTokenizer training case, and the reference's own hash of the 256 merges of tests/taylorswift.txt."""
    from oracle import pyref
    for prim in golden["primitives"]:
        ids = prim["ids"]
        st = pyref.get_stats(ids)
        assert [[a, b, c] for (a, b), c in st.items()] == prim["stats"]
        if prim["stats"]:
            assert list(max(st, key=st.get)) == prim["argmax"]
            assert pyref.merge(ids, tuple(prim["argmax"]), 1000) == prim["merged"]
            assert pyref.merge(ids, tuple(prim["merged_same"]["pair"]), 1001) == prim["merged_same"]["out"]
    n = 0
    for case in golden["train"]:
        if case["kind"] != "basic":
            continue
        data, _ = data_for(case, native)
        nm = case["vocab_size"] - 256
        if case["raises_value_error"]:
            with pytest.raises(ValueError):
                pyref.train(data, nm)
            continue
        if len(data) * nm > 30_000_000:  # (keep the CPU suite short: the C oracle covers the long cases)
            continue
        pairs, counts = pyref.train(data, nm)
        assert [list(p) for p in pairs] == case["merges"], case["name"]
        n += 1
    assert n >= 3
    # ... and equal to the C oracle on a text with ties, counts included
    data = native.synth_text(30_000, 9)
    pairs, counts = pyref.train(data, 40)
    op, oc, _ = oracle.train(data, 40)
    assert [tuple(p) for p in pairs] == [tuple(p) for p in op] and list(counts) == list(oc)
    # the reference's own taylorswift.txt answer, as far as a few seconds of Python reach: the first 24 merges
    p = os.path.join(os.path.dirname(__file__), "golden", "taylorswift.txt")
    text = open(p, encoding="utf-8").read().encode()
    pairs, _ = pyref.train(text, 24)
    assert [tuple(p) for p in pairs] == [tuple(p) for p in oracle.train(text, 24)[0]]
