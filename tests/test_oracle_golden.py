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
def test_against_live_reference_taylorswift(golden):
    sys.modules.setdefault("tiktoken", types.ModuleType("tiktoken"))
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
def test_against_live_reference_random(native):
    """Differential check: the reference itself (imported from its read-only tree) against the
    oracle on random inputs -- tie-heavy alphabets, runs, multi-byte text, special tokens."""
    import random
    sys.modules.setdefault("tiktoken", types.ModuleType("tiktoken"))
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
