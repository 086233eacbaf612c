"""CPU: the host side of the drop-in classes end to end, with the device engine replaced by a
test double built on the oracle (tests/fake_engine.py).  What runs here is minbpe_amd/tokenizer.py
itself -- native pre-split, de-duplication, special-token splicing, batch offsets, vocab tables,
error mapping -- against the golden vectors generated from the reference.  The kernels behind the
same calls are covered by test_gpu_parity.py."""
import pytest

from helpers import case_text, toy_rank_table


@pytest.fixture()
def classes(native, monkeypatch):
    import minbpe_amd.tokenizer as T
    from fake_engine import OracleEngine
    eng = OracleEngine()
    monkeypatch.setattr(T, "engine", lambda device=None: eng)
    return T


@pytest.mark.parametrize("dedup", [True, False, "auto"])
def test_golden_cases_through_the_classes(golden, native, classes, dedup):
    for case in golden["train"]:
        cls = classes.BasicTokenizer if case["kind"] == "basic" else classes.RegexTokenizer
        tok = cls(case["pattern"]) if case.get("pattern") else cls()
        tok.dedup = dedup
        text = case_text(case, native)
        if len(text) > 300_000:
            continue  # the oracle-backed double is a CPU loop
        if case["raises_value_error"]:
            with pytest.raises(ValueError):
                tok.train(text, case["vocab_size"])
            assert tok.merges == {}
            continue
        tok.train(text, case["vocab_size"])
        assert [list(p) for p in tok.merges] == case["merges"]
        for enc in case["encode"]:
            ids = tok.encode(enc["text"])
            assert ids == enc["ids"]
            assert tok.decode(ids) == enc["text"]
            assert tok.decode_batch(ids) == enc["text"].encode("utf-8")


def test_specials_batches_and_errors(golden, classes):
    sp = golden["specials"]
    tok = classes.RegexTokenizer()
    tok.train(sp["train_text"], sp["vocab_size"])
    tok.register_special_tokens(sp["special_tokens"])
    assert tok.encode(sp["text"], allowed_special="all") == sp["ids_all"]
    assert tok.encode(sp["text"], allowed_special="none") == sp["ids_none"]
    assert tok.encode(sp["text"], allowed_special={"<|endoftext|>"}) == sp["ids_set"]
    assert tok.decode_batch(sp["ids_all"]).decode("utf-8") == sp["text"] == tok.decode(sp["ids_all"])
    with pytest.raises(ValueError, match="invalid token id: 999999"):
        tok.decode_batch([65, 999999])
    with pytest.raises(TypeError):
        tok.decode_batch([1.5])
    # documents in one batch == a loop over encode_ordinary; byte offsets of the round trip
    docs = [sp["train_text"][:200], "", "hello world", " ", sp["train_text"][200:900]]
    ids, doff = tok.encode_ordinary_batch(docs)
    assert [ids[int(a):int(b)].tolist() for a, b in zip(doff[:-1], doff[1:])] == [tok.encode_ordinary(d) for d in docs]
    raw, boff = tok.decode_batch(ids, doff)
    assert [raw[int(a):int(b)] for a, b in zip(boff[:-1], boff[1:])] == [d.encode("utf-8") for d in docs]
    b = classes.BasicTokenizer()
    b.train("aaabdaaabac", 259)
    assert b.encode("aaabdaaabac") == [258, 100, 258, 97, 99]
    with pytest.raises(KeyError):
        b.decode_batch([1, 4000])


def test_custom_pattern_takes_the_regex_module_path(classes, native):
    tok = classes.RegexTokenizer(r"\p{L}+|\p{N}+|[^\p{L}\p{N}]+")
    text = native.synth_text(20_000, 5).decode() + " one two 33 three, four!! 5"
    tok.train(text, 256 + 20)
    ids, doff = tok.encode_ordinary_batch(["one two", "", "33 three"])
    assert ids[int(doff[0]):int(doff[1])].tolist() == tok.encode_ordinary("one two")
    assert ids[int(doff[2]):int(doff[3])].tolist() == tok.encode_ordinary("33 three")
    assert tok.decode(tok.encode(text)) == text


def test_gpt4_tokenizer_from_a_rank_table(classes, native, tmp_path, monkeypatch):
    """GPT4Tokenizer end to end (gpt4.py:57-130: merge recovery, byte shuffle, ids = ranks) on a
    toy rank table -- the cl100k_base ranks themselves are not available offline."""
    import base64
    base = classes.RegexTokenizer()
    text = native.synth_text(30_000, 77).decode()
    base.train(text, 256 + 120)
    perm, ranks = toy_rank_table(base, 5)
    path = tmp_path / "toy.tiktoken"
    with open(path, "wb") as f:
        for tok, r in sorted(ranks.items(), key=lambda kv: kv[1]):
            f.write(base64.b64encode(tok) + b" " + str(r).encode() + b"\n")
    probe = text[:4000] + " don't  stop 12345 ünïcödé 😉"
    want = [perm[i] if i < 256 else i for i in base.encode_ordinary(probe)]
    for src in (ranks, str(path)):
        g = classes.GPT4Tokenizer(src)
        assert len(g.merges) == 120 and set(g.merges.values()) == set(range(256, 376))
        assert g.encode_ordinary(probe) == want
        assert g.decode(want) == probe
        assert g.decode_batch(want) == probe.encode("utf-8")
        ids = g.encode("<|endoftext|>" + probe[:200], allowed_special="all")
        assert ids[0] == 100257 and ids[1:] == g.encode_ordinary(probe[:200])
        with pytest.raises(KeyError):  # gpt4.py:89 looks ids up in vocab only: specials do not decode
            g.decode([100257])
        with pytest.raises(KeyError):
            g.decode_batch([100257])
        with pytest.raises(NotImplementedError):
            g.train("x", 300)
    with pytest.raises(ImportError):
        classes.GPT4Tokenizer()  # no tiktoken in this environment
    # ... and a `tiktoken` that is not the package (a namespace stub some other code planted) is no tiktoken either
    import sys
    import types
    monkeypatch.setitem(sys.modules, "tiktoken", types.ModuleType("tiktoken"))
    with pytest.raises(ImportError):
        classes.GPT4Tokenizer()
