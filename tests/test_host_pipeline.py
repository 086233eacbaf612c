"""CPU: the host side of the drop-in classes end to end, with the device engine replaced by a
test double built on the oracle (tests/fake_engine.py).  What runs here is minbpe_amd/tokenizer.py
itself -- native pre-split, de-duplication, special-token splicing, batch offsets, vocab tables,
error mapping -- against the golden vectors generated from the reference.  The kernels behind the
same calls are covered by test_gpu_parity.py."""
import pytest

from helpers import case_text


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
