"""CPU: host-side logic of the drop-in classes that needs no GPU: the .model /
.vocab formats (base.py:97-165), decode (basic.py:51-55, regex.py:78-90),
vocab construction, chunk concatenation."""

import pytest

from minbpe_amd import BasicTokenizer, RegexTokenizer, Tokenizer
from minbpe_amd.tokenizer import _concat_chunks, render_token, GPT4_SPLIT_PATTERN


def test_model_file_format_matches_reference(golden, tmp_path):
    tok = RegexTokenizer()
    tok.merges = {(97, 98): 256, (32, 256): 257}
    tok.vocab = tok._build_vocab()
    tok.register_special_tokens({"<|endoftext|>": 1000})
    prefix = str(tmp_path / "tok")
    tok.save(prefix)
    assert open(prefix + ".model", encoding="utf-8").read() == golden["model_file"]["text"]
    assert open(prefix + ".vocab", encoding="utf-8").read() == golden["model_file"]["vocab"]
    t2 = RegexTokenizer()
    t2.load(prefix + ".model")
    assert t2.merges == tok.merges and list(t2.merges) == list(tok.merges)
    assert t2.special_tokens == {"<|endoftext|>": 1000}
    assert t2.pattern == GPT4_SPLIT_PATTERN
    assert t2.vocab[257] == b" ab" and t2.vocab[1000] == b"<|endoftext|>"
    # like the reference, load() does not rebuild inverse_special_tokens, yet decode works
    assert t2.decode([1000, 257]) == "<|endoftext|> ab"


def test_load_asserts(tmp_path):
    t = Tokenizer()
    with pytest.raises(AssertionError):
        t.load(str(tmp_path / "x.txt"))
    p = tmp_path / "bad.model"
    p.write_text("minbpe v2\n\n0\n")
    with pytest.raises(AssertionError):
        t.load(str(p))


def test_decode_paths():
    b = BasicTokenizer()
    assert b.decode([104, 105]) == "hi"
    assert b.decode([0xff]) == "�"  # errors="replace"
    with pytest.raises(KeyError):
        b.decode([999])
    r = RegexTokenizer()
    with pytest.raises(ValueError, match="invalid token id: 999"):
        r.decode([999])
    r.register_special_tokens({"<|x|>": 999})
    assert r.decode([104, 999]) == "h<|x|>"


def test_base_class_is_abstract():
    t = Tokenizer()
    for call in (lambda: t.train("a", 256), lambda: t.encode("a"), lambda: t.decode([1])):
        with pytest.raises(NotImplementedError):
            call()
    assert len(t.vocab) == 256 and t.pattern == "" and t.merges == {}


def test_train_asserts_vocab_size_before_touching_the_gpu():
    with pytest.raises(AssertionError):
        BasicTokenizer().train("abc", 255)
    with pytest.raises(AssertionError):
        RegexTokenizer().train("abc", 10)


def test_encode_allowed_special_validation():
    r = RegexTokenizer()
    r.register_special_tokens({"<|x|>": 999})
    with pytest.raises(ValueError, match="not understood"):
        r.encode("abc", allowed_special="bogus")
    with pytest.raises(AssertionError):
        r.encode("a<|x|>b")  # none_raise


def test_concat_chunks_drops_empty():
    data, offs = _concat_chunks([b"ab", b"", b"c", b"def"])
    assert data == b"abcdef" and offs.tolist() == [0, 2, 3]
    data, offs = _concat_chunks([])
    assert data == b"" and len(offs) == 0


def test_render_token():
    assert render_token(b"a\nb") == "a\\u000ab"
    assert render_token(b"\xff") == "�"


def test_gpt4_merge_recovery_from_ranks(native):
    """GPT4Tokenizer needs tiktoken's cl100k ranks, which are not available offline; the part
    that is ours -- rebuilding the merge pairs from a {token bytes: rank} table (gpt4.py:29-46)
    -- is checked on a rank table made from merges we trained ourselves."""
    import oracle
    from minbpe_amd.tokenizer import _recover_merges
    text = native.synth_text(30_000, 5)
    pairs, _, _ = oracle.train(text, 120)
    vocab = {i: bytes([i]) for i in range(256)}
    for i, (a, b) in enumerate(pairs):
        vocab[256 + i] = vocab[a] + vocab[b]
    if len(set(vocab.values())) != len(vocab):  # a rank table needs distinct byte strings
        pytest.skip("two merges produced the same byte string")
    ranks = {tok: idx for idx, tok in vocab.items()}
    rec = _recover_merges(ranks)
    assert rec == {p: 256 + i for i, p in enumerate(pairs)}


def test_reference_import_paths():
    # the reference's module layout (minbpe/base.py, basic.py, regex.py, gpt4.py) resolves here too
    import minbpe_amd
    from minbpe_amd.base import Tokenizer as T1, get_stats, merge, render_token  # noqa: F401
    from minbpe_amd.basic import BasicTokenizer as B1
    from minbpe_amd.regex import GPT2_SPLIT_PATTERN, GPT4_SPLIT_PATTERN as P4, RegexTokenizer as R1  # noqa: F401
    from minbpe_amd.gpt4 import GPT4_SPECIAL_TOKENS, GPT4Tokenizer as G1, bpe, recover_merges
    assert (T1, B1, R1, G1) == (minbpe_amd.Tokenizer, minbpe_amd.BasicTokenizer, minbpe_amd.RegexTokenizer,
                                minbpe_amd.GPT4Tokenizer)
    assert P4 == minbpe_amd.GPT4_SPLIT_PATTERN and GPT4_SPECIAL_TOKENS["<|endoftext|>"] == 100257
    # gpt4.py:11-46 on a toy rank table: "ab" merged first, then "abc"
    ranks = {bytes([i]): i for i in range(256)}
    ranks[b"ab"] = 256
    ranks[b"abc"] = 257
    assert bpe(ranks, b"abc", max_rank=257) == [b"ab", b"c"]
    assert recover_merges(ranks) == {(97, 98): 256, (256, 99): 257}


def test_utf8_encode_by_all_threads_equals_str_encode():
    """_native.utf8_encode (bpe_utf8_encode: a CPython str's code points transcoded by segments in parallel) gives the
    bytes of text.encode("utf-8") for strings of every storage kind -- Latin-1, BMP, beyond the BMP --, goes to str.encode
    for short, ASCII or otherwise unfit strings, and leaves a lone surrogate to str.encode's own exception (what the
    reference raises, basic.py:25)."""
    import numpy as np
    from minbpe_amd import _native
    rng = np.random.default_rng(5)
    n = (1 << 22) + 12345
    pools = {1: [0x41, 0x7A, 0xE9, 0xFF, 0x20], 2: [0x41, 0xE9, 0x20AC, 0x4E2D, 0xFFFD, 0x7FF, 0x800],
             4: [0x41, 0xE9, 0x20AC, 0x1F600, 0x10FFFF, 0x10000, 0xFFFF]}
    for kind, pool in pools.items():
        text = "".join(map(chr, rng.choice(pool, n)))
        got = _native.utf8_encode(text)
        assert isinstance(got, np.ndarray), kind  # (the native path ran)
        assert bytes(got) == text.encode("utf-8"), kind
        assert bytes(_native.utf8_encode(text, threads=3)) == text.encode("utf-8")
    assert _native.utf8_encode("abc") == b"abc" and _native.utf8_encode("") == b""
    assert _native.utf8_encode("a" * n) == b"a" * n  # (ASCII: str.encode is a copy already)
    with pytest.raises(UnicodeEncodeError):
        _native.utf8_encode("ab\ud800" * (n // 3))
