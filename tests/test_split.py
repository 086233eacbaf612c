"""CPU: the native pre-split (bpe_split, minbpe_amd/csrc/split.cpp) against
`regex.findall` with the reference's two patterns (minbpe/regex.py:18-19), on
adversarial strings, a tricky-code-point fuzz, and longer mixed text."""
import random

import numpy as np
import pytest
import regex as re

from minbpe_amd.tokenizer import GPT2_SPLIT_PATTERN, GPT4_SPLIT_PATTERN

PATS = {2: re.compile(GPT2_SPLIT_PATTERN), 4: re.compile(GPT4_SPLIT_PATTERN)}

TRICKY = [" ", "  ", "\t", "\n", "\r", "\r\n", "\x0b", "\x0c", "\x1c", "\x1f", "\x85", "\xa0", " ", " ",
          " ", " ", "　", "​", "﻿",
          "a", "Z", "é", "ß", "ſ", "K", "K", "中", "א", "ǅ", "ª", "ᾳ",
          "0", "7", "٣", "²", "½", "Ⅷ", "〇", "𝟗",
          "!", "'", "’", "\"", "-", "_", "€", "$", "…", "́", "⃣", "😉", "🇦", "\x00", "\x7f",
          "'s", "'S", "'ſ", "'t", "'T", "'ll", "'LL", "'lL", "'ve", "'Ve", "'re", "'RE", "'d", "'m", "'x", "''",
          "123", "1234", "12345678", "3.14", "x1y2"]


def ref_offsets(which, text):
    chunks = [c.encode("utf-8") for c in re.findall(PATS[which], text)]
    assert b"".join(chunks) == text.encode("utf-8")  # both patterns match every character
    chunks = [c for c in chunks if c]
    return np.cumsum([0] + [len(c) for c in chunks[:-1]], dtype=np.uint64) if chunks else np.empty(0, np.uint64)


def check(native, text, threads=1):
    data = text.encode("utf-8")
    for which in (2, 4):
        got = native.split_offsets(data, which, threads)
        exp = ref_offsets(which, text)
        assert np.array_equal(got, exp), (which, text[:80])


def test_unicode_tables_match_the_regex_module(native):
    """every class used by the scanner, for every code point, against `regex` itself"""
    # probe the tables through the splitter: a lone character's class decides how "xC" splits
    L, N, S = re.compile(r"\\p{L}"), re.compile(r"\\p{N}"), re.compile(r"\\s")
    rng = random.Random(1)
    cps = list(range(0, 0x3000)) + rng.sample(range(0x3000, 0xD800), 3000) + rng.sample(range(0xE000, 0x110000), 3000)
    # batch: "<c>a <c>1 <c>!" for many c, compared chunk-for-chunk with regex
    for base in range(0, len(cps), 500):
        text = "".join(f"{chr(c)}a {chr(c)}1{chr(c)}!{chr(c)}\\n" for c in cps[base:base + 500])
        check(native, text)


@pytest.mark.parametrize("text", [
    "", " ", "  ", "\\n", "a", "'", "''", "'s", "'ſ", "'S'T", "don't I'll we've you're he'd I'm", "DON'T I'LL",
    "hello world", " hello  world ", "   ", "a   ", "a   b", "a \\n b", "a\\n\\nb", "a \\n\\n b", "a\\r\\nb", " \\r\\n\\t x",
    "x\\n", "x\\n ", "x \\n \\n  y", "\\t\\tindented", "\\tx", " \\tx", "\\xa0x", "!\\n\\nx", "!!!\\r\\n\\r\\n", " !!!?\\n",
    "12345 678 9", "a1b22c333d4444", " 12", "1 2", "٣٤٥٦", "x²y", "½ cup", "hello123!!!? (안녕하세요!) lol123 😉",
    "<|endoftext|>Hello", "e\\u0301a", "a\\u0301b", "\\u0301\\u0301", "🇦🇧 flags", "\\x00\\x01a", "'x'y'z", "a'b", "a'sb",
    "end with space ", "end with spaces   ", "tab\\tsep\\tvalues", "mixed \\u3000 wide space", "\\u2028line sep",
])
def test_adversarial_strings(native, text):
    check(native, text.encode().decode("unicode_escape") if "\\\\" in text else text)


def test_fuzz_tricky_code_points(native):
    rng = random.Random(20260921)
    for trial in range(4000):
        k = rng.randint(1, 14)
        text = "".join(rng.choice(TRICKY) for _ in range(k))
        check(native, text)


def test_longer_mixed_text_and_threads(native):
    rng = random.Random(7)
    base = native.synth_text(400_000, 13).decode()
    # sprinkle tricky material into natural-looking text
    parts = []
    pos = 0
    while pos < len(base):
        step = rng.randint(20, 400)
        parts.append(base[pos:pos + step])
        parts.append("".join(rng.choice(TRICKY) for _ in range(rng.randint(0, 4))))
        pos += step
    text = "".join(parts)
    check(native, text, threads=1)
    data = text.encode()
    for which in (2, 4):  # segment-parallel scan gives the same offsets
        assert np.array_equal(native.split_offsets(data, which, 1), native.split_offsets(data, which, 5))
    big = native.synth_text(7_000_000, 14)
    for which in (2, 4):
        one = native.split_offsets(big, which, 1)
        assert np.array_equal(one, native.split_offsets(big, which, 7))
        # more segments than cores, every sink copied out by its own thread (over 2^20 offsets)
        assert len(one) > (1 << 20) and np.array_equal(one, native.split_offsets(big, which, 64))


def test_tokenizer_chunking_matches_regex_path(native):
    """RegexTokenizer._chunked: native scanner for the GPT patterns, `regex` for anything else;
    like the reference, chunking follows compiled_pattern (load() does not recompile it)."""
    from minbpe_amd import RegexTokenizer
    from minbpe_amd.tokenizer import _concat_chunks
    text = "Don't panic: it's 12345 o'clock...\n\n  The cafe\u0301 (Ελληνικά) costs 2,50 €!\r\n" + native.synth_text(20000, 3).decode()
    for pat in (None, GPT2_SPLIT_PATTERN, r"\w+|\s+|[^\w\s]+", r"[a-z]+"):
        tok = RegexTokenizer(pat)
        data, offs = tok._chunked(text)
        exp_data, exp_offs = _concat_chunks(tok._split(text))
        assert data == exp_data and np.array_equal(offs, exp_offs), pat
    tok = RegexTokenizer()
    tok.pattern = "something else"          # what load() does; compiled_pattern is untouched
    assert tok._chunked("a b")[1].tolist() == [0, 1]
