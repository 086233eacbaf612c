"""Generate tests/golden/golden.json by RUNNING THE REFERENCE (karpathy/minbpe at
/root/reference, read-only) in the build container.  The GPU box has no
/root/reference, so the vectors travel as this committed fixture.

    python tests/golden/gen_golden.py

Inputs are either inline strings or synth_text(n, seed) (regenerated at test
time from the pinned generator in libbpe_hip.so; sha256 stored here).
tiktoken is not installed: a stub module satisfies
minbpe/__init__.py's import; GPT4Tokenizer is never constructed.
"""
import hashlib
import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.modules.setdefault("tiktoken", types.ModuleType("tiktoken"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from minbpe import BasicTokenizer, RegexTokenizer  # noqa: E402  (the reference)
from minbpe.base import get_stats, merge  # noqa: E402
from minbpe_amd import synth_text  # noqa: E402  (host-only generator)

PROSE = (
    "The harbour at first light is a quiet ledger of small sounds: rope against "
    "timber, the slap of water on a hull, a gull arguing with nobody. Marta counted "
    "the crates twice -- forty-one, forty-one -- and wrote the number in the book. "
    "\"If the ferry is late again,\" she said, \"we'll lose the tide, and then we'll "
    "lose the day.\" Nobody answered; they'd heard it before.\n\n"
    "By 7:45 the fog had lifted. The café on the corner (Ελληνικός καφές, 2,50 €) "
    "opened its shutters, and the baker's boy ran past with 12 loaves under one arm. "
    "It's strange, she thought, how the same street can be three different streets "
    "in one morning... Привет, сказал кто-то. 你好, said someone else. 😀\n"
    "She'd written: don't forget the invoices; they're in the blue folder, not the red one!\n"
)


def run_train(cls, text, vocab_size, pattern=None):
    tok = cls() if pattern is None else cls(pattern)
    try:
        tok.train(text, vocab_size)
        return tok, [list(p) for p in tok.merges], False
    except ValueError as e:
        assert "empty" in str(e)
        return None, None, True


def main():
    rng = random.Random(20260921)
    cases = []

    def add_train(name, text=None, synth=None, vocab_size=256, kinds=("basic", "regex"),
                  encode_texts=(), pattern=None):
        src = text if synth is None else synth_text(*synth).decode("utf-8")
        for kind in kinds:
            cls = BasicTokenizer if kind == "basic" else RegexTokenizer
            tok, merges, err = run_train(cls, src, vocab_size, pattern if kind == "regex" else None)
            case = dict(name=f"{name}-{kind}", kind=kind, vocab_size=vocab_size, merges=merges,
                        raises_value_error=err)
            if pattern is not None and kind == "regex":
                case["pattern"] = pattern  # RegexTokenizer(pattern) (regex.py:24-32)
            if synth is None:
                case["text"] = text
            else:
                case["synth"] = list(synth)
                case["sha256"] = hashlib.sha256(src.encode()).hexdigest()
            if tok is not None:
                enc = []
                for et in encode_texts:
                    enc.append(dict(text=et, ids=tok.encode(et)))
                case["encode"] = enc
            cases.append(case)
            print(name, kind, "err" if err else len(merges), flush=True)

    add_train("wiki", text="aaabdaaabac", vocab_size=259, encode_texts=["aaabdaaabac", "", "a", "aaaa"])
    add_train("abab", text="ab ab ab ab", vocab_size=258, encode_texts=["ab ab", "ba ba ab"])
    add_train("exhaust-ab", text="ab", vocab_size=258)
    add_train("abcd", text="abcd", vocab_size=259)
    add_train("empty", text="", vocab_size=256)
    add_train("empty-1", text="", vocab_size=257)
    add_train("single", text="?", vocab_size=257)
    add_train("multilingual", text="hello world!!!? (안녕하세요!) lol123 😉", vocab_size=256 + 12,
              encode_texts=["hello world!!!? (안녕하세요!) lol123 😉", "hello lol", "😉😉"])
    add_train("prose", text=PROSE, vocab_size=256 + 96, encode_texts=[PROSE, PROSE[:200], "the tide"])
    add_train("runs-a1001", text="a" * 1001, vocab_size=256 + 9, encode_texts=["a" * 77])
    add_train("runs-mixed", text=("aaab" * 40 + "aa aaa aaaa aaaaa " * 30 + "b" * 33), vocab_size=256 + 24,
              encode_texts=["aaaaaaa aaab"])
    for k in (2, 4, 16):
        s = "".join(chr(97 + rng.randrange(k)) for _ in range(3000))
        add_train(f"alpha{k}", text=s, vocab_size=256 + 60, kinds=("basic",), encode_texts=[s[:500]])
        s2 = "".join(rng.choice([chr(97 + rng.randrange(k)), chr(97 + rng.randrange(k)), " "])
                     for _ in range(3000))
        add_train(f"alpha{k}-spaces", text=s2, vocab_size=256 + 60, encode_texts=[s2[:500]])
    add_train("synth-60k", synth=(60_000, 7), vocab_size=256 + 160,
              encode_texts=[synth_text(4000, 8).decode()])
    # other split patterns: the GPT-2 one (regex.py:18; native scanner) and a custom one (regex module)
    from minbpe.regex import GPT2_SPLIT_PATTERN
    add_train("prose-gpt2", text=PROSE, vocab_size=256 + 64, kinds=("regex",), pattern=GPT2_SPLIT_PATTERN,
              encode_texts=[PROSE[:300], "don't  stop\r\n  now"])
    add_train("synth-gpt2", synth=(50_000, 9), vocab_size=256 + 120, kinds=("regex",), pattern=GPT2_SPLIT_PATTERN,
              encode_texts=[synth_text(3000, 10).decode()])
    add_train("prose-custom", text=PROSE, vocab_size=256 + 48, kinds=("regex",),
              pattern=r"\p{L}+|\p{N}+|[^\p{L}\p{N}]+", encode_texts=[PROSE[:300]])
    add_train("synth-150k", synth=(150_000, 11), vocab_size=256 + 300, kinds=("regex",),
              encode_texts=[synth_text(5000, 12).decode()])

    # get_stats / merge primitives on random lists (dict order matters)
    prims = []
    for t in range(40):
        k = rng.choice([2, 3, 5, 50, 300])
        n = rng.choice([0, 1, 2, 3, 10, 100, 1000])
        ids = [rng.randrange(k) for _ in range(n)]
        st = get_stats(ids)
        entry = dict(ids=ids, stats=[[a, b, c] for (a, b), c in st.items()])
        if st:
            pair = max(st, key=st.get)
            entry["argmax"] = list(pair)
            entry["merged"] = merge(ids, pair, 1000)
            same = (ids[0], ids[0]) if ids else (0, 0)
            entry["merged_same"] = dict(pair=list(same), out=merge(ids, same, 1001))
        prims.append(entry)

    # save/load format sample (base.py:97-165)
    tok = RegexTokenizer()
    tok.train("ab ab ab ab", 258)
    tok.register_special_tokens({"<|endoftext|>": 1000})
    tok.save("/tmp/_golden_tok")
    model_text = open("/tmp/_golden_tok.model", encoding="utf-8").read()
    vocab_text = open("/tmp/_golden_tok.vocab", encoding="utf-8").read()
    tok2 = RegexTokenizer()
    tok2.train(PROSE, 256 + 40)
    tok2.register_special_tokens({"<|endoftext|>": 100257, "<|fim_prefix|>": 100258})
    sp_text = "<|endoftext|>" + PROSE[:300] + "<|fim_prefix|>" + PROSE[300:500] + "<|endoftext|>"
    specials = dict(train_text=PROSE, vocab_size=256 + 40,
                    special_tokens=tok2.special_tokens, text=sp_text,
                    ids_all=tok2.encode(sp_text, allowed_special="all"),
                    ids_none=tok2.encode(sp_text, allowed_special="none"),
                    ids_set=tok2.encode(sp_text, allowed_special={"<|endoftext|>"}))

    out = dict(
        generated_by="tests/golden/gen_golden.py against /root/reference (karpathy/minbpe)",
        train=cases, primitives=prims,
        model_file=dict(text=model_text, vocab=vocab_text),
        specials=specials,
        # values recorded in SURVEY.md section 8c, reproduced by this script's sibling
        # check in tests/test_oracle_golden.py when /root/reference is present
        taylorswift=dict(sha256_prefix="c2e39cb822d4ae0c", basic512_merges_hash="96e771b35363a8bb",
                         regex512_merges_hash="9f07a31fd677129a",
                         basic512_encode_hash="ce7a17d6d8d7a290", basic512_encode_len=78746,
                         regex512_encode_hash="b541abe880dfdc74", regex512_encode_len=87339),
    )
    with open(os.path.join(HERE, "golden.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote golden.json", os.path.getsize(os.path.join(HERE, "golden.json")))


if __name__ == "__main__":
    main()
