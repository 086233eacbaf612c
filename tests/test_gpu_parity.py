"""GPU: the HIP path (through the C-ABI) against the CPU oracle and the golden
vectors generated from the reference.  Integer work: every comparison is exact."""
import random

import numpy as np
import pytest

import oracle
from helpers import case_text, data_for, split_chunks, toy_rank_table

pytestmark = pytest.mark.gpu


def starts_to_offsets(starts, n):
    return None if starts is None else np.asarray(starts, dtype=np.uint64)


# ---------------------------------------------------------------------------
# single-step primitives

def test_primitives_golden(golden, engine):
    for prim in golden["primitives"]:
        ids = prim["ids"]
        engine.load_ids(ids)
        st = engine.get_stats()
        assert [[a, b, c] for (a, b), c, _ in st] == prim["stats"]
        if prim["stats"]:
            pair, cnt = engine.argmax()
            assert list(pair) == prim["argmax"]
            assert cnt == max(c for _, _, c in prim["stats"])
            engine.merge(prim["argmax"], 1000)
            assert engine.read_ids().tolist() == prim["merged"]
            engine.load_ids(ids)
            engine.merge(prim["merged_same"]["pair"], 1001)
            assert engine.read_ids().tolist() == prim["merged_same"]["out"]
        else:
            with pytest.raises(ValueError):
                engine.argmax()


@pytest.mark.parametrize("k,n", [(2, 5000), (3, 70000), (40, 200000), (256, 1 << 20), (2, 4096),
                                 (2, 4097), (2, 8191), (1, 10000), (5, 1), (5, 2), (5, 0)])
def test_get_stats_argmax_merge_vs_oracle(engine, k, n):
    rng = np.random.default_rng(n * 31 + k)
    ids = rng.integers(0, k, size=n, dtype=np.int32)
    # ragged chunks: boundaries at random places, including adjacent ones
    nb = min(n, max(1, n // 7))
    offs = np.unique(np.concatenate([[0], rng.integers(0, max(n, 1), size=nb)])).astype(np.uint64) if n else None
    for offsets in (None, offs):
        if offsets is None and n == 0:
            continue
        engine.load_ids(ids, offsets)
        ref = oracle.get_stats(ids, offsets)
        got = engine.get_stats()
        assert got == ref
        if not ref:
            with pytest.raises(ValueError):
                engine.argmax()
            continue
        best = max(ref, key=lambda e: e[1])  # first max in insertion order
        pair, cnt = engine.argmax()
        assert (pair, cnt) == (best[0], best[1])
        for mp in (pair, (int(ids[0]), int(ids[0]))):
            engine.load_ids(ids, offsets)
            engine.merge(mp, 300)
            exp_ids, off_full = oracle.merge_chunks(ids, offsets, mp, 300)
            nl = len(exp_ids)
            assert len(engine) == nl
            assert np.array_equal(engine.read_ids(), exp_ids)
            if offsets is not None:
                # chunk starts follow the tokens; empty chunks leave no mark
                exp = sorted(set(int(off_full[c]) for c in range(len(off_full) - 1)
                                 if off_full[c + 1] > off_full[c]))
                assert engine.read_chunk_starts().tolist() == exp


@pytest.mark.parametrize("mimpl,scan_sup", [(0, 1024), (0, 0), (1, 1024)])
def test_long_runs_cross_tile_boundaries(engine, mimpl, scan_sup):
    # a == b merges with runs spanning many 4096-id tiles, odd/even lengths; scan_sup = 0: the tile scan of long
    # streams (k_tile_sup / k_tile_scan_sup / k_tile_expand) on every stream, 1024 (the default): beyond 1024 tiles
    engine.set_option("merge", mimpl)
    engine.set_option("scan_sup", scan_sup)
    for n in (4095, 4096, 4097, 12289, 100001, 64 * 4096 + 3, 300 * 4096):
        ids = np.full(n, 7, np.int32)
        engine.load_ids(ids)
        engine.merge((7, 7), 300)
        exp = oracle.merge(ids, (7, 7), 300)
        assert np.array_equal(engine.read_ids(), exp)
    rng = np.random.default_rng(5)
    ids = np.where(rng.random(300000) < 0.97, 7, 8).astype(np.int32)
    engine.load_ids(ids)
    engine.merge((7, 7), 300)
    assert np.array_equal(engine.read_ids(), oracle.merge(ids, (7, 7), 300))
    # a != b over many tiles, sites straddling tile boundaries (period 3 vs tile 4096)
    ids = np.tile(np.array([1, 2, 3], np.int32), 200000)
    for pair in ((1, 2), (3, 1), (2, 3)):
        engine.load_ids(ids)
        engine.merge(pair, 300)
        assert np.array_equal(engine.read_ids(), oracle.merge(ids, pair, 300))
    engine.set_option("merge", 0)
    engine.set_option("scan_sup", 1024)


# ---------------------------------------------------------------------------
# the training loop

def test_train_golden_cases(golden, engine, native):
    for case in golden["train"]:
        data, offs = data_for(case, native)
        nm = case["vocab_size"] - 256
        engine.load_bytes(data, offs)
        if case["raises_value_error"]:
            with pytest.raises(ValueError, match="empty sequence"):
                engine.train(nm)
            continue
        res = engine.train(nm)
        assert [list(p) for p in res["pairs"]] == case["merges"], case["name"]


# engine variants: (mode, merge impl, slots, sparse, lean) -- slots 2 = the second slotted form (default);
# sparse 2 = every a != b pass goes through the inverted index and the sparse kernel; lean 1 (default) =
# lean iterations (k_lean.hip: three launches, table updated at the merge sites, a == b deferred to the
# general path) once the host has seen a count <= lean_count, 2 = from the first merge on, 0 = never,
# 3 = as 2 but every selection reads the whole row-maxima array (k_rowsel_lean; option lean_sum = 0)
# instead of the previous table update's per-wave records (k_sel_lean, the default), 4 = as 2 but every
# iteration selects (option lean_chain = 0: no tied pair is merged off an earlier selection's list), 5 = as 2
# but a == b passes visit every slot and mark what they rewrite for an index rebuild (option aa_sparse = 0)
VARIANTS = [(0, 0, 0, 1, 1), (1, 0, 1, 1, 1), (1, 0, 0, 1, 1), (1, 1, 0, 1, 1), (0, 1, 0, 1, 1), (1, 0, 2, 1, 1),
            (1, 0, 2, 2, 1), (1, 0, 2, 0, 1), (1, 0, 2, 1, 0), (1, 0, 2, 2, 0), (1, 0, 2, 1, 2), (1, 0, 2, 2, 2),
            (1, 0, 2, 2, 3), (1, 0, 2, 2, 4), (1, 0, 2, 2, 5), (1, 0, 2, 1, 7), (1, 0, 2, 2, 7), (1, 0, 2, 1, 8),
            (1, 0, 2, 1, 9), (1, 0, 2, 2, 9), (1, 0, 2, 2, 10)]


def set_variant(engine, mode, mimpl, slots, sparse, lean=1):
    """lean: 0 never | 1 the default engine (chain steps, k_chain.hip, wherever lean iterations would run with the
    index live) | 2 lean iterations forced onto every merge, no chain steps | 3, 4, 5 variants of 2 (selection from
    the whole row-maxima array; no chained merges; a == b passes over every slot) | 7 = 2 with chain steps |
    8 = 1 without chain steps (round 3's default engine) | 9 = 1 and 10 = 7 with the re-packing into 256-id slots (kernels of
    namespace bpe_g1) forced onto streams of any size at the first index build (option small_slots = 2; the default does
    it for streams of more than 16 Ki slots only)"""
    engine.set_option("mode", mode)
    engine.set_option("merge", mimpl)
    engine.set_option("slots", slots)
    engine.set_option("sparse", sparse)
    engine.set_option("lean", 1 if lean in (1, 8, 9) else (2 if lean >= 2 else 0))
    engine.set_option("chain", 1 if lean in (1, 7, 9, 10) else 0)
    engine.set_option("small_slots", 2 if lean in (9, 10) else 1)
    engine.set_option("lean_sum", 0 if lean == 3 else 1)
    engine.set_option("lean_chain", 0 if lean == 4 else 1)
    engine.set_option("aa_sparse", 0 if lean == 5 else 1)
    # lean >= 2 forces the lean iterations onto every merge (coverage of their kernels and of the hand-back):
    # no general-path stretches after clustered deferrals there
    engine.set_option("lean_backoff", 0 if lean in (2, 3, 4, 5, 7, 10) else 1)


def reset_variant(engine):
    set_variant(engine, 1, 0, 2, 1, 1)
    engine.set_option("depth", 8)


@pytest.mark.parametrize("mode,mimpl,slots,sparse,lean", VARIANTS)
@pytest.mark.parametrize("k,n,nm", [(2, 3000, 40), (4, 50000, 120), (16, 200000, 150), (3, 9000, 300),
                                    (1, 70000, 20), (2, 4096 * 3 + 1, 64)])
def test_train_tie_heavy_vs_oracle(engine, mode, mimpl, slots, sparse, lean, k, n, nm):
    rng = random.Random(k * 1000 + n)
    data = bytes(97 + rng.randrange(k) for _ in range(n))
    set_variant(engine, mode, mimpl, slots, sparse, lean)
    try:
        engine.load_bytes(data)
        exp = oracle.train(data, nm, raise_on_empty=False)
        if len(exp[0]) < nm:  # the oracle ran out of pairs: so must we, at the same merge
            with pytest.raises(ValueError):
                engine.train(nm)
            res = engine.last_train
        else:
            res = engine.train(nm)
        assert res["pairs"] == exp[0]
        assert res["counts"] == exp[1]
        assert res["lens"] == exp[2]
        # the resident stream is the oracle's final stream
        ids = np.frombuffer(data, dtype=np.uint8).astype(np.int32)
        for i, p in enumerate(exp[0]):
            ids = oracle.merge(ids, p, 256 + i)
        assert np.array_equal(engine.read_ids(), ids)
    finally:
        reset_variant(engine)


@pytest.mark.parametrize("mode,mimpl,slots,sparse,lean", [v for v in VARIANTS if v != (0, 1, 0, 1, 1)])
@pytest.mark.parametrize("kind", ["basic", "regex"])
def test_train_synth_2mb_vs_oracle(engine, native, kind, mode, mimpl, slots, sparse, lean):
    text = native.synth_text(2_000_000, 11)
    if kind == "basic":
        data, offs = text, None
    else:
        data, offs = split_chunks(text.decode())
    nm = 400
    exp = oracle.train(data, nm, offs)
    set_variant(engine, mode, mimpl, slots, sparse, lean)
    try:
        engine.load_bytes(data, offs)
        res = engine.train(nm)
        stats = engine.train_stats()
        if sparse == 2:
            # every a != b pass a sparse one; a chain step (k_chain.hip) is ONE pass for all the merges of its batch
            chain_merges = stats["chained"] + stats["selections"] if stats["steps"] else 0
            assert stats["sparse"] == nm - chain_merges + stats["steps"]
        if slots == 2 and mode == 1:
            n_same = sum(a == b for a, b in exp[0])
            if lean == 0:
                assert stats["lean"] == 0 and stats["deferred"] == 0
            elif lean in (2, 3, 4, 5, 7):
                # every merge is a lean iteration or was handed back to the general path: all a == b ones
                # are, and (index live) those whose tie the lean selection could not settle by itself
                assert stats["lean"] + stats["deferred"] == nm and stats["deferred"] >= n_same
                if sparse != 2 and lean != 7:
                    assert stats["deferred"] == n_same
                if lean == 7:
                    assert stats["steps"] > 0 and stats["selections"] <= stats["steps"]
            else:
                assert stats["lean"] > 0
            if lean in (9, 10) and stats["index_builds"]:
                assert stats["slot_ids"] == 256, stats  # (the stream was re-packed into 256-id slots when the index was built)
        assert res["pairs"] == exp[0]
        assert res["counts"] == exp[1]
        assert res["lens"] == exp[2]
        # re-running from the resident bytes gives the same answer (bench does this)
        assert engine.train(nm)["pairs"] == exp[0]
        # depth 0 = host waits for every iteration: same result
        engine.set_option("depth", 0)
        assert engine.train(nm)["pairs"] == exp[0]
        # the resident stream after training is the oracle's (slots are re-packed at the end)
        ids = np.frombuffer(data, dtype=np.uint8).astype(np.int32)
        o = None if offs is None else offs
        for i, p in enumerate(exp[0]):
            ids, off_full = oracle.merge_chunks(ids, o, p, 256 + i)
            o = off_full[:-1]
        assert np.array_equal(engine.read_ids(), ids)
    finally:
        reset_variant(engine)


# ---------------------------------------------------------------------------
# the drop-in classes

@pytest.mark.parametrize("dedup", [True, False])
def test_classes_golden(golden, native, dedup):
    from minbpe_amd import BasicTokenizer, RegexTokenizer
    for case in golden["train"]:
        if case["kind"] == "basic" and not dedup:
            continue  # BasicTokenizer has one chunk: nothing to de-duplicate either way
        cls = BasicTokenizer if case["kind"] == "basic" else RegexTokenizer
        tok = cls(case["pattern"]) if case.get("pattern") else cls()
        tok.dedup = dedup
        text = case_text(case, native)
        if case["raises_value_error"]:
            with pytest.raises(ValueError):
                tok.train(text, case["vocab_size"])
            assert tok.merges == {}
            continue
        tok.train(text, case["vocab_size"])
        assert [list(p) for p in tok.merges] == case["merges"]
        assert list(tok.merges.values()) == list(range(256, 256 + len(case["merges"])))
        for enc in case["encode"]:
            ids = tok.encode(enc["text"])
            assert ids == enc["ids"], case["name"]
            assert tok.decode(ids) == enc["text"]


def test_wikipedia_example():
    # the reference's own known-answer test (tests/test_tokenizer.py:80-107)
    from minbpe_amd import BasicTokenizer, RegexTokenizer
    for cls in (BasicTokenizer, RegexTokenizer):
        tok = cls()
        tok.train("aaabdaaabac", 256 + 3)
        assert tok.encode("aaabdaaabac") == [258, 100, 258, 97, 99]
        assert tok.decode(tok.encode("aaabdaaabac")) == "aaabdaaabac"


def test_specials_and_save_load(golden, tmp_path):
    # mirrors tests/test_tokenizer.py:109-132 with our own text
    from minbpe_amd import RegexTokenizer
    sp = golden["specials"]
    tok = RegexTokenizer()
    tok.train(sp["train_text"], sp["vocab_size"])
    tok.register_special_tokens(sp["special_tokens"])
    assert tok.encode(sp["text"], allowed_special="all") == sp["ids_all"]
    assert tok.encode(sp["text"], allowed_special="none") == sp["ids_none"]
    assert tok.encode(sp["text"], allowed_special={"<|endoftext|>"}) == sp["ids_set"]
    assert tok.decode(sp["ids_all"]) == sp["text"]
    with pytest.raises(AssertionError):
        tok.encode(sp["text"])
    prefix = str(tmp_path / "t")
    tok.save(prefix)
    t2 = RegexTokenizer()
    t2.load(prefix + ".model")
    assert t2.encode(sp["text"], "all") == sp["ids_all"]
    assert t2.decode(sp["ids_all"]) == sp["text"]


def test_verbose_print_matches_reference_format(capsys):
    from minbpe_amd import BasicTokenizer
    tok = BasicTokenizer()
    tok.train("aaabdaaabac", 259, verbose=True)
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "merge 1/3: (97, 97) -> 256 (b'aa') had 4 occurrences"
    assert out[2] == "merge 3/3: (257, 98) -> 258 (b'aaab') had 2 occurrences"


def test_module_level_helpers():
    from minbpe_amd import get_stats, merge
    assert get_stats([1, 2, 3, 1, 2]) == {(1, 2): 2, (2, 3): 1, (3, 1): 1}
    assert list(get_stats([5, 6, 7, 8, 5, 6, 7, 8])) == [(5, 6), (6, 7), (7, 8), (8, 5)]
    assert get_stats([1, 2], {(1, 2): 5}) == {(1, 2): 6}
    assert merge([1, 2, 3, 1, 2], (1, 2), 4) == [4, 3, 4]
    assert merge([7, 7, 7, 7, 7], (7, 7), 9) == [9, 9, 7]
    assert merge([], (1, 2), 4) == [] and get_stats([]) == {} and get_stats([3]) == {}


# ---------------------------------------------------------------------------
# K4: batched encode through the C-ABI

def _train_pairs(native, n, nm, seed, kind):
    text = native.synth_text(n, seed)
    data, offs = (text, None) if kind == "basic" else split_chunks(text.decode())
    return oracle.train(data, nm, offs)[0]


# cache: (enc_cache, enc_hash_bits) -- the chunk cache on (default) / off; on with the chunk hash cut to 3 bits,
# so that thousands of different chunks share a slot and only the byte comparison keeps them apart; bits = -1:
# the default cache with the three-launch offsets + placement instead of the chained pass (enc_chain = 0)
ENC_VARIANTS = [(1, 0), (0, 0), (1, 3), (1, -1)]


def _enc_variant(engine, cache, bits):
    engine.set_option("enc_cache", cache)
    engine.set_option("enc_hash_bits", max(bits, 0))
    engine.set_option("enc_chain", 0 if bits < 0 else 1)


@pytest.mark.parametrize("cache,bits", ENC_VARIANTS)
@pytest.mark.parametrize("kind", ["basic", "regex"])
def test_encode_batch_vs_oracle(engine, native, kind, cache, bits):
    pairs = _train_pairs(native, 300_000, 600, 21, kind)
    text = native.synth_text(150_000, 22)
    data, offs = (text, None) if kind == "basic" else split_chunks(text.decode())
    exp_ids, exp_off = oracle.encode(pairs, data, offs)
    _enc_variant(engine, cache, bits)
    try:
        ids, out_off = engine.encode_batch(np.array(pairs, np.int32), None, data, offs)
    finally:
        _enc_variant(engine, 1, 0)
    assert np.array_equal(ids, exp_ids)
    assert np.array_equal(out_off, exp_off)


@pytest.mark.parametrize("kind", ["own_text", "other_text", "letters"])
def test_encode_one_giant_chunk_is_a_replay_of_training(engine, native, kind):
    """BasicTokenizer.encode (basic.py:57-74): the whole text is ONE chunk of megabytes.  With a merge list of the shape
    training makes, the library replays the list through the training engine (forced selections: k_forced_sel /
    k_forced_pair) instead of sweeping the stream once per rank -- equal to oracle.encode, and to the stream-wide rounds
    (option enc_replay = 0), on the text the merges were trained on, on another text (most late pairs never occur: merges
    with zero sites) and on a four-letter text (a == a merges at every other step, runs of one letter: F2)."""
    if kind == "letters":
        rng = np.random.default_rng(3)
        text = bytes(97 + rng.integers(0, 4, size=1_300_000).astype(np.uint8))
        pairs = oracle.train_fast(text[:400_000], 300)[0]
    else:
        text = native.synth_text(1_500_000, 81)
        pairs = oracle.train_fast(text if kind == "own_text" else native.synth_text(700_000, 82), 700)[0]
    exp_ids, exp_off = oracle.encode(pairs, text, None)
    tp = np.array(pairs, np.int32)
    ids, out_off = engine.encode_batch(tp, None, text, None)
    assert np.array_equal(ids, exp_ids) and list(out_off) == [0, len(exp_ids)]
    assert engine.train_stats()["steps"] > 0  # (it did go through the training engine's chain steps)
    engine.set_option("enc_replay", 0)
    try:
        ids2, _ = engine.encode_batch(tp, None, text, None)
    finally:
        engine.set_option("enc_replay", 1)
    assert np.array_equal(ids2, exp_ids)
    if kind == "own_text":  # training applies its merges in order: the stream it leaves IS encode of its own text
        engine.load_bytes(text)
        engine.train(len(pairs))
        assert np.array_equal(engine.read_ids(), exp_ids)


@pytest.mark.parametrize("cache,bits", ENC_VARIANTS)
def test_encode_batch_repeated_and_unaligned_chunks(engine, native, cache, bits):
    """The cache's own cases: the same chunk at every byte alignment, chunks that differ in one byte (first,
    last, beyond the 8th / 16th / 24th), prefixes of one another, lengths 1..32, many repeats."""
    pairs = _train_pairs(native, 200_000, 500, 23, "regex")
    base = native.synth_text(4000, 26)
    rng = np.random.default_rng(5)
    words = []
    for rep in range(40):
        for L in (1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 25, 31, 32):
            w = bytearray(base[100:100 + L])
            words.append(bytes(w))
            if rep % 3 == 1:
                w[int(rng.integers(0, L))] ^= 1  # one byte off, somewhere
                words.append(bytes(w))
            if rep % 5 == 2:
                words.append(bytes(w[:max(1, L - 1)]))  # a prefix
        words.append(b"x" * (1 + rep % 7))  # shifts the alignment of everything that follows
    data = b"".join(words)
    offs = np.cumsum([0] + [len(w) for w in words[:-1]]).astype(np.uint64)
    exp_ids, exp_off = oracle.encode(pairs, data, offs)
    _enc_variant(engine, cache, bits)
    try:
        ids, out_off = engine.encode_batch(np.array(pairs, np.int32), None, data, offs)
    finally:
        _enc_variant(engine, 1, 0)
    assert np.array_equal(ids, exp_ids)
    assert np.array_equal(out_off, exp_off)


@pytest.mark.parametrize("acc", [0, 5, 200, 10000])
@pytest.mark.parametrize("kind", ["basic", "regex"])
def test_repack_policy_vs_oracle(engine, native, kind, acc):
    """Option repack_acc (when the dense phase re-packs its slots: at a fixed fill of 31/32, after nearly every sweep,
    the default, never) changes the slot layout a pass works on, never the merges: the default engine on a 3 MB text,
    dense sweeps first, then the index."""
    text = native.synth_text(3_000_000, 19)
    data, offs = (text, None) if kind == "basic" else split_chunks(text.decode())
    nm = 500
    exp = oracle.train(data, nm, offs)
    reset_variant(engine)
    engine.set_option("repack_acc", acc)
    try:
        engine.load_bytes(data, offs)
        res = engine.train(nm)
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2]
        assert engine.train_stats()["dense"] > 0
    finally:
        engine.set_option("repack_acc", 200)


CHAIN_OPTIONS = [
    (("chain_kcap", 1),), (("chain_kcap", 4),), (("chain_kcap", 8),),   # batches of one, four, eight (default 15)
    (("count_is_removed", 0),),                      # the ids a merge removes are counted, not taken from the pair's count
    (("chain_prefetch", 0),),                        # no register prefetch of the next candidate slot
    (("small_slots", 0),), (("small_slots", 2),),    # 1024-id slots throughout / 256-id slots from the first index build
    (("small_slots", 2), ("chain_kcap", 2), ("pool_hint", 64)),
    (("chain_scan", 1),), (("chain_scan", 63),), (("chain_scan", 255),),    # one / 63 / 255 scanning workgroups in a pool rebuild (default 127)
    # a step as ONE launch (k_step: selection -> published batch -> merge pass -> grid barrier -> table update) instead of three
    (("fuse_step", 1),), (("fuse_step", 1), ("chain_kcap", 4)), (("fuse_step", 1), ("count_is_removed", 0)),
    (("fuse_step", 1), ("small_slots", 2)), (("fuse_step", 1), ("small_slots", 0)),
    (("fuse_step", 1), ("lean_grid", 8)), (("fuse_step", 1), ("lean_grid", 70)),   # ... on a grid of 8 / 70 workgroups (its phases deal the work by the grid)
    (("lean_grid", 8),),
]


@pytest.mark.parametrize("opts", CHAIN_OPTIONS)
@pytest.mark.parametrize("kind", ["regex", "ties"])
def test_chain_step_options_cross_check(engine, native, kind, opts):
    """Every option of the chain steps -- batch caps, the removal counters, the
    slot prefetch, both slot geometries, the number of scanning workgroups -- gives the oracle's merges: on a GPT-4-split
    text with every merge a chain step through the index (sparse = 2, lean = 2), and on a three-letter corpus where nearly
    every level is a tie, a == b pairs head the pool again and again and the table runs empty."""
    if kind == "regex":
        data, offs = split_chunks(native.synth_text(1_500_000, 71).decode())
        nm = 500
    else:
        rng = np.random.default_rng(12)
        chunks = [b" " + bytes(97 + rng.integers(0, 3, size=rng.integers(1, 6))) for _ in range(4000)]
        data = b"".join(chunks)
        offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
        nm = 400
    exp = oracle.train(data, nm, offs, raise_on_empty=False)
    defaults = {"chain_kcap": 15, "count_is_removed": 1, "chain_prefetch": 1,
                "small_slots": 1, "pool_hint": 0, "chain_scan": 127, "fuse_step": 0, "lean_grid": 256}
    set_variant(engine, 1, 0, 2, 2, 7)
    try:
        for k, v in opts:
            engine.set_option(k, v)
        engine.load_bytes(data, offs)
        if len(exp[0]) < nm:
            with pytest.raises(ValueError):
                engine.train(nm)
            res = engine.last_train
        else:
            res = engine.train(nm)
        st = engine.train_stats()
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2], opts
        assert st["steps"] > 0
        # the step is ONE launch (k_step.hip) wherever its selection is the pool and the option stands
        if ("fuse_step", 1) in opts:
            assert st["fused_steps"] > 0
        else:
            assert st["fused_steps"] == 0
        if ("small_slots", 2) in opts and st["index_builds"]:
            assert st["slot_ids"] == 256
        if ("small_slots", 0) in opts:
            assert st["slot_ids"] == 1024
    finally:
        for k, _ in opts:
            engine.set_option(k, defaults[k])
        reset_variant(engine)


def _long_chunk_text(native, seed):
    """chunks of 33 .. 6000 bytes among ordinary ones: URLs, identifiers, whitespace runs (a == a merges inside a chunk),
    a run of one letter, base64-like noise -- what code and logs put behind a GPT-style split"""
    rng = np.random.default_rng(seed)
    base = native.synth_text(60_000, seed).decode()
    words = base.split(" ")
    parts = []
    for i, w in enumerate(words):
        parts.append(w)
        if i % 17 == 3:
            parts.append("https://" + "".join(rng.choice(list("abcdefghij/._-0123456789"), int(rng.integers(30, 400)))))
        if i % 29 == 5:
            parts.append("x" + "_".join(words[max(0, i - 9):i + 1])[:int(rng.integers(33, 300))])
        if i % 41 == 7:
            parts.append("a" * int(rng.integers(33, 700)))
        if i % 53 == 11:
            parts.append("".join(rng.choice(list("ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"),
                                            int(rng.integers(500, 6000)))))
        if i % 37 == 9:
            parts.append(" " * int(rng.integers(34, 900)) + "\n")
        if i % 997 == 13:  # 4097 .. 9216 bytes (sixteen waves per chunk), and beyond (the stream-wide rounds)
            parts.append("".join(rng.choice(list("ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"),
                                            int(rng.integers(6000, 11000)))))
    return " ".join(parts)


@pytest.mark.parametrize("enc_long", [1, 0])
@pytest.mark.parametrize("cache", [1, 0])
def test_encode_batch_long_chunks_on_the_device(engine, native, cache, enc_long):
    """Chunks of more than 32 bytes (regex.py:92-109 on URLs, identifiers, whitespace runs): one wave per chunk with
    the chunk in LDS (k_enc_long, lengths up to 512, 4096 and 9216 bytes), the stream-wide rounds beyond that -- and
    option enc_long = 0, every long chunk through the rounds -- all equal to oracle.encode."""
    pairs = _train_pairs(native, 300_000, 900, 61, "regex")
    text = _long_chunk_text(native, 62)
    data, offs = split_chunks(text)
    lens = np.diff(np.append(offs, len(data)))
    assert (lens > 32).sum() > 300 and (lens > 512).sum() > 20 and (lens > 4096).sum() >= 3 and (lens > 9216).sum() >= 1
    assert ((lens > 4096) & (lens <= 9216)).sum() >= 3
    exp_ids, exp_off = oracle.encode(pairs, data, offs)
    engine.set_option("enc_cache", cache)
    engine.set_option("enc_long", enc_long)
    try:
        ids, out_off = engine.encode_batch(np.array(pairs, np.int32), None, data, offs)
    finally:
        engine.set_option("enc_cache", 1)
        engine.set_option("enc_long", 1)
    assert np.array_equal(out_off, exp_off)
    assert np.array_equal(ids, exp_ids)


_BIG_TABLE = {}


def _cl100k_sized(engine, native, kind):
    """100,000 merges around 20,000 trained ones (trained on the GPU: training parity has its own tests;
    what is compared below is encode against oracle.encode for the same table)."""
    from helpers import cl100k_shaped_table
    if "base" not in _BIG_TABLE:
        text = native.synth_text(6_000_000, 41)
        data, offs = text, native.split_offsets(text, 4)
        engine.load_bytes(data, offs)
        _BIG_TABLE["base"] = engine.train(20_000)["pairs"]
    if kind not in _BIG_TABLE:
        _BIG_TABLE[kind] = cl100k_shaped_table(_BIG_TABLE["base"], 100_000, 9, kind)
    return _BIG_TABLE[kind]


@pytest.mark.parametrize("cache,bits", ENC_VARIANTS)
@pytest.mark.parametrize("ids_kind", ["rank", "sparse"])
def test_encode_batch_with_a_cl100k_sized_rank_table(engine, native, cache, bits, ids_kind):
    """BASELINE.json configs[4] is GPT4Tokenizer.encode with cl100k_base: 100,000 merges, token ids up to
    100,255 (gpt4.py:60-79).  The ranks themselves are not available offline; a table of that size is
    (helpers.cl100k_shaped_table): more than 65,535 ranks, ids >= 65,536 (ids_kind "rank": 256 + rank;
    "sparse": a merges dict with non-consecutive values, ids up to 301,000), through every encoder variant."""
    pairs, mids = _cl100k_sized(engine, native, ids_kind)
    assert len(pairs) == 100_000 and native._lib.bpe_encode_uses_16bit(None, len(pairs)) == 0
    text = native.synth_text(1_500_000, 43) + " don't  stop 12345 ünïcödé 😉 ".encode() * 50
    offs = native.split_offsets(text, 4)
    exp_ids, exp_off = oracle.encode(pairs, text, offs, merge_ids=mids)
    assert int(exp_ids.max()) >= 65536 and len(np.unique(exp_ids)) > 5000
    _enc_variant(engine, cache, bits)
    try:
        ids, out_off = engine.encode_batch(pairs, mids, text, offs)
    finally:
        _enc_variant(engine, 1, 0)
    assert np.array_equal(ids, exp_ids)
    assert np.array_equal(out_off, exp_off)


def test_encode_one_stream_with_a_cl100k_sized_rank_table(engine, native):
    """BasicTokenizer.encode (one chunk: the stream-wide rounds, k_min_rank + merge) with the same table"""
    pairs, mids = _cl100k_sized(engine, native, "sparse")
    text = native.synth_text(60_000, 44)
    exp_ids, exp_off = oracle.encode(pairs, text, None, merge_ids=mids)
    ids, out_off = engine.encode_batch(pairs, mids, text, None)
    assert np.array_equal(ids, exp_ids) and np.array_equal(out_off, exp_off)
    assert int(exp_ids.max()) >= 65536


@pytest.mark.parametrize("kind", ["regex", "basic"])
def test_encode_batch_resident_equals_host_form(engine, native, kind):
    """bpe_encode_batch_resident (batch and outputs in HBM, the caller's buffers used in place) against bpe_encode_batch
    and the oracle: short chunks, chunks beyond 32 bytes (the stream-wide rounds read their offsets back), one stream;
    bad offsets are refused by the device-side check"""
    torch = pytest.importorskip("torch")
    pairs = _train_pairs(native, 300_000, 400, 33, kind)
    text = native.synth_text(700_000, 34) + b" " + b"x" * 200 + b" " + bytes(range(97, 123)) * 9 + b" tail"
    if kind == "regex":
        data, offs = split_chunks(text.decode())
    else:
        data, offs = text, np.zeros(1, np.uint64)
    exp_ids, exp_off = oracle.encode(pairs, data, offs)
    ids, out_off = engine.encode_batch(np.array(pairs, np.int32), None, data, offs)
    assert np.array_equal(ids, exp_ids) and np.array_equal(out_off, exp_off)
    dev = torch.device("cuda", 0)
    d_bytes = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_ids = torch.full((len(data),), -1, dtype=torch.int32, device=dev)
    d_ooff = torch.full((len(offs) + 1,), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = engine.encode_batch_resident(np.array(pairs, np.int32), None, d_bytes.data_ptr(), len(data), d_offs.data_ptr(),
                                         len(offs), d_ids.data_ptr(), d_ooff.data_ptr())
    assert total == len(exp_ids)
    assert np.array_equal(d_ids[:total].cpu().numpy(), exp_ids)
    assert np.array_equal(d_ooff.cpu().numpy().astype(np.uint64), exp_off)
    if kind == "regex":
        bad = d_offs.clone()
        bad[5] = bad[7]  # (no longer ascending at 6)
        torch.cuda.synchronize()
        with pytest.raises(Exception, match="ascend"):
            engine.encode_batch_resident(np.array(pairs, np.int32), None, d_bytes.data_ptr(), len(data), bad.data_ptr(),
                                         len(offs), d_ids.data_ptr(), d_ooff.data_ptr())


def test_encode_batch_long_and_short_chunks_mixed(engine, native):
    # chunk lengths around ENC_LMAX (32) and far beyond it, runs of one symbol, empty batch
    pairs = _train_pairs(native, 200_000, 500, 23, "regex")
    rng = np.random.default_rng(3)
    words = []
    base = native.synth_text(60_000, 24)
    pos = 0
    for L in [1, 2, 31, 32, 33, 34, 64, 65, 200, 5000, 3, 32, 33, 1, 40000, 7]:
        words.append(base[pos:pos + L])
        pos += L
    words += [b"a" * 33, b"a" * 32, b"ab" * 40, b"z"]
    data = b"".join(words)
    offs = np.cumsum([0] + [len(w) for w in words[:-1]]).astype(np.uint64)
    exp_ids, exp_off = oracle.encode(pairs, data, offs)
    ids, out_off = engine.encode_batch(np.array(pairs, np.int32), None, data, offs)
    assert np.array_equal(ids, exp_ids)
    assert np.array_equal(out_off, exp_off)
    ids, out_off = engine.encode_batch(np.array(pairs, np.int32), None, b"", None)
    assert len(ids) == 0


def test_encode_batch_custom_ids_and_no_merges(engine):
    # merge ids need not be 256 + rank (GPT4Tokenizer: id = rank value)
    pairs = np.array([[104, 105], [1000, 33]], np.int32)
    mids = np.array([1000, 2000], np.int32)
    ids, off = engine.encode_batch(pairs, mids, b"hi!hi!", np.array([0, 3], np.uint64))
    assert ids.tolist() == [2000, 2000] and off.tolist() == [0, 1, 2]
    ids, off = engine.encode_batch(np.zeros((0, 2), np.int32), None, b"abc", None)
    assert ids.tolist() == [97, 98, 99] and off.tolist() == [0, 3]


def test_encode_many_docs(engine, native):
    # cfg5 in miniature: a batch of documents, regex-chunked, against the oracle
    pairs = _train_pairs(native, 400_000, 1000, 31, "regex")
    text = native.synth_text(1_000_000, 32).decode()
    docs = [d for d in text.split("\n\n") if d]
    assert len(docs) > 500
    chunks = []
    import regex as re
    from minbpe_amd.tokenizer import GPT4_SPLIT_PATTERN
    pat = re.compile(GPT4_SPLIT_PATTERN)
    for d in docs:
        chunks += [c.encode() for c in re.findall(pat, d)]
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp_ids, exp_off = oracle.encode(pairs, data, offs)
    ids, out_off = engine.encode_batch(np.array(pairs, np.int32), None, data, offs)
    assert np.array_equal(ids, exp_ids) and np.array_equal(out_off, exp_off)


# ---------------------------------------------------------------------------
# data-parallel path (bpe_dp_*): several ranks emulated on the one GPU we have

def _space_chunks(text: bytes):
    import re
    return [c for c in re.findall(rb" ?[^ ]+| +", text) if c]


def _lockstep(native, chunks, nm, world, slots=2, dedup=False, sparse=1):
    """Drive `world` ctxs through the dist.py protocol in lock-step; reductions done by hand."""
    import torch
    from minbpe_amd.dist import GpuShard, shard_chunks
    shards = []
    for r in range(world):
        lo, hi = shard_chunks(len(chunks), r, world)
        mine = chunks[lo:hi]
        eng = native.Engine(0)
        eng.set_option("slots", slots)
        eng.set_option("sparse", sparse)
        data = b"".join(mine)
        offs = np.cumsum([0] + [len(c) for c in mine[:-1]]).astype(np.uint64) if mine else None
        if dedup:  # every rank de-duplicates its own shard
            d2, o2, w, _ = native.dedup_chunks(data, offs)
            eng.load_bytes(d2, o2, w)
        else:
            eng.load_bytes(data, offs)
        sh = GpuShard(eng, 0)
        sh.begin(nm, r, world)
        shards.append(sh)

    def allreduce(name, op):
        stack = torch.stack([getattr(sh, name) for sh in shards])
        red = stack.sum(0) if op == "sum" else stack.min(0).values
        for sh in shards:
            getattr(sh, name).copy_(red.to(getattr(sh, name).dtype))

    allreduce("table", "sum")
    for sh in shards:
        sh.table_ready()
    for i in range(nm):
        for sh in shards:
            sh.select(i)
        allreduce("key", "min")
        for sh in shards:
            sh.merge(i)
        allreduce("delta", "sum")
        for sh in shards:
            sh.apply(i)
    pairs, counts, lens = [], [], []
    for i in range(nm):
        recs = [sh.poll(i) for sh in shards]
        if recs[0][3] != 0:
            assert all(r[3] == recs[0][3] for r in recs)
            break
        assert all(r[0] == recs[0][0] and r[1] == recs[0][1] for r in recs)
        pairs.append(recs[0][0])
        counts.append(recs[0][1])
        lens.append(sum(r[2] for r in recs))
    for sh in shards:
        sh.end()
        sh.eng.close()
    return pairs, counts, lens


@pytest.mark.parametrize("world,slots,sparse", [(1, 1, 1), (2, 1, 1), (3, 1, 1), (2, 0, 1), (1, 2, 1), (2, 2, 1),
                                                (3, 2, 2), (2, 2, 2)])
def test_dp_lockstep_matches_oracle(native, world, slots, sparse):
    torch = pytest.importorskip("torch")
    chunks = _space_chunks(native.synth_text(1_500_000, 51))
    nm = 300
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp = oracle.train(data, nm, offs)
    got = _lockstep(native, chunks, nm, world, slots, sparse=sparse)
    assert got[0] == exp[0] and got[1] == exp[1] and got[2] == exp[2]


def test_dp_lockstep_ties_and_exhaustion(native):
    pytest.importorskip("torch")
    rng = np.random.default_rng(9)
    chunks = [b" " + bytes(97 + rng.integers(0, 3, size=rng.integers(1, 6))) for _ in range(3000)]
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp = oracle.train(data, 400, offs, raise_on_empty=False)
    for slots, sparse in ((1, 1), (2, 1), (2, 2)):
        got = _lockstep(native, chunks, 400, 3, slots, sparse=sparse)
        assert got[0] == exp[0] and got[1] == exp[1] and got[2] == exp[2], (slots, sparse)
    assert len(exp[0]) < 400  # the table ran empty: every rank stopped at the same merge


def test_dp_global_count_guard(native):
    """table entries are 32-bit, the first all-reduce carries the byte-pair counts as 16-bit limbs, so the
    sum over the ranks is exact -- and a job in which some pair occurs 2^32 times or more fails on EVERY rank at
    bpe_dp_table_ready (BPE_E_LIMIT), before any further collective.  A degenerate shard (one letter, 48 M pairs) as
    rank r of 128: summing 128 identical payloads is what the all-reduce of 128 such shards would deliver."""
    torch = pytest.importorskip("torch")
    from minbpe_amd.dist import GpuShard
    data = b"a" * 48_000_001
    ok_world, bad_world = 64, 128  # 64 x 48 M < 2^32 <= 128 x 48 M
    for world, fails in ((ok_world, False), (bad_world, True)):
        for rank in (0, world - 1):
            eng = native.Engine(0)
            try:
                eng.load_bytes(data)
                sh = GpuShard(eng, 0)
                sh.begin(4, rank, world)
                assert sh.table.numel() == 2 * 65536
                lo, hi = int(sh.table[97 * 256 + 97]), int(sh.table[65536 + 97 * 256 + 97])
                assert (hi << 16) + lo == 48_000_000
                sh.table.mul_(world)  # (the SUM over `world` identical shards: no limb sum reaches 2^31)
                assert int(sh.table.max()) < 2**31
                if fails:
                    with pytest.raises(RuntimeError, match="32-bit"):
                        sh.table_ready()
                else:
                    sh.table_ready()
                    sh.select(0)
                    torch.cuda.synchronize()
                sh.end()
            finally:
                eng.close()


def _chain_ranks(native, chunks, nm, world, opts=(), dedup=False):
    """bpe_dp_train_cb (the sharded loop of chain steps) on `world` ctxs of the one GPU we have, one thread per rank;
    the all-reduces are done on the host between barriers.  Returns every rank's result dict."""
    import threading
    import torch
    from minbpe_amd.dist import _DevicePtr, shard_chunks
    dev = torch.device("cuda", 0)
    engs = []
    for r in range(world):
        lo, hi = shard_chunks(len(chunks), r, world)
        mine = chunks[lo:hi]
        eng = native.Engine(0)
        for k, v in opts:
            eng.set_option(k, v)
        data = b"".join(mine)
        offs = np.cumsum([0] + [len(c) for c in mine[:-1]]).astype(np.uint64) if mine else None
        if dedup:
            d2, o2, w, _ = native.dedup_chunks(data, offs)
            eng.load_bytes(d2, o2, w)
        else:
            eng.load_bytes(data, offs)
        engs.append(eng)
    bar = threading.Barrier(world)
    slot = [None] * world
    calls = [0] * world

    def make(r):
        def allreduce(ptr, count, dtype, op, _stream):
            calls[r] += 1
            t = torch.as_tensor(_DevicePtr(ptr, count, "<i8" if dtype == 1 else "<i4"), device=dev)
            torch.cuda.synchronize()
            slot[r] = (t.cpu().numpy().copy(), count, dtype, op)
            bar.wait(timeout=120)
            assert all(s[1:] == slot[0][1:] for s in slot), [s[1:] for s in slot]  # the same collective on every rank
            stack = np.stack([s[0] for s in slot])
            red = stack.sum(0, dtype=stack.dtype) if op == 0 else stack.min(0)
            bar.wait(timeout=120)
            t.copy_(torch.from_numpy(red))
            torch.cuda.synchronize()
        return allreduce

    out, errs = [None] * world, [None] * world

    def run(r):
        try:
            out[r] = engs[r].dp_train_cb(nm, r, world, make(r))
        except ValueError as e:
            out[r] = engs[r].last_train
            errs[r] = e
        except BaseException as e:  # noqa: BLE001 (reported by the caller; the barrier must not strand the peers)
            errs[r] = e
            bar.abort()

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    stats = [e.train_stats() for e in engs]
    for e in engs:
        e.close()
    for e in errs:
        if e is not None and not isinstance(e, ValueError):
            raise e
    assert len(set(calls)) == 1, calls
    return out, errs, stats


@pytest.mark.parametrize("world", [1, 3])
def test_dp_dense_chain_steps_match_oracle(native, world):
    """the sharded loop's DENSE chain steps (the early merges of a big corpus: several pairs per sweep over every slot,
    ties left to the general path), forced onto a small corpus by a low lean_count; the switch to the indexed steps
    happens mid-run"""
    pytest.importorskip("torch")
    chunks = _space_chunks(native.synth_text(1_200_000, 57))
    nm = 500
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp = oracle.train(data, nm, offs)
    out, errs, stats = _chain_ranks(native, chunks, nm, world, (("lean_count", 1500),))
    assert not any(errs)
    for res in out:
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2]
    assert all(s["dense"] > 20 and s["sparse"] > 20 for s in stats), stats


@pytest.mark.parametrize("world", [1, 2, 3])
def test_dp_chain_steps_match_oracle(native, world):
    """the sharded loop of chain steps: GPT-like chunks spread over 1, 2 and 3 ranks -> the oracle's merges, counts
    and (summed) lengths on every rank, most of them done by chain steps"""
    pytest.importorskip("torch")
    chunks = _space_chunks(native.synth_text(1_500_000, 51))
    nm = 600
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp = oracle.train(data, nm, offs)
    out, errs, stats = _chain_ranks(native, chunks, nm, world)
    assert not any(errs)
    for res in out:
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2]
    assert all(s["steps"] > 0 and s["lean"] > nm // 2 for s in stats), stats


@pytest.mark.parametrize("opts", [(), (("dp_kcap", 1),), (("dp_kcap", 8), ("sparse", 2)), (("sparse", 0),)])
def test_dp_chain_steps_ties_and_exhaustion(native, opts):
    """three letters, thousands of short chunks on three ranks: nearly every selection is a tie whose pairs first
    occur on different ranks, a == b pairs head the list again and again, and the table runs empty before the last
    merge -- every rank stops at the same merge with the oracle's list"""
    pytest.importorskip("torch")
    rng = np.random.default_rng(9)
    chunks = [b" " + bytes(97 + rng.integers(0, 3, size=rng.integers(1, 6))) for _ in range(3000)]
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp = oracle.train(data, 400, offs, raise_on_empty=False)
    assert len(exp[0]) < 400
    out, errs, stats = _chain_ranks(native, chunks, 400, 3, opts)
    assert all(isinstance(e, ValueError) for e in errs)
    for res in out:
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2], opts


def test_dp_chain_steps_weighted_shards(native):
    """every rank de-duplicates its own shard (weights in the id words): same merges as the plain chunk list"""
    pytest.importorskip("torch")
    chunks = _space_chunks(native.synth_text(600_000, 53))
    nm = 300
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp = oracle.train(data, nm, offs)
    out, errs, _ = _chain_ranks(native, chunks, nm, 2, dedup=True)
    assert not any(errs)
    for res in out:
        assert res["pairs"] == exp[0] and res["counts"] == exp[1]


def test_dp_train_sharded_solo(native, engine):
    pytest.importorskip("torch")
    from minbpe_amd.dist import GpuShard, SoloComm, train_sharded
    text = native.synth_text(500_000, 52)
    data, offs = split_chunks(text.decode())
    exp = oracle.train(data, 300, offs)
    engine.load_bytes(data, offs)
    res = train_sharded(GpuShard(engine, 0), SoloComm(), 300)
    assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2]


def test_train_slotted_edge_cases(engine):
    """slotted loop: runs of one symbol (every merge is a == b: contiguous fallback each time),
    tiny streams, streams of exactly one / several tiles, chunk starts at slot boundaries."""
    cases = [(b"a" * 50000, None, 12), (b"ab" * 4096, None, 5), (b"abc" * 4096 + b"x", None, 20),
             (b"ab" * 30000, np.arange(0, 60000, 4096, dtype=np.uint64), 10),
             (b"", None, 3), (b"z", None, 2)]
    for data, offs, nm in cases:
        exp = oracle.train(data, nm, offs, raise_on_empty=False)
        engine.load_bytes(data, offs)
        if len(exp[0]) < nm:
            with pytest.raises(ValueError):
                engine.train(nm)
            res = engine.last_train
        else:
            res = engine.train(nm)
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2], (len(data), nm)


def test_fused_load_and_count_equals_the_three_passes(engine, native):
    """k_load_count (bytes -> id words + chunk starts + the first get_stats in one pass, the default) against
    k_widen + k_mark_starts + k_pair_count_bytes (option fuse_load = 0) and the oracle: stream lengths around the
    kernel's 4,096-position segments and its 4-byte groups, chunk starts on segment boundaries, empty chunks
    (repeated offsets), one chunk, no chunk list, a stream that is all chunk starts."""
    R = 4096
    rng = random.Random(5)
    text = native.synth_text(40 * R + 7, 77)
    cases = []
    for n in (2, 3, 5, 255, 256, 257, R - 1, R, R + 1, 2 * R, 2 * R + 3, 16 * R - 1, 16 * R, 16 * R + 2, 40 * R + 7):
        data = text[:n]
        cases.append((data, None))
        cuts = sorted(set([0] + [rng.randrange(n) for _ in range(max(n // 6, 1))]))
        cases.append((data, np.array(cuts, dtype=np.uint64)))
    data = text[:2 * R + 3]
    cases.append((data, np.array([0, 1, 2, R - 1, R, R, R, R + 1, 2 * R - 1, 2 * R, 2 * R + 2], dtype=np.uint64)))
    cases.append((data, np.array([0], dtype=np.uint64)))
    cases.append((data[:5000], np.arange(0, 5000, dtype=np.uint64)))  # every position starts a chunk: no pair at all
    for data, offs in cases:
        # the stream the pass leaves (train(0) merges nothing): the bytes as ids, a chunk start wherever an offset points
        want_starts = sorted(set(int(o) for o in (offs if offs is not None else [0]) if o < len(data)))
        for fuse in (1, 0):
            engine.set_option("fuse_load", fuse)
            try:
                engine.load_bytes(data, offs)
                engine.train(0)
                assert engine.read_ids().tolist() == list(data), (len(data), fuse)
                assert engine.read_chunk_starts().tolist() == want_starts, (len(data), fuse)
            finally:
                engine.set_option("fuse_load", 1)
        nm = 12
        exp = oracle.train(data, nm, offs, raise_on_empty=False)
        got = []
        for fuse in (1, 0):
            engine.set_option("fuse_load", fuse)
            try:
                engine.load_bytes(data, offs)
                if len(exp[0]) < nm:
                    with pytest.raises(ValueError):
                        engine.train(nm)
                    res = engine.last_train
                else:
                    res = engine.train(nm)
                # (the resident stream is compared after a train() that ran to the end)
                full = len(exp[0]) == nm
                got.append((res["pairs"], res["counts"], res["lens"], engine.read_ids().tolist() if full else None,
                            engine.read_chunk_starts().tolist() if full else None))
            finally:
                engine.set_option("fuse_load", 1)
        assert got[0] == got[1], (len(data), None if offs is None else len(offs))
        assert (got[0][0], got[0][1], got[0][2]) == (exp[0], exp[1], exp[2]), (len(data), None if offs is None else len(offs))


def test_train_fused_row_maxima_option(engine, native):
    """option fused_rows=1: row maxima recomputed by extra blocks of the table-update launch."""
    data = native.synth_text(400_000, 61)
    exp = oracle.train(data, 200)
    engine.set_option("fused_rows", 1)
    engine.set_option("slots", 1)  # (an option of the first slotted form's table update)
    try:
        engine.load_bytes(data)
        res = engine.train(200)
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2]
    finally:
        engine.set_option("fused_rows", 0)
        engine.set_option("slots", 2)


def test_train_invariant_at_scale(engine, native):
    """size-independent property (used at sizes the oracle cannot reach): every a != b merge
    shortens the stream by exactly the count the pair table reported for it."""
    data = native.synth_text(20_000_000, 71)
    engine.load_bytes(data)
    res = engine.train(300)
    lens = np.array([len(data)] + res["lens"], dtype=np.int64)
    cnt = np.array(res["counts"], dtype=np.int64)
    same = np.array([a == b for a, b in res["pairs"]])
    removed = lens[:-1] - lens[1:]
    assert np.all(removed[~same] == cnt[~same])
    assert np.all(removed[same] <= cnt[same]) and np.all(removed > 0)
    # and the first merges agree with the oracle on the full 20 MB stream
    exp = oracle.train(data, 3)
    assert res["pairs"][:3] == exp[0] and res["counts"][:3] == exp[1] and res["lens"][:3] == exp[2]


def test_dp_native_rccl_solo(native):
    """bpe_dp_train: the sharded loop inside the library, RCCL called directly (world of one here)."""
    eng = native.Engine(0)
    try:
        text = native.synth_text(800_000, 81)
        data, offs = split_chunks(text.decode())
        exp = oracle.train(data, 250, offs)
        eng.load_bytes(data, offs)
        eng.comm_init(0, 1, native.Engine.comm_unique_id())
        eng.set_option("dp_force_comm", 1)  # (a world of one skips its collectives unless told otherwise)
        res = eng.dp_train(250)
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2]
        eng.set_option("dp_force_comm", 0)
        res = eng.dp_train(250)
        assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2]
        eng.set_option("dp_force_comm", 1)
        # exhaustion through the same path
        eng.load_bytes(b"ab", None)
        with pytest.raises(ValueError):
            eng.dp_train(5)
        assert eng.last_train["pairs"] == [(97, 98)]
        # and the ctx still trains normally afterwards
        eng.load_bytes(data, offs)
        assert eng.train(50)["pairs"] == exp[0][:50]
    finally:
        eng.close()


# ---------------------------------------------------------------------------
# batch decode (N4): b"".join(vocab[idx] for idx in ids) on the device

def test_decode_batch_engine_level(engine, native):
    rng = np.random.default_rng(5)
    V = 1000
    toks = [bytes(rng.integers(0, 256, size=int(L), dtype=np.uint8)) for L in rng.integers(0, 12, size=V)]
    toks[3] = b""
    toks[V - 1] = b"\xff" * 40
    lens = np.array([len(t) for t in toks], dtype=np.int64)
    offs = np.zeros(V + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    engine.decode_set_vocab(b"".join(toks), offs)
    for n in (0, 1, 63, 64, 4096, 4097, 300_000):
        ids = rng.integers(0, V, size=n, dtype=np.int32)
        want = b"".join(toks[i] for i in ids)
        assert engine.decode_batch(ids) == want
        pos = np.unique(np.concatenate([[0, n], rng.integers(0, n + 1, size=50)])).astype(np.uint64)
        got, boff = engine.decode_batch(ids, pos)
        cum = np.concatenate([[0], np.cumsum(lens[ids])]).astype(np.int64)
        assert got == want
        assert boff.tolist() == cum[pos.astype(np.int64)].tolist()
    # ids outside the table: the FIRST offending position is reported, nothing is returned
    ids = rng.integers(0, V, size=10_000, dtype=np.int32)
    ids[7000] = V
    ids[400] = -1
    with pytest.raises(native.InvalidToken) as ei:
        engine.decode_batch(ids)
    assert ei.value.args[0] == 400
    ids[400] = 0
    with pytest.raises(native.InvalidToken) as ei:
        engine.decode_batch(ids)
    assert ei.value.args[0] == 7000
    # an empty table decodes nothing but the empty batch
    engine.decode_set_vocab(b"", np.zeros(1, np.uint64))
    assert engine.decode_batch(np.empty(0, np.int32)) == b""
    with pytest.raises(native.InvalidToken):
        engine.decode_batch(np.zeros(3, np.int32))


def test_decode_batch_tokenizers(native):
    from minbpe_amd import BasicTokenizer, RegexTokenizer
    text = native.synth_text(2_000_000, 77).decode()
    tok = RegexTokenizer()
    tok.train(text[:300_000], 256 + 300)
    tok.register_special_tokens({"<|endoftext|>": 100257})
    doc = text + "<|endoftext|>" + text[:1000]
    ids = tok.encode(doc, allowed_special="all")
    raw = tok.decode_batch(ids)
    assert raw == doc.encode("utf-8")
    assert raw.decode("utf-8", errors="replace") == tok.decode(ids)
    cut = ids.index(100257)
    raw2, boff = tok.decode_batch(ids, [0, cut, cut + 1, len(ids)])
    assert raw2 == raw
    assert boff.tolist() == [0, len(text.encode()), len(text.encode()) + len("<|endoftext|>"), len(raw)]
    with pytest.raises(ValueError, match="invalid token id: 90000"):
        tok.decode_batch([65, 90000, 90001])
    assert tok.decode_batch([]) == b""
    b = BasicTokenizer()
    b.train(text[:100_000], 256 + 50)
    part = text[:200_000]
    assert b.decode_batch(np.array(b.encode(part))) == part.encode("utf-8")
    with pytest.raises(KeyError):
        b.decode_batch([1, 4000])
    # the two tokenizers share one engine: each call re-installs its own table when needed
    assert tok.decode_batch(ids[:100]) == b"".join(tok.vocab[i] for i in ids[:100])


def test_encode_decode_round_trip_20mb(native):
    """size-independent property at a bench-sized batch: decode(encode(x)) == x"""
    from minbpe_amd import RegexTokenizer
    text = native.synth_text(20_000_000, 78).decode()
    tok = RegexTokenizer()
    tok.train(text[:1_000_000], 256 + 1000)
    data, offs = tok._chunked(text)
    ids, out_off = tok._encode_flat(data, offs)
    assert len(out_off) == len(offs) + 1 and int(out_off[-1]) == len(ids) < len(data)
    raw, boff = tok.decode_batch(ids, out_off)
    assert raw == data
    assert np.array_equal(boff[:-1], offs) and int(boff[-1]) == len(data)


# ---------------------------------------------------------------------------
# weighted chunks (N1): the distinct chunks, weighted, train like the full chunk list

def _weighted_cases(native):
    rng = np.random.default_rng(12)
    yield native.synth_text(600_000, 71).decode(), 300
    # tiny alphabet: ties at the maximum, a == b runs, multiplicities with many set bits; runs empty
    words = ["".join("ab"[int(x)] for x in rng.integers(0, 2, size=int(L))) for L in rng.integers(1, 9, size=60)]
    yield " ".join(words[int(i)] for i in rng.integers(0, len(words), size=150_000)), 120
    # runs of one symbol longer than a slot (4096 ids), repeated so that they carry weight
    yield ("a" * 9001 + " ") * 3 + "aaaaaaa " * 5000 + "aaaa bbbb abab " * 3000 + ("b" * 4097 + " ") * 6, 40


def _train_or_partial(eng, nm):
    try:
        return eng.train(nm)
    except ValueError:
        return eng.last_train


@pytest.mark.parametrize("opts", [{}, {"slots": 0}, {"mode": 0}, {"merge": 1}])
def test_weighted_training_matches_full_list(native, opts):
    eng = native.Engine(0)
    for name, value in opts.items():
        eng.set_option(name, value)
    for text, nm in _weighted_cases(native):
        data, offs = split_chunks(text)
        exp_pairs, exp_counts, _ = oracle.train(data, nm, offs, raise_on_empty=False)
        d2, o2, w, nd = native.dedup_chunks(data, offs)
        assert len(o2) < len(offs) and int(w.max()) >= 2
        eng.load_bytes(d2, o2, w)
        got = _train_or_partial(eng, nm)
        assert got["pairs"] == exp_pairs and got["counts"] == exp_counts
        # same engine, unweighted again: no state leaks from the weighted run
        eng.load_bytes(data, offs)
        got = _train_or_partial(eng, nm)
        assert got["pairs"] == exp_pairs and got["counts"] == exp_counts
    eng.close()


def test_weighted_single_steps(engine, native):
    # bpe_get_stats / bpe_argmax on a weighted stream: weighted counts, first-appearance order
    text = "the cat the dog the end of the cat and the dog " * 7
    data, offs = split_chunks(text)
    d2, o2, w, _ = native.dedup_chunks(data, offs)
    engine.load_bytes(data, offs)
    full = engine.get_stats()
    full_arg = engine.argmax()
    engine.load_bytes(d2, o2, w)
    got = engine.get_stats()
    assert [(p, c) for p, c, _ in got] == [(p, c) for p, c, _ in full]
    assert engine.argmax() == full_arg


def test_weighted_sharded_lockstep(native):
    pytest.importorskip("torch")
    chunks = _space_chunks(native.synth_text(1_200_000, 53))
    nm = 200
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp = oracle.train(data, nm, offs)
    got = _lockstep(native, chunks, nm, 3, dedup=True)
    assert got[0] == exp[0] and got[1] == exp[1]


def test_encode_ordinary_batch_and_back(native):
    """cfg5 at the drop-in level: many documents -> one encode batch -> one decode batch."""
    from minbpe_amd import RegexTokenizer
    from minbpe_amd.tokenizer import GPT2_SPLIT_PATTERN
    text = native.synth_text(1_500_000, 91).decode()
    docs = [d for d in text.split("\n\n")] + ["", " ", "don't  stop\r\n", "x"]
    for pattern in (None, GPT2_SPLIT_PATTERN, r"\p{L}+|\p{N}+|[^\p{L}\p{N}]+"):
        tok = RegexTokenizer(pattern)
        tok.train(text[:300_000], 256 + 400)
        ids, doff = tok.encode_ordinary_batch(docs)
        assert len(doff) == len(docs) + 1 and int(doff[-1]) == len(ids)
        for d in list(range(0, len(docs), 97)) + list(range(len(docs) - 4, len(docs))):
            assert ids[int(doff[d]):int(doff[d + 1])].tolist() == tok.encode_ordinary(docs[d]), d
        raw, boff = tok.decode_batch(ids, doff)
        assert [raw[int(boff[d]):int(boff[d + 1])] for d in range(len(docs))] == [d.encode() for d in docs]
    ids, doff = tok.encode_ordinary_batch([])
    assert len(ids) == 0 and doff.tolist() == [0]
    ids, doff = tok.encode_ordinary_batch(["", ""])
    assert len(ids) == 0 and doff.tolist() == [0, 0, 0]


def test_gpt4_tokenizer_on_a_toy_rank_table(native):
    """GPT4Tokenizer through the device (byte shuffle + merge ids that are ranks, gpt4.py:57-130);
    cl100k_base itself is not available offline, so the rank table is built from a tokenizer
    trained here."""
    from minbpe_amd import GPT4Tokenizer, RegexTokenizer
    text = native.synth_text(300_000, 79).decode()
    base = RegexTokenizer()
    base.train(text, 256 + 400)
    perm, ranks = toy_rank_table(base, 6)
    g = GPT4Tokenizer(ranks)
    probe = text[:50_000] + " don't  stop 12345 ünïcödé 😉"
    want = [perm[i] if i < 256 else i for i in base.encode_ordinary(probe)]
    assert g.encode_ordinary(probe) == want
    assert g.decode(want) == probe and g.decode_batch(want) == probe.encode("utf-8")
    ids = g.encode("<|endoftext|>" + probe[:200], allowed_special="all")
    assert ids[0] == 100257 and ids[1:] == g.encode_ordinary(probe[:200])


@pytest.mark.parametrize("k1", [1, 3])
def test_recount_histogram_variants_vs_oracle(native, k1):
    """the literal path (mode=0: get_stats of EVERY iteration, base.py:13-22) with each general histogram kernel --
    k_pair_count_lds (1), k_pair_count_h32 (3): merges and counts =
    the oracle's, on a GPT-4-split text (chunk flags at every few ids), on one stream without chunks (span ends fall
    anywhere) and on a two-letter text whose counts pass 2^14 in every workgroup's table (the drain)"""
    eng = native.Engine(0)
    eng.set_option("mode", 0)
    eng.set_option("k1", k1)
    text = native.synth_text(3_000_000, 17).decode()
    data, offs = split_chunks(text)
    exp_pairs, exp_counts, _ = oracle.train(data, 48, offs)
    eng.load_bytes(data, offs)
    got = eng.train(48)
    assert got["pairs"] == exp_pairs and got["counts"] == exp_counts
    raw = native.synth_text(5_000_003, 18)
    exp_pairs, exp_counts, _ = oracle.train(raw, 24, None)
    eng.load_bytes(raw)
    got = eng.train(24)
    assert got["pairs"] == exp_pairs and got["counts"] == exp_counts
    rng = np.random.default_rng(19)
    ab = rng.integers(97, 99, size=40_000_000, dtype=np.uint8).tobytes()
    exp_pairs, exp_counts, _ = oracle.train(ab, 6, None)
    eng.load_bytes(ab)
    got = eng.train(6)
    assert got["pairs"] == exp_pairs and got["counts"] == exp_counts
    eng.close()


def test_load_bytes_refuses_bad_offsets_wherever_they_are(engine):
    """bpe_load_bytes checks the chunk offsets on the host before anything is uploaded (they must ascend and stay <= n) --
    with up to 16 threads once there are 2^22 of them: the FIRST bad index is named, as by one thread, whichever segment
    it falls in and also when several segments hold one; a good list of that size loads."""
    n_chunks = (1 << 22) + 12345
    data = np.full(2 * n_chunks, 97, np.uint8).tobytes()
    good = (np.arange(n_chunks, dtype=np.uint64) * 2)
    engine.load_bytes(data, good)
    assert len(engine) == len(data)
    for where in ([7], [n_chunks // 2 + 3], [n_chunks - 1], [n_chunks // 3, 2 * n_chunks // 3 + 1, n_chunks - 5]):
        bad = good.copy()
        for w in where:
            bad[w] = bad[w - 1] - 1  # (descends at w)
        with pytest.raises(ValueError, match=rf"chunk_offsets\[{where[0]}\] = .*ascend"):
            engine.load_bytes(data, bad)
    bad = good.copy()
    bad[n_chunks - 2] = len(data) + 1  # beyond the text
    with pytest.raises(ValueError, match=rf"chunk_offsets\[{n_chunks - 2}\]"):
        engine.load_bytes(data, bad)
    engine.load_bytes(data, good)
