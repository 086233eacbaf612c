"""GPU: full-length parity at BASELINE sizes.

* cfg1: the reference's own fixture text (tests/golden/taylorswift.txt, a verbatim copy of
  /root/reference/tests/taylorswift.txt, sha256-pinned) through the drop-in classes, against
  the hashes the reference itself produced (tests/golden/golden.json["taylorswift"]).
* cfg2 and a cfg3-shaped run: every merge (pair, count, stream length) against digests the CPU
  oracle produced once, offline (tests/golden/gen_big_golden.py -> big_golden.json).
Integer work: every comparison is exact."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import checkpoint_digests, first_divergence

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def big_golden():
    with open(os.path.join(HERE, "golden", "big_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def taylor():
    with open(os.path.join(HERE, "golden", "taylorswift.txt"), encoding="utf-8") as f:
        return f.read()


def h16(obj):
    return hashlib.sha256(repr(obj).encode()).hexdigest()[:16]


@pytest.mark.parametrize("kind", ["basic", "regex"])
def test_taylorswift_vocab512_reference_hashes(golden, taylor, kind):
    """configs[0]: train.py's workload (train.py:20-31), hashes from the live reference."""
    from minbpe_amd import BasicTokenizer, RegexTokenizer
    g = golden["taylorswift"]
    assert hashlib.sha256(taylor.encode("utf-8")).hexdigest()[:16] == g["sha256_prefix"]
    tok = BasicTokenizer() if kind == "basic" else RegexTokenizer()
    tok.train(taylor, 512)
    assert h16(list(tok.merges.items())) == g[f"{kind}512_merges_hash"]
    ids = tok.encode(taylor)
    assert len(ids) == g[f"{kind}512_encode_len"]
    assert h16(ids) == g[f"{kind}512_encode_hash"]
    assert tok.decode(ids) == taylor


def _check_digests(res, g):
    assert len(res["pairs"]) == g["done"]
    got = checkpoint_digests(res["pairs"], res["counts"], res["lens"], g["step"])
    bad = first_divergence(got, g["digests"])
    assert bad is None, f"first differing checkpoint: merge {bad}"
    assert got[-1] == g["digests"][-1]
    assert res["lens"][-1] == g["final_len"]


def test_cfg2_all_3840_merges_equal_oracle(native, engine, big_golden):
    """configs[1] at full length: BasicTokenizer.train, 100 MB, vocab 4096."""
    g = big_golden["cfg2"]
    data = native.synth_text(g["bytes"], g["seed"])
    assert hashlib.sha256(data).hexdigest() == g["data_sha256"]
    engine.load_bytes(data)
    _check_digests(engine.train(g["merges"]), g)


def test_cfg3_shape_gpt4_split_all_merges_equal_oracle(native, engine, big_golden):
    """configs[2] shape: GPT-4 split, 150 MB, 8192 merges; chunk offsets (produced by the `regex`
    module when the golden was made) must be reproduced by the native splitter first."""
    g = big_golden["cfg3s"]
    data = native.synth_text(g["bytes"], g["seed"])
    assert hashlib.sha256(data).hexdigest() == g["data_sha256"]
    offs = native.split_offsets(data, 4)
    assert len(offs) == g["n_chunks"]
    assert hashlib.sha256(np.ascontiguousarray(offs, dtype=np.uint64).tobytes()).hexdigest() == g["offsets_sha256"]
    engine.load_bytes(data, offs)
    _check_digests(engine.train(g["merges"]), g)


# (sparse, lean): the default engine; every a != b pass through the index (k_rowsel_lean breaks the ties, or
# defers them); the general path alone, without and with the index
# (lean 3: lean iterations that select from the whole row-maxima array, k_rowsel_lean, not from the table
# update's per-wave records)
# (lean 4: every lean iteration selects, no chained merges; lean 5: a == b passes over every slot + index rebuild;
# lean 6: no general-path stretches after clustered deferrals -- every tie the lean selection cannot settle is one)
# lean 7: chain steps (k_chain.hip) forced onto every merge; lean 8: the default engine without chain steps
# lean 9: the default engine with the stream re-packed into 256-id slots at the first index build (option small_slots = 2)
FULL_VARIANTS = [(1, 1), (2, 1), (1, 0), (2, 0), (1, 3), (1, 4), (1, 5), (1, 6), (1, 7), (2, 7), (1, 8), (1, 9), (2, 9)]


@pytest.mark.parametrize("sparse,lean", FULL_VARIANTS)
@pytest.mark.parametrize("name", ["full16r", "full12b", "full8r"])
def test_whole_vocab_range_all_31744_merges_equal_oracle(native, engine, big_golden, name, sparse, lean):
    """The headline's whole vocabulary range (vocab 32000 = 31,744 merges) on inputs the oracle finishes in
    half an hour: 16 MB with the GPT-4 split (regex.py:49-63) and 12 MB as one stream (basic.py:31-42).
    Counts fall to 2-3 here: hundreds of pairs tied at the maximum (more than TIE_CAP), rows with several
    maximal columns, the vocabulary beyond 8448 -- every select / table path the 1 GB run takes late."""
    g = big_golden[name]
    data = native.synth_text(g["bytes"], g["seed"])
    assert hashlib.sha256(data).hexdigest() == g["data_sha256"]
    offs = None
    if g["chunked"]:
        offs = native.split_offsets(data, 4)
        assert hashlib.sha256(np.ascontiguousarray(offs, dtype=np.uint64).tobytes()).hexdigest() == g["offsets_sha256"]
    engine.set_option("sparse", sparse)
    engine.set_option("lean", 2 if lean == 7 else (1 if lean >= 3 else lean))
    engine.set_option("chain", 1 if lean in (1, 7, 9) else 0)
    engine.set_option("small_slots", 2 if lean == 9 else 1)
    engine.set_option("lean_sum", 0 if lean == 3 else 1)
    engine.set_option("lean_chain", 0 if lean == 4 else 1)
    engine.set_option("aa_sparse", 0 if lean == 5 else 1)
    engine.set_option("lean_backoff", 0 if lean in (6, 7) else 1)
    try:
        engine.load_bytes(data, offs)
        _check_digests(engine.train(g["merges"]), g)
        st = engine.train_stats()
        if (sparse, lean) == (1, 1):
            # counts of 2-3 late in these runs: hundreds of tied pairs per selection, which a lean selection hands
            # back to the general path (~200 us each).  The default engine must notice and stay on the general
            # path for such stretches instead of deferring merge after merge.
            assert st["lean"] > 1000 and st["deferred"] <= 400, st
    finally:
        engine.set_option("small_slots", 1)
        engine.set_option("lean_backoff", 1)
        engine.set_option("chain", 1)
        engine.set_option("sparse", 1)
        engine.set_option("lean", 1)
        engine.set_option("lean_sum", 1)
        engine.set_option("lean_chain", 1)
        engine.set_option("aa_sparse", 1)


def test_cross_mode_large_chunked(native, engine):
    """The literal reference loop (mode 0: clear, get_stats, arg-max, merge every iteration) and the
    default engine (pair table kept current by the merge pass) must agree merge for merge on a stream
    far larger than the oracle-sized tests: 400 MB, GPT-4 split, first 600 merges."""
    data = native.synth_text(400_000_000, 7)
    offs = native.split_offsets(data, 4)
    engine.load_bytes(data, offs)
    fast = engine.train(600)
    engine.set_option("mode", 0)
    try:
        slow = engine.train(600)
    finally:
        engine.set_option("mode", 1)
    assert slow["pairs"] == fast["pairs"]
    assert slow["counts"] == fast["counts"] and slow["lens"] == fast["lens"]


@pytest.mark.parametrize("pinned", [1, 0])
def test_large_buffers_survive_the_staged_copies(engine, native, pinned):
    """Caller buffers of hundreds of MB go to the device (and ids come back) through a ring of pinned staging buffers filled
    by several host threads (api_ctx.hip: upload_h2d / download_d2h; option pinned_upload): text and chunk offsets are two
    uploads back to back (the second must not touch a buffer the first one's tail is still being copied out of), the ids
    of a batch encode one download -- every byte and every offset must arrive, with the option on and off."""
    import oracle
    from minbpe_amd import _native
    data = native.synth_text(180_000_000, 41)
    offs = _native.split_offsets(data, 4)
    assert len(offs) * 8 > 128 << 20  # (both buffers take the ring, several times around)
    engine.set_option("pinned_upload", pinned)
    try:
        engine.load_bytes(data, offs)
        ids = engine.read_ids()
        assert len(ids) == len(data) and np.array_equal(ids.astype(np.uint8), np.frombuffer(data, dtype=np.uint8))
        assert np.array_equal(np.asarray(engine.read_chunk_starts(), dtype=np.uint64), np.asarray(offs, dtype=np.uint64))
        # one merge list, the whole text as a batch of GPT-4-split chunks: 180 MB up, 1.4 GB of offsets up, ~0.5 GB of ids down
        pairs = oracle.train_fast(data[:2_000_000], 300, _native.split_offsets(data[:2_000_000], 4))[0]
        got, got_off = engine.encode_batch(np.array(pairs, np.int32), None, data, offs)
        exp, exp_off = oracle.encode(pairs, data, offs)
        assert np.array_equal(got, exp) and np.array_equal(np.asarray(got_off, dtype=np.uint64), np.asarray(exp_off, dtype=np.uint64))
    finally:
        engine.set_option("pinned_upload", 1)
