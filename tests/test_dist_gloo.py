"""CPU, world_size 2, gloo: the data-parallel training protocol of
minbpe_amd/dist.py (sharding by chunks, SUM all-reduce of the table deltas, MIN
all-reduce of the tie-break key, lock-step stop on exhaustion) against the
single-process oracle.  The per-rank engine is tests/cpu_shard.py (numpy); the
GPU kernels behind the same protocol are covered by test_gpu_parity.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, chunks, num_merges, out_q, dedup=False, fail=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from minbpe_amd.dist import TorchComm, shard_chunks, train_sharded
        from cpu_shard import CpuShard
        lo, hi = shard_chunks(len(chunks), rank, world)
        mine = chunks[lo:hi]
        data = b"".join(mine)
        offs = np.cumsum([0] + [len(c) for c in mine[:-1]]).astype(np.uint64) if mine else None
        wexp = None
        if dedup and mine:  # every rank folds the repeats of ITS shard into weights
            from minbpe_amd import _native
            data, offs, wexp, _ = _native.dedup_chunks(data, offs)
        try:
            fail_at = fail[1] if fail and fail[0] == rank else None
            res = train_sharded(CpuShard(data, offs, wexp, fail_at=fail_at), TorchComm(), num_merges, depth=3)
            out_q.put((rank, "ok", res["pairs"], res["counts"], res["lens"]))
        except RuntimeError as e:
            out_q.put((rank, "failed", str(e), None, None))
        except ValueError as e:
            out_q.put((rank, "empty", e.partial["pairs"], e.partial["counts"], e.partial["lens"]))
    finally:
        dist.destroy_process_group()


def _run(chunks, num_merges, world=2, dedup=False, fail=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, chunks, num_merges, q, dedup, fail))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(outs)


def _chunks(text: bytes):
    # split before every space: " word" chunks, like the GPT patterns' common case
    import re
    return [c for c in re.findall(rb" ?[^ ]+| +", text) if c]


def _oracle(chunks, num_merges):
    import oracle
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    return oracle.train(data, num_merges, offs, raise_on_empty=False)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_training_matches_single_process(native, world):
    text = native.synth_text(6000, 41)
    chunks = _chunks(text)
    exp = _oracle(chunks, 60)
    outs = _run(chunks, 60, world)
    for rank, status, pairs, counts, lens in outs:
        assert status == "ok"
        assert pairs == exp[0] and counts == exp[1] and lens == exp[2], f"rank {rank}"


def test_sharded_ties_prefer_lowest_rank_then_position(native):
    # tie-heavy: a tiny alphabet, so first-occurrence order across shards decides most merges
    rng = np.random.default_rng(9)
    words = [bytes(97 + rng.integers(0, 3, size=rng.integers(1, 6))) for _ in range(400)]
    chunks = [b" " + w for w in words]
    exp = _oracle(chunks, 25)
    for rank, status, pairs, counts, lens in _run(chunks, 25):
        assert status == "ok" and pairs == exp[0] and counts == exp[1] and lens == exp[2]


def test_sharded_training_with_per_rank_dedup(native):
    # tie-heavy and repetitive: weights with several set bits on every rank
    rng = np.random.default_rng(10)
    words = [bytes(97 + rng.integers(0, 3, size=rng.integers(1, 5))) for _ in range(30)]
    chunks = [b" " + words[int(i)] for i in rng.integers(0, len(words), size=1500)]
    exp = _oracle(chunks, 30)
    for rank, status, pairs, counts, lens in _run(chunks, 30, world=3, dedup=True):
        assert pairs == exp[0] and counts == exp[1]  # lens refer to the de-duplicated shards


@pytest.mark.parametrize("bad_rank", [0, 2])
def test_rank_local_failure_stops_every_rank_without_deadlock(native, bad_rank):
    """One rank's shard fails inside merge() of iteration 7 (as a timed-out device-side wait would):
    no rank may be left blocked in an all-reduce, and every rank must report the failure."""
    chunks = _chunks(native.synth_text(4000, 44))
    outs = _run(chunks, 30, world=3, fail=(bad_rank, 7))
    assert [o[1] for o in outs] == ["failed"] * 3
    assert all("failed with status" in o[2] for o in outs)


def test_sharded_exhaustion_stops_all_ranks_together():
    chunks = [b"ab", b"ab", b"cd", b"ab"]
    exp = _oracle(chunks, 6)
    assert len(exp[0]) < 6
    for rank, status, pairs, counts, lens in _run(chunks, 6):
        assert status == "empty" and pairs == exp[0] and lens == exp[2]


# ---------------------------------------------------------------------------
# init_native_comm: every rank must come back with the same answer, and none may be left
# waiting in a collective, whichever rank fails to set the library's communicator up

class _FakeEngine:
    def __init__(self, rank, fail_uid=False, fail_init_on=None, no_lib_on=None):
        self.rank, self.fail_uid, self.fail_init_on = rank, fail_uid, fail_init_on
        self.no_lib_on = no_lib_on
        self.inited = None

    def comm_available(self):
        return self.no_lib_on != self.rank

    def comm_unique_id(self):
        if self.fail_uid:
            raise RuntimeError("librccl not found")
        return bytes(range(1, 129))

    def comm_init(self, rank, world, uid):
        if self.fail_init_on == rank:
            raise RuntimeError("ncclCommInitRank failed")
        self.inited = (rank, world, uid)


def _comm_worker(rank, world, port, mode, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from minbpe_amd.dist import TorchComm, init_native_comm
        eng = _FakeEngine(rank, fail_uid=(mode == "uid"), fail_init_on=(1 if mode == "init" else None),
                          no_lib_on=(1 if mode == "nolib" else None))
        ok = init_native_comm(eng, TorchComm())
        out_q.put((rank, ok, eng.inited))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ok", "uid", "init", "nolib"])
def test_native_comm_setup_is_all_or_nothing(mode):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [o[1] for o in outs] == [mode == "ok"] * world
    if mode == "ok":  # every rank got rank 0's id
        assert all(o[2] == (o[0], world, bytes(range(1, 129))) for o in outs)
    if mode == "nolib":  # one rank without librccl: NO rank may have entered comm_init
        assert all(o[2] is None for o in outs)


# ---------------------------------------------------------------------------
# train_tokenizer: the drop-in class trained over a corpus spread across ranks

def _tok_worker(rank, world, port, parts, vocab_size, dedup, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from minbpe_amd import RegexTokenizer
        from minbpe_amd.dist import train_tokenizer
        from cpu_shard import CpuShard
        tok = RegexTokenizer()
        tok.dedup = dedup
        try:
            train_tokenizer(tok, parts[rank], vocab_size, make_shard=CpuShard)
            out_q.put((rank, "ok", list(tok.merges.items()), tok.vocab[max(tok.vocab)]))
        except ValueError:
            out_q.put((rank, "empty", list(tok.merges.items()), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dedup", [True, False])
def test_train_tokenizer_over_ranks_equals_single_process(native, dedup):
    import regex as re
    from minbpe_amd.tokenizer import GPT4_SPLIT_PATTERN
    text = native.synth_text(9000, 43).decode()
    docs = text.split("\n\n")
    cut = len(docs) // 2
    parts = ["\n\n".join(docs[:cut]) + "\n\n", "\n\n".join(docs[cut:])]  # a document boundary is a chunk boundary
    whole = parts[0] + parts[1]
    chunks = [c.encode() for c in re.findall(GPT4_SPLIT_PATTERN, whole)]
    assert chunks == [c.encode() for p in parts for c in re.findall(GPT4_SPLIT_PATTERN, p)]
    exp = _oracle(chunks, 50)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tok_worker, args=(r, world, port, parts, 256 + 50, dedup, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, status, merges, last_tok in outs:
        assert status == "ok"
        assert [m[0] for m in merges] == exp[0] and [m[1] for m in merges] == list(range(256, 306))
    assert outs[0][3] == outs[1][3]


# ---------------------------------------------------------------------------------------------------------------------
# The sharded CHAIN STEP (api_rccl.hip: dp_train_loop; k_pool_sel / k_pool_sel_dp, k_dp_fold_chain, k_apply_chain) on the
# CPU model of tests/cpu_chain_shard.py, which speaks the device's payload layouts: the first exchange as 16-bit limbs,
# the MIN payload of a selection (rank << 33 | first local position per pool entry to locate), the SUM payload of a batch
# (format-B vectors per pair, adj words, the status word at tail[16]).

def _worker_chain(rank, world, port, chunks, num_merges, out_q, kcap, capacity, depth, fail_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from minbpe_amd.dist import TorchComm, shard_chunks
        from cpu_chain_shard import CpuChainShard, train_chain_sharded
        lo, hi = shard_chunks(len(chunks), rank, world)
        mine = chunks[lo:hi]
        data = b"".join(mine)
        offs = np.cumsum([0] + [len(c) for c in mine[:-1]]).astype(np.uint64) if mine else None
        shard = CpuChainShard(data, offs, kcap=kcap, capacity=capacity, depth=depth)
        if fail_rank == rank:  # a rank-local failure inside the third step's merge pass
            orig, calls = shard.merge_batch, [0]

            def failing(batch):
                calls[0] += 1
                if calls[0] == 3:
                    shard._status = -7
                orig(batch)
            shard.merge_batch = failing
        res = train_chain_sharded(shard, TorchComm(), num_merges)
        out_q.put((rank, res["status"], res["pairs"], res["counts"], (res["steps"], res["generals"])))
    finally:
        dist.destroy_process_group()


def _run_chain(chunks, num_merges, world, kcap=8, capacity=24, depth=4, fail_rank=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chain, args=(r, world, port, chunks, num_merges, q, kcap, capacity, depth, fail_rank))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(outs)


@pytest.mark.parametrize("world,kcap", [(2, 8), (3, 15), (2, 2)])
def test_sharded_chain_steps_match_single_process(native, world, kcap):
    """GPT-like chunks: most merges come off the pool in batches; every rank reports the oracle's merges and counts."""
    chunks = _chunks(native.synth_text(9000, 45))
    exp = _oracle(chunks, 90)
    outs = _run_chain(chunks, 90, world, kcap=kcap)
    for rank, status, pairs, counts, (steps, generals) in outs:
        assert status == 0 and pairs == exp[0] and counts == exp[1], f"rank {rank}"
        assert steps < len(pairs) or kcap == 2  # (batches of several merges)
    assert len({o[4] for o in outs}) == 1  # every rank ran the same units


def test_sharded_chain_steps_ties_across_ranks_and_exhaustion(native):
    """A three-letter corpus: nearly every level is a tie whose pairs first occur on different ranks (the MIN payload
    orders them: lowest rank, then position), a == b pairs head the pool again and again (the general path's merges), and
    the table runs empty before the last merge -- every rank stops at the same one."""
    rng = np.random.default_rng(9)
    words = [bytes(97 + rng.integers(0, 3, size=rng.integers(1, 6))) for _ in range(500)]
    chunks = [b" " + w for w in words]
    exp = _oracle(chunks, 400)
    assert len(exp[0]) < 400
    for world in (2, 3):
        outs = _run_chain(chunks, 400, world, kcap=8, capacity=12, depth=3)
        for rank, status, pairs, counts, (steps, generals) in outs:
            assert pairs == exp[0] and counts == exp[1], f"world {world} rank {rank}"
            assert generals > 0
        assert len({(tuple(o[2]), o[4]) for o in outs}) == 1


def test_sharded_chain_step_failure_reaches_every_rank_in_the_same_step(native):
    """tail[16] of the SUM payload: a status raised inside one rank's merge pass is summed into every rank's payload
    BEFORE the table update of that step -- all ranks stop with the same merges done."""
    chunks = _chunks(native.synth_text(6000, 46))
    outs = _run_chain(chunks, 60, 3, fail_rank=1)
    done = {len(o[2]) for o in outs}
    assert len(done) == 1 and 0 < done.pop() < 60
    assert all(o[1] != 0 for o in outs)


def test_chain_shard_model_solo_and_the_limb_guard(native):
    """world of one, no network: the model equals the oracle; and the first exchange's limbs -- summed as 128 identical
    shards would be -- are refused by table_ready when a global count reaches 2^32 (and joined exactly when it does not)."""
    from minbpe_amd.dist import SoloComm
    from cpu_chain_shard import CpuChainShard, train_chain_sharded
    chunks = _chunks(native.synth_text(5000, 47))
    data = b"".join(chunks)
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    exp = _oracle(chunks, 70)
    res = train_chain_sharded(CpuChainShard(data, offs, kcap=15, capacity=48, depth=8), SoloComm(), 70)
    assert res["pairs"] == exp[0] and res["counts"] == exp[1] and res["steps"] < 40
    sh = CpuChainShard(b"a" * 4_000, None)
    for world, fits in ((64, True), (128, False)):  # 64 x 40 M < 2^32 <= 128 x 40 M
        sh.begin(4, 0, world)
        assert sh.table.numel() == 2 * 65536
        i = 97 * 256 + 97
        assert int(sh.table[i]) == 3_999 and int(sh.table[65536 + i]) == 0
        sh.table[i], sh.table[65536 + i] = 40_000_000 & 0xFFFF, 40_000_000 >> 16  # (a shard with 40 M pairs (a, a))
        sh.table.mul_(world)  # the SUM over `world` such shards: every limb sum stays far below 2^31
        assert int(sh.table.max()) < 2**31
        if fits:
            sh.table_ready()
            assert int(sh.tab[97, 97]) == 64 * 40_000_000
        else:
            with pytest.raises(OverflowError):
                sh.table_ready()
