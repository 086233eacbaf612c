"""Full-length golden digests at BASELINE sizes, produced by the CPU oracle
(oracle/bpe_oracle.c, itself pinned against the reference by gen_golden.py /
tests/test_oracle_golden.py) in the build container.  One-off CPU jobs of
10-40 minutes each; the digests are committed in tests/golden/big_golden.json and
consumed by tests/test_gpu_big.py and bench.py.

    python tests/golden/gen_big_golden.py cfg2      # Basic, 100 MB, 3840 merges
    python tests/golden/gen_big_golden.py cfg3s     # GPT-4 split, 150 MB, 8192 merges
    python tests/golden/gen_big_golden.py basic1g   # Basic, 1 GB, first 2048 merges
    python tests/golden/gen_big_golden.py regex1g   # GPT-4 split, 1 GB, first 2048 merges (the headline input)
    python tests/golden/gen_big_golden.py full16r   # GPT-4 split, 16 MB, ALL 31,744 merges of vocab 32000
    python tests/golden/gen_big_golden.py full12b   # Basic, 12 MB, ALL 31,744 merges
    python tests/golden/gen_big_golden.py full8r    # GPT-4 split, 8 MB (another seed), ALL 31,744 merges
    python tests/golden/gen_big_golden.py regex1g_w # the headline input, ALL 31,744 merges, through the WEIGHTED
                                                    # oracle (orc_train_weighted on the distinct chunks, in order
                                                    # of first appearance, each with its multiplicity); the
                                                    # `regex1g` digests above (plain loop, 2.6 h) are its first
                                                    # 2048 merges and must come out identical -- checked here
    python tests/golden/gen_big_golden.py cfg3s_w   # the same cross-check at 150 MB: weighted == plain, all 8192
    python tests/golden/gen_big_golden.py regex1g_dp2_w   # (dp4, dp8) the SHARDED headline of bench.py --gpus N: rank r's
                                                    # shard is 1 GB of seed 2 + r, split on its own; the job's text is the
                                                    # shards back to back; all 31,744 merges by the weighted oracle

The GPT-4 split of `cfg3s` is done here with the `regex` module exactly as the
reference does (regex.py:19,41), NOT with the native splitter: the digest of the
chunk offsets is stored too, so the GPU-side test also proves bpe_split == regex
on 150 MB of text.
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle  # noqa: E402
from helpers import checkpoint_digests  # noqa: E402
from minbpe_amd import synth_text  # noqa: E402

OUT = os.path.join(HERE, "big_golden.json")
GPT4_SPLIT_PATTERN = r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]++[\r\n]*|\s*[\r\n]|\s+(?!\S)|\s+"""

CASES = {
    # name: (bytes, seed, merges, chunked)
    "cfg2": (100_000_000, 1, 3840, False),
    "cfg3s": (150_000_000, 2, 8192, True),
    "basic1g": (1_000_000_000, 2, 2048, False),
    "regex1g": (1_000_000_000, 2, 2048, True),
    # the WHOLE vocab range of the headline (31,744 merges) on inputs the oracle finishes in
    # under an hour: mass low-count ties, V > 8448, every select/apply path above merge 8192
    "full16r": (16_000_000, 11, 31744, True),
    "full12b": (12_000_000, 12, 31744, False),
    "full8r": (8_000_000, 21, 31744, True),
    # the weighted form: (bytes, seed, merges, chunked, name of the plain case whose digests it must reproduce)
    "regex1g_w": (1_000_000_000, 2, 31744, True, "regex1g"),
    "cfg3s_w": (150_000_000, 2, 8192, True, "cfg3s"),   # a second text for the whole range (made after the chained merges went in)
    # 3.9 GB on one GPU (`bench.py --bytes 3900000000`): no plain case exists at this size (neither the `regex` module
    # nor the plain oracle gets there); the split is the native scanner's -- pinned against `regex` on 150 MB and on the
    # adversarial / fuzz cases of tests/test_split.py, not at this size -- and the merges are the weighted oracle's
    "regex3p9g_w": (3_900_000_000, 2, 31744, True, None),
}
STEP = {"basic1g": 16, "regex1g": 16}


def regex_offsets(data: bytes) -> np.ndarray:
    import regex as re
    pat = re.compile(GPT4_SPLIT_PATTERN)
    text = data.decode("utf-8")
    from array import array
    offs = array("Q")
    pos = 0
    ascii_only = len(text) == len(data)
    for m in pat.finditer(text):
        offs.append(pos if not ascii_only else m.start())
        if not ascii_only:
            pos += len(m.group().encode("utf-8"))
    return np.frombuffer(offs, dtype=np.uint64).copy()


def main_weighted(name):
    """All merges of a chunked case through orc_train_weighted.  The split comes from the native
    scanner here (the `regex` module needs > 10 minutes and tens of GB for 1 GB of text) and must
    reproduce the offsets digest the `regex` module gave for the plain case."""
    from helpers import first_divergence
    from minbpe_amd import _native
    nbytes, seed, merges, _, plain = CASES[name]
    with open(OUT) as f:
        allg = json.load(f)
    t0 = time.time()
    data = synth_text(nbytes, seed)
    offs = np.ascontiguousarray(_native.split_offsets(data, 4), dtype=np.uint64)   # 4 = the GPT-4 pattern
    if plain is None:  # no plain case at this size: the native split's own digests
        ref = {"data_sha256": hashlib.sha256(data).hexdigest(), "n_chunks": int(len(offs)),
               "offsets_sha256": hashlib.sha256(offs.tobytes()).hexdigest(), "done": 0}
    else:
        ref = allg[plain]
    assert hashlib.sha256(data).hexdigest() == ref["data_sha256"]
    assert len(offs) == ref["n_chunks"]
    assert hashlib.sha256(offs.tobytes()).hexdigest() == ref["offsets_sha256"], "split differs from the regex module's"
    print(f"{name}: {len(offs)} chunks" + (" == the regex module's" if plain else " (native split; no regex answer at this size)")
          + f", {time.time() - t0:.0f}s", flush=True)
    t1 = time.time()
    ddata, doffs, wts, _first = oracle.dedup(data, offs)
    entry = {"bytes": nbytes, "seed": seed, "merges": merges, "chunked": True, "weighted": True,
             "data_sha256": ref["data_sha256"], "n_chunks": ref["n_chunks"],
             "offsets_sha256": ref["offsets_sha256"], "n_distinct": int(len(doffs)),
             "distinct_bytes": len(ddata), "weight_sum": int(wts.sum()),
             "dedup_seconds": round(time.time() - t1, 1)}
    assert entry["weight_sum"] == ref["n_chunks"]
    print(f"{name}: {len(doffs)} distinct chunks, {len(ddata)} bytes, {entry['dedup_seconds']}s", flush=True)
    del data, offs
    t1 = time.time()
    pairs, counts, lens = oracle.train(ddata, merges, doffs, weights=wts)
    entry["oracle_seconds"] = round(time.time() - t1, 1)
    entry["done"] = len(pairs)
    entry["first"] = [list(p) for p in pairs[:4]]
    entry["last"] = [list(p) for p in pairs[-2:]]
    entry["final_len"] = lens[-1]
    entry["step"] = 256
    entry["digests"] = checkpoint_digests(pairs, counts, lens, 256)
    # the plain oracle's digests of the same input (the reference's own loop, hours of CPU) must be a prefix
    k = ref["done"]
    if plain is not None:
        same = checkpoint_digests(pairs[:k], counts[:k], lens[:k], ref["step"])
        bad = first_divergence(same, ref["digests"])
        assert bad is None and same[-1] == ref["digests"][-1], f"weighted oracle != plain oracle at merge {bad}"
    else:
        entry["split"] = "native scanner (bpe_split); no `regex` answer at this size"
    entry["equals_plain_oracle_first"] = k
    allg[name] = entry
    with open(OUT + ".tmp", "w") as f:
        json.dump(allg, f, indent=1)
    os.replace(OUT + ".tmp", OUT)
    print(f"{name}: {len(pairs)} merges, weighted oracle {entry['oracle_seconds']}s, first {k} == plain oracle, "
          f"final digest {entry['digests'][-1][1]}", flush=True)


def main_sharded(name, world):
    """bench.py --gpus `world`: every rank makes and splits its own shard (seed + rank); the merges are those of the
    shards back to back (chunks never span shards).  Weighted oracle on the distinct chunks of the whole job."""
    from minbpe_amd import _native
    nbytes, seed, merges = 1_000_000_000, 2, 31744
    t0 = time.time()
    parts, offl, shas, base = [], [], [], 0
    for r in range(world):
        d = synth_text(nbytes, seed + r)
        o = np.ascontiguousarray(_native.split_offsets(d, 4), dtype=np.uint64)
        shas.append(hashlib.sha256(d).hexdigest())
        parts.append(d)
        offl.append(o + np.uint64(base))
        base += len(d)
        print(f"{name}: shard {r}: {len(o)} chunks, {time.time() - t0:.0f}s", flush=True)
    data = b"".join(parts)
    del parts
    offs = np.concatenate(offl)
    del offl
    t1 = time.time()
    ddata, doffs, wts, _first = oracle.dedup(data, offs)
    entry = {"bytes": nbytes, "seed": seed, "merges": merges, "chunked": True, "weighted": True, "world": world,
             "shard_sha256": shas, "n_chunks": int(len(offs)), "n_distinct": int(len(doffs)),
             "distinct_bytes": len(ddata), "weight_sum": int(wts.sum()), "dedup_seconds": round(time.time() - t1, 1),
             "split": "native scanner (bpe_split), every shard on its own"}
    assert entry["weight_sum"] == entry["n_chunks"]
    n_total = len(data)
    del data, offs
    t1 = time.time()
    pairs, counts, lens = oracle.train(ddata, merges, doffs, weights=wts)
    entry["oracle_seconds"] = round(time.time() - t1, 1)
    entry["done"] = len(pairs)
    entry["first"] = [list(p) for p in pairs[:4]]
    entry["last"] = [list(p) for p in pairs[-2:]]
    entry["total_bytes"] = n_total
    entry["final_len"] = lens[-1]
    entry["step"] = 256
    entry["digests"] = checkpoint_digests(pairs, counts, lens, 256)
    with open(OUT) as f:
        allg = json.load(f)
    allg[name] = entry
    with open(OUT + ".tmp", "w") as f:
        json.dump(allg, f, indent=1)
    os.replace(OUT + ".tmp", OUT)
    print(f"{name}: {len(pairs)} merges, {entry['n_chunks']} chunks -> {entry['n_distinct']} distinct, weighted oracle "
          f"{entry['oracle_seconds']}s, final digest {entry['digests'][-1][1]}", flush=True)


def main():
    name = sys.argv[1]
    if name.startswith("regex1g_dp") and name.endswith("_w"):
        return main_sharded(name, int(name[len("regex1g_dp"):-2]))
    if len(CASES[name]) == 5:
        return main_weighted(name)
    nbytes, seed, merges, chunked = CASES[name]
    t0 = time.time()
    data = synth_text(nbytes, seed)
    entry = {"bytes": nbytes, "seed": seed, "merges": merges, "chunked": chunked,
             "data_sha256": hashlib.sha256(data).hexdigest()}
    offs = None
    if chunked:
        offs = regex_offsets(data)
        entry["n_chunks"] = int(len(offs))
        entry["offsets_sha256"] = hashlib.sha256(offs.tobytes()).hexdigest()
        print(f"{name}: regex split {len(offs)} chunks in {time.time() - t0:.0f}s", flush=True)
    t1 = time.time()
    pairs, counts, lens = oracle.train(data, merges, offs)
    entry["oracle_seconds"] = round(time.time() - t1, 1)
    entry["done"] = len(pairs)
    entry["first"] = [list(p) for p in pairs[:4]]
    entry["last"] = [list(p) for p in pairs[-2:]]
    entry["final_len"] = lens[-1]
    entry["step"] = STEP.get(name, 256 if merges >= 256 else 16)
    entry["digests"] = checkpoint_digests(pairs, counts, lens, entry["step"])
    import fcntl
    with open(OUT + ".lock", "w") as lk:          # several cases may be generated side by side
        fcntl.flock(lk, fcntl.LOCK_EX)
        allg = {}
        if os.path.exists(OUT):
            with open(OUT) as f:
                allg = json.load(f)
        allg[name] = entry
        with open(OUT + ".tmp", "w") as f:
            json.dump(allg, f, indent=1)
        os.replace(OUT + ".tmp", OUT)
    print(f"{name}: {len(pairs)} merges, oracle {entry['oracle_seconds']}s, final digest "
          f"{entry['digests'][-1][1]}", flush=True)


if __name__ == "__main__":
    main()
