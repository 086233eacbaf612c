"""Shared helpers for the parity tests."""
import numpy as np
import regex as re

from minbpe_amd.tokenizer import GPT4_SPLIT_PATTERN

_PAT = re.compile(GPT4_SPLIT_PATTERN)


def case_text(case, native):
    if "text" in case:
        return case["text"]
    n, seed = case["synth"]
    return native.synth_text(n, seed).decode("utf-8")


def split_chunks(text, pattern=_PAT):
    """(data, start offsets) the way RegexTokenizer.train chunks its input."""
    chunks = [c.encode("utf-8") for c in re.findall(pattern, text)]
    chunks = [c for c in chunks if c]
    offs = np.zeros(len(chunks), dtype=np.uint64)
    if len(chunks) > 1:
        np.cumsum(np.fromiter((len(c) for c in chunks[:-1]), dtype=np.uint64), out=offs[1:])
    return b"".join(chunks), offs


def pattern_of(case):
    """compiled split pattern of a golden case (default: the GPT-4 pattern, regex.py:29)"""
    return re.compile(case["pattern"]) if case.get("pattern") else _PAT


def data_for(case, native):
    text = case_text(case, native)
    if case["kind"] == "basic":
        return text.encode("utf-8"), None
    return split_chunks(text, pattern_of(case))


def toy_rank_table(base_tok, seed):
    """A cl100k-shaped {token bytes: rank} table from a trained tokenizer: single bytes get a
    permutation of 0..255 (tiktoken's byte order is not the identity), merged tokens rank = id."""
    import random
    perm = list(range(256))
    random.Random(seed).shuffle(perm)
    ranks = {bytes([b]): perm[b] for b in range(256)}
    for idx in range(256, 256 + len(base_tok.merges)):
        assert base_tok.vocab[idx] not in ranks  # vocabularies this small have no duplicate byte strings
        ranks[base_tok.vocab[idx]] = idx
    return perm, ranks


def cl100k_shaped_table(base_pairs, total, seed, ids="rank"):
    """A rank table of cl100k_base's SIZE (cl100k: 100,000 merges, ids up to 100,255; its ranks are not
    available offline) around a trained merge list: the `base_pairs` (ids 256 + position) keep their order
    but are spread over `total` ranks, the ranks in between are filled with merges of tokens defined so
    far (half of them pairs of early tokens, which do occur in text and change how it encodes).
    ids = "rank": merge r writes id 256 + r (RegexTokenizer; None is returned for the id list);
    ids = "sparse": merge r writes 1000 + 3 r (a merges dict whose values are not consecutive, as
    GPT4Tokenizer's are ranks: gpt4.py:65).  Returns (pairs as an (total, 2) int32 array, ids or None)."""
    rng = np.random.default_rng(seed)
    nb = len(base_pairs)
    assert total >= nb
    real_at = np.zeros(total, dtype=bool)
    real_at[rng.choice(total, nb, replace=False)] = True
    id_of = (lambda r: 256 + r) if ids == "rank" else (lambda r: 1000 + 3 * r)
    new_of_old = list(range(256)) + [0] * nb      # old id (256 + k) -> id in the big table
    old_of_new = {i: i for i in range(256)}
    real_old = {tuple(p) for p in base_pairs}
    defined = list(range(256))
    used = set()
    out = np.empty((total, 2), dtype=np.int32)
    k = 0
    coin = rng.random(total)
    pick = rng.integers(0, 1 << 30, size=(total, 2))
    for r in range(total):
        if real_at[r]:
            a, b = base_pairs[k]
            pair = (new_of_old[a], new_of_old[b])
            new_of_old[256 + k] = id_of(r)
            old_of_new[id_of(r)] = 256 + k
            k += 1
        else:
            j = 0
            while True:
                lim = min(len(defined), 400) if coin[r] < 0.5 else len(defined)
                pair = (defined[(pick[r, 0] + j) % lim], defined[(pick[r, 1] + 7 * j) % lim])
                old = (old_of_new.get(pair[0]), old_of_new.get(pair[1]))
                if pair not in used and old not in real_old:
                    break
                j += 1
        used.add(pair)
        out[r] = pair
        defined.append(id_of(r))
    mids = None if ids == "rank" else np.array([id_of(r) for r in range(total)], dtype=np.int32)
    return out, mids


def checkpoint_digests(pairs, counts, lens, step):
    """[[k, sha256-prefix of the first k merges], ...] every `step` merges and at the end.
    The digest covers pairs, counts AND stream lengths, as little-endian int64 triples
    (a, b, count, len) -- cheap to recompute for tens of thousands of merges."""
    import hashlib
    n = len(pairs)
    arr = np.empty((n, 4), dtype="<i8")
    if n:
        arr[:, 0:2] = np.asarray(pairs, dtype=np.int64).reshape(n, 2)
        arr[:, 2] = np.asarray(counts, dtype=np.int64)
        arr[:, 3] = np.asarray(lens, dtype=np.int64)
    h = hashlib.sha256()
    out = []
    k = 0
    while k < n:
        k2 = min(k + step, n)
        h.update(arr[k:k2].tobytes())
        out.append([k2, h.copy().hexdigest()[:16]])
        k = k2
    return out


def first_divergence(got, want):
    """Compare two checkpoint lists; returns None if every common checkpoint agrees, else the
    merge count of the first checkpoint that differs."""
    want_d = {k: d for k, d in want}
    for k, d in got:
        if k in want_d and want_d[k] != d:
            return k
    return None
