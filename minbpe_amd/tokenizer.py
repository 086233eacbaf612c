"""Drop-in tokenizer classes backed by the MI355X engine (libbpe_hip.so).

Same public surface as karpathy/minbpe (minbpe/__init__.py:1-4): `Tokenizer`,
`BasicTokenizer`, `RegexTokenizer`, `GPT4Tokenizer`, plus the module-level
`get_stats` / `merge` helpers of minbpe/base.py:13-41.  Signatures, results,
exception types and the `.model` / `.vocab` file formats are the reference's;
the implementation is not: all pair counting, arg-max selection, merging and
encoding runs on the GPU through the C-ABI in include/bpe_hip.h.  Host Python
only does what the reference leaves to the `regex` module and to O(vocab)
bookkeeping ("out of scope as kernels").
"""
import os
import unicodedata

import numpy as np
import regex as re

from . import _native

# the GPT split patterns are data (regex.py:18-19), reproduced verbatim because
# chunking must be byte-identical to the reference's
GPT2_SPLIT_PATTERN = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""
GPT4_SPLIT_PATTERN = r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]++[\r\n]*|\s*[\r\n]|\s+(?!\S)|\s+"""

# compiled_pattern.pattern -> bpe_split's `which` (the scanner knows exactly these two)
_NATIVE_SPLIT = {GPT2_SPLIT_PATTERN: 2, GPT4_SPLIT_PATTERN: 4}

_engines = {}


def engine(device=None) -> "_native.Engine":
    """Process-wide engine (one ctx) per GPU, created on first use."""
    if device is None:
        device = int(os.environ.get("MINBPE_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _engines:
        _engines[device] = _native.Engine(device)
    return _engines[device]


# ---------------------------------------------------------------------------
# module-level helpers (minbpe/base.py:13-41), device-backed

def get_stats(ids, counts=None):
    """Adjacent-pair counts of `ids` as a dict in first-occurrence order;
    accumulates into `counts` when given (base.py:13-22)."""
    counts = {} if counts is None else counts
    if len(ids) < 2:
        return counts
    eng = engine()
    eng.load_ids(ids)
    for pair, c, _first in eng.get_stats():
        counts[pair] = counts.get(pair, 0) + c
    return counts


def merge(ids, pair, idx):
    """Replace every left-to-right non-overlapping occurrence of `pair` in `ids`
    by `idx` (base.py:25-41)."""
    if len(ids) == 0:
        return []
    eng = engine()
    eng.load_ids(ids)
    eng.merge(pair, idx)
    return eng.read_ids().tolist()


# ---------------------------------------------------------------------------
# rendering helpers for the human-readable .vocab file (base.py:44-61)

def replace_control_characters(s: str) -> str:
    return "".join(ch if unicodedata.category(ch)[0] != "C" else f"\\u{ord(ch):04x}" for ch in s)


def render_token(t: bytes) -> str:
    return replace_control_characters(t.decode("utf-8", errors="replace"))


def _concat_chunks(chunks):
    """list[bytes] -> (data, start offsets) with empty chunks dropped."""
    chunks = [c for c in chunks if c]
    lens = np.fromiter((len(c) for c in chunks), dtype=np.uint64, count=len(chunks))
    offs = np.zeros(len(chunks), dtype=np.uint64)
    if len(chunks) > 1:
        np.cumsum(lens[:-1], out=offs[1:])
    return b"".join(chunks), offs


class Tokenizer:
    """Base class: state, vocab construction, save/load (base.py:66-165)."""

    def __init__(self):
        self.merges = {}          # (int, int) -> int, insertion-ordered by idx
        self.pattern = ""
        self.special_tokens = {}  # str -> int
        self.vocab = self._build_vocab()

    def train(self, text, vocab_size, verbose=False):
        raise NotImplementedError

    def encode(self, text):
        raise NotImplementedError

    def decode(self, ids):
        raise NotImplementedError

    def _build_vocab(self):
        vocab = {i: bytes([i]) for i in range(256)}
        for (left, right), idx in self.merges.items():
            vocab[idx] = vocab[left] + vocab[right]
        for tok, idx in self.special_tokens.items():
            vocab[idx] = tok.encode("utf-8")
        return vocab

    # -- device training shared by Basic/Regex ------------------------------------
    def _train_on_device(self, data: bytes, offsets, vocab_size: int, verbose: bool, weight_exp=None):
        assert vocab_size >= 256
        num_merges = vocab_size - 256
        eng = engine()
        eng.load_bytes(data, offsets, weight_exp)
        failure = None
        try:
            res = eng.train(num_merges)
        except ValueError as e:  # stats ran empty: same exception as max() in the reference
            res, failure = eng.last_train, e
        merges, vocab = {}, {i: bytes([i]) for i in range(256)}
        for i, pair in enumerate(res["pairs"]):
            idx = 256 + i
            merges[pair] = idx
            vocab[idx] = vocab[pair[0]] + vocab[pair[1]]
            if verbose:
                print(f"merge {i+1}/{num_merges}: {pair} -> {idx} ({vocab[idx]}) had "
                      f"{res['counts'][i]} occurrences")
        if failure is not None:
            raise failure  # like the reference, merges/vocab are NOT updated
        self.merges = merges
        self.vocab = vocab

    # -- device encoding of a batch of byte chunks ----------------------------------
    def _encode_chunks(self, chunks):
        """chunks: list[bytes] -> (ids ndarray, token start offset of each
        non-empty chunk).  One device batch: K4 (bpe_encode_batch)."""
        data, offs = _concat_chunks(chunks)
        if not data:
            return np.empty(0, np.int32), np.empty(0, np.uint64)
        pairs, mids = self._merge_table()
        ids, out_off = engine().encode_batch(pairs, mids, data, offs)
        return ids, out_off[:-1]

    def _merge_table(self):
        """merges as device arrays, ordered by priority = the dict's value
        (`min(stats, key=merges.get)` in the reference, basic.py:64)."""
        # the cache holds the dict OBJECT (an id() alone can be recycled by a later dict once
        # train()/load() have dropped the old one) and its length
        key = (self.merges, len(self.merges))
        cached = getattr(self, "_mt_cache", None)
        if cached is None or cached[0][0] is not key[0] or cached[0][1] != key[1]:
            items = sorted(self.merges.items(), key=lambda kv: kv[1])
            pairs = np.array([p for p, _ in items], dtype=np.int32).reshape(-1, 2)
            mids = np.array([i for _, i in items], dtype=np.int32)
            self._mt_cache = (key, pairs, mids)
        return self._mt_cache[1], self._mt_cache[2]

    # -- device decoding of a batch of token ids ---------------------------
    def _decode_extra(self):
        """ids decode() accepts besides self.vocab: {id: str}"""
        return {}

    def _invalid_token(self, idx):
        return KeyError(idx)  # what `self.vocab[idx]` raises (basic.py:53, gpt4.py:89)

    def _finish_bytes(self, raw):
        return raw

    def _decode_table(self):
        """vocab (+ special tokens) as one dense table for the device: (blob, offsets, V,
        sparse) where ids 0..V-1 are table indices as they are and `sparse` maps every
        other known id to its index."""
        extra = self._decode_extra()
        key = (self.vocab, len(self.vocab), tuple(sorted(extra.items())))
        cached = getattr(self, "_dt_cache", None)
        if (cached is None or cached[0][0] is not key[0] or cached[0][1] != key[1]
                or cached[0][2] != key[2]):
            vocab = self.vocab
            V = 0
            while V in vocab:
                V += 1
            table = [vocab[i] for i in range(V)]
            sparse = {}
            for idx, tok in vocab.items():
                if not 0 <= idx < V:
                    sparse[idx] = len(table)
                    table.append(tok)
            for idx, tok in extra.items():
                if idx not in vocab:
                    sparse[idx] = len(table)
                    table.append(tok.encode("utf-8"))
            offs = np.zeros(len(table) + 1, dtype=np.uint64)
            np.cumsum(np.fromiter((len(t) for t in table), dtype=np.uint64, count=len(table)), out=offs[1:])
            self._dt_cache = (key, b"".join(table), offs, V, sparse)
        return self._dt_cache[1:]

    def decode_batch(self, ids, doc_offsets=None):
        """The bytes decode() joins (`b"".join(vocab[idx] for idx in ids)`), produced on the
        device for a whole batch of token ids at once; not in the reference (its decode is
        a per-token Python loop).  With doc_offsets -- token positions, e.g. the start of
        every document plus len(ids) -- also returns the byte offset of each position, so
        document d is out[b[d]:b[d+1]].  Unknown ids raise what decode() raises, for the
        first such id.  Returns bytes (no UTF-8 decoding: documents are cut on bytes)."""
        arr = np.asarray(ids)
        if arr.size and arr.dtype.kind not in "iu":
            raise TypeError("token ids must be integers")
        arr = arr.astype(np.int64, copy=False).reshape(-1)
        blob, offs, V, sparse = self._decode_table()
        outside = np.flatnonzero((arr < 0) | (arr >= V))
        if len(outside):
            arr = arr.copy()
            for p in outside.tolist():
                j = sparse.get(int(arr[p]))
                if j is None:
                    raise self._invalid_token(int(arr[p]))
                arr[p] = j
        eng = engine()
        if getattr(eng, "_decode_owner", None) is not blob:  # the table stays resident per engine
            eng.decode_set_vocab(blob, offs)
            eng._decode_owner = blob
        res = eng.decode_batch(arr.astype(np.int32), doc_offsets)
        if doc_offsets is None:
            return self._finish_bytes(res)
        return self._finish_bytes(res[0]), res[1]

    # -- persistence (file formats of base.py:97-165, byte for byte) -----------------
    def save(self, file_prefix):
        lines = ["minbpe v1", f"{self.pattern}", f"{len(self.special_tokens)}"]
        lines += [f"{tok} {idx}" for tok, idx in self.special_tokens.items()]
        lines += [f"{a} {b}" for a, b in self.merges]
        with open(file_prefix + ".model", "w") as f:
            f.write("\n".join(lines) + "\n")
        parents = {idx: pair for pair, idx in self.merges.items()}
        with open(file_prefix + ".vocab", "w", encoding="utf-8") as f:
            for idx, tok in self.vocab.items():
                shown = render_token(tok)
                if idx in parents:
                    a, b = parents[idx]
                    f.write(f"[{render_token(self.vocab[a])}][{render_token(self.vocab[b])}]"
                            f" -> [{shown}] {idx}\n")
                else:
                    f.write(f"[{shown}] {idx}\n")

    def load(self, model_file):
        assert model_file.endswith(".model")
        with open(model_file, "r", encoding="utf-8") as f:
            assert f.readline().strip() == "minbpe v1"
            self.pattern = f.readline().strip()
            specials = {}
            for _ in range(int(f.readline().strip())):
                tok, idx = f.readline().strip().split()
                specials[tok] = int(idx)
            merges = {}
            for idx, line in enumerate(f, start=256):
                a, b = map(int, line.split())
                merges[(a, b)] = idx
        self.merges = merges
        self.special_tokens = specials
        self.vocab = self._build_vocab()


class BasicTokenizer(Tokenizer):
    """Byte-level BPE over the whole text as one chunk (basic.py:15-74)."""

    def __init__(self):
        super().__init__()

    def train(self, text, vocab_size, verbose=False):
        # (a long text is encoded by all host threads: _native.utf8_encode -- the same bytes as text.encode("utf-8"))
        self._train_on_device(_native.utf8_encode(text), None, vocab_size, verbose)

    def decode(self, ids):
        return b"".join(self.vocab[i] for i in ids).decode("utf-8", errors="replace")

    def encode(self, text):
        ids, _ = self._encode_chunks([text.encode("utf-8")])
        return ids.tolist()


class RegexTokenizer(Tokenizer):
    """BPE over regex-split chunks with optional special tokens (regex.py:22-164)."""

    def __init__(self, pattern=None):
        super().__init__()
        self.pattern = GPT4_SPLIT_PATTERN if pattern is None else pattern
        self.compiled_pattern = re.compile(self.pattern)
        self.special_tokens = {}
        self.inverse_special_tokens = {}

    def _split(self, text):
        return [piece.encode("utf-8") for piece in re.findall(self.compiled_pattern, text)]

    def _chunked(self, text, for_training=False):
        """(utf-8 bytes of the chunks back to back, chunk start offsets) -- what
        re.findall(self.compiled_pattern, text) yields (regex.py:41,114).  The two GPT
        patterns go through the native scanner (bpe_split, checked against `regex` in
        tests/test_split.py); any other pattern through the `regex` module itself."""
        which = _NATIVE_SPLIT.get(self.compiled_pattern.pattern)
        if which is not None:
            # (train() takes any buffer of bytes: a long text is encoded by all host threads there)
            data = _native.utf8_encode(text) if for_training else text.encode("utf-8")
            return data, _native.split_offsets(data, which)
        return _concat_chunks(self._split(text))

    # Train on the DISTINCT chunks, each weighted by how often it occurs (bpe_dedup_chunks):
    # same merges, counts and tie-breaks as over the full chunk list, a fraction
    # of the stream.  True / False force it.  "auto": round 1's engine made the host pass worth
    # ~1500 device merges over the full list, so it was on from 2000 merges (DEDUP_AUTO_MERGES:
    # still the rule of the sharded path, dist.py); since round 6 the device trains 1 GB to
    # vocab 32000 in 0.58 s against 0.19 s on the distinct chunks, and the host pass costs
    # 0.45-0.6 s per GB on a 256-thread host (profiles/r6_final_bench.json: the whole call 0.88 s
    # without it, 1.09 s with it) -- so a single process turns it on only for texts of
    # DEDUP_AUTO_BYTES or more, where the stream nears the 2^32-byte limit of one GPU
    #  and the distinct chunks are what still fits.
    dedup = "auto"
    DEDUP_AUTO_MERGES = 2000
    DEDUP_AUTO_BYTES = 1 << 31

    def train(self, text, vocab_size, verbose=False):
        data, offs = self._chunked(text, for_training=True)
        wexp = None
        want = (len(data) >= self.DEDUP_AUTO_BYTES) if self.dedup == "auto" else bool(self.dedup)
        if want and len(offs) > 1:
            data, offs, wexp, _ = _native.dedup_chunks(data, offs)
        self._train_on_device(data, offs, vocab_size, verbose, wexp)

    def register_special_tokens(self, special_tokens):
        self.special_tokens = special_tokens
        self.inverse_special_tokens = {idx: tok for tok, idx in special_tokens.items()}

    def decode(self, ids):
        parts = []
        for idx in ids:
            if idx in self.vocab:
                parts.append(self.vocab[idx])
            elif idx in self.inverse_special_tokens:
                parts.append(self.inverse_special_tokens[idx].encode("utf-8"))
            else:
                raise ValueError(f"invalid token id: {idx}")
        return b"".join(parts).decode("utf-8", errors="replace")

    def _decode_extra(self):
        return self.inverse_special_tokens

    def _invalid_token(self, idx):
        return ValueError(f"invalid token id: {idx}")  # regex.py:87

    def _prepare_chunk(self, chunk_bytes):
        return chunk_bytes  # GPT4Tokenizer permutes bytes here (a per-byte map)

    def _encode_chunk(self, text_bytes):
        ids, _ = self._encode_chunks([self._prepare_chunk(text_bytes)])
        return ids.tolist()

    def _encode_flat(self, data, offs):
        """encode chunks given as one byte string + start offsets (one device batch)"""
        if not data:
            return np.empty(0, np.int32), np.zeros(1, np.uint64)
        pairs, mids = self._merge_table()
        return engine().encode_batch(pairs, mids, self._prepare_chunk(data), offs)

    def encode_ordinary(self, text):
        data, offs = self._chunked(text)
        return self._encode_flat(data, offs)[0].tolist()

    def encode_ordinary_batch(self, texts):
        """encode_ordinary() of every text, as one device batch (not in the reference, whose
        encode takes one string).  Returns (ids, doc_offsets): int32 ids of all texts back to
        back and len(texts) + 1 offsets, text d being ids[doc_offsets[d]:doc_offsets[d + 1]].
        Every text is split on its own, so the ids are those of a loop over encode_ordinary."""
        enc = [t.encode("utf-8") for t in texts]
        if not enc:
            return np.empty(0, np.int32), np.zeros(1, np.uint64)
        which = _NATIVE_SPLIT.get(self.compiled_pattern.pattern)
        if which is not None:
            data = b"".join(enc)
            doff = np.zeros(len(enc), dtype=np.uint64)
            np.cumsum(np.fromiter((len(e) for e in enc[:-1]), dtype=np.uint64, count=len(enc) - 1), out=doff[1:])
            offs, first = _native.split_docs(data, doff, which)
        else:  # any other pattern: the regex module, one text at a time
            pieces, first = [], [0]
            for t in texts:
                pieces += [p for p in self._split(t) if p]
                first.append(len(pieces))
            data, offs = _concat_chunks(pieces)
            first = np.array(first, dtype=np.uint64)
        ids, out_off = self._encode_flat(data, offs)
        return ids, out_off[first.astype(np.int64)]

    def encode(self, text, allowed_special="none_raise"):
        if allowed_special == "all":
            special = self.special_tokens
        elif allowed_special == "none":
            special = {}
        elif allowed_special == "none_raise":
            special = {}
            assert all(tok not in text for tok in self.special_tokens)
        elif isinstance(allowed_special, set):
            special = {k: v for k, v in self.special_tokens.items() if k in allowed_special}
        else:
            raise ValueError(f"allowed_special={allowed_special} not understood")
        if not special:
            return self.encode_ordinary(text)
        splitter = "(" + "|".join(re.escape(k) for k in special) + ")"
        parts = re.split(splitter, text)
        # one device batch for all ordinary parts; specials spliced back in order
        datas, offs_list, nchunks, base = [], [], [], 0
        for part in parts:
            if part in special:
                nchunks.append(None)
                continue
            d, o = self._chunked(part)
            datas.append(d)
            offs_list.append(o + np.uint64(base))
            nchunks.append(len(o))
            base += len(d)
        data = b"".join(datas)
        offs = np.concatenate(offs_list) if offs_list else np.empty(0, np.uint64)
        ids, out_off = self._encode_flat(data, offs)
        out, ci = [], 0
        for part, k in zip(parts, nchunks):
            if k is None:
                out.append(special[part])
            elif k:
                out.extend(ids[int(out_off[ci]):int(out_off[ci + k])].tolist())
                ci += k
        return out


class GPT4Tokenizer(RegexTokenizer):
    """cl100k_base through the same engine (gpt4.py:57-130).  Needs `tiktoken`
    for the published ranks; without it construction raises ImportError."""

    SPECIAL_TOKENS = {
        '<|endoftext|>': 100257, '<|fim_prefix|>': 100258, '<|fim_middle|>': 100259,
        '<|fim_suffix|>': 100260, '<|endofprompt|>': 100276,
    }

    def __init__(self, ranks=None):
        """ranks: None -> cl100k_base from the `tiktoken` package, as the reference does
        (gpt4.py:63-64); or the same table given directly -- a {token bytes: rank} dict, or the
        path of a `.tiktoken` file (lines of "<base64 token> <rank>") -- for machines without
        the package or without network access."""
        super().__init__(pattern=GPT4_SPLIT_PATTERN)
        if ranks is None:
            try:
                import tiktoken
                get_encoding = tiktoken.get_encoding  # (a namespace stub or a broken install is no tiktoken either)
            except (ImportError, AttributeError) as e:
                raise ImportError("GPT4Tokenizer needs the `tiktoken` package for cl100k_base ranks "
                                  "(or pass ranks= a dict / a .tiktoken file)") from e
            ranks = get_encoding("cl100k_base")._mergeable_ranks
        elif not isinstance(ranks, dict):
            ranks = load_tiktoken_ranks(ranks)
        self.merges = _recover_merges(ranks)
        vocab = {i: bytes([i]) for i in range(256)}
        for (a, b), idx in self.merges.items():
            vocab[idx] = vocab[a] + vocab[b]
        self.vocab = vocab
        self.byte_shuffle = {i: ranks[bytes([i])] for i in range(256)}
        self.inverse_byte_shuffle = {v: k for k, v in self.byte_shuffle.items()}
        self._shuffle_lut = bytes(self.byte_shuffle[i] for i in range(256))
        self._unshuffle_lut = bytes(self.inverse_byte_shuffle[i] for i in range(256))
        self.register_special_tokens(self.SPECIAL_TOKENS)

    def _prepare_chunk(self, chunk_bytes):
        return chunk_bytes.translate(self._shuffle_lut)

    def decode(self, ids):
        raw = b"".join(self.vocab[i] for i in ids).translate(self._unshuffle_lut)
        return raw.decode("utf-8", errors="replace")

    def _decode_extra(self):
        return {}  # gpt4.py:89 looks ids up in self.vocab only

    def _invalid_token(self, idx):
        return KeyError(idx)

    def _finish_bytes(self, raw):
        return raw.translate(self._unshuffle_lut)

    def train(self, text, vocab_size, verbose=False):
        raise NotImplementedError

    def save(self, file_prefix):
        raise NotImplementedError("GPT4Tokenizer cannot be saved.")

    def load(self, model_file):
        raise NotImplementedError("GPT4Tokenizer cannot be loaded.")

    def save_vocab(self, vocab_file):
        vocab = {i: bytes([self.inverse_byte_shuffle[i]]) for i in range(256)}
        for (a, b), idx in self.merges.items():
            vocab[idx] = vocab[a] + vocab[b]
        parents = {idx: pair for pair, idx in self.merges.items()}
        with open(vocab_file, "w", encoding="utf-8") as f:
            for idx, tok in vocab.items():
                if idx in parents:
                    a, b = parents[idx]
                    f.write(f"[{render_token(vocab[a])}][{render_token(vocab[b])}]"
                            f" -> [{render_token(tok)}] {idx}\n")
                else:
                    f.write(f"[{render_token(tok)}] {idx}\n")


def load_tiktoken_ranks(path):
    """A `.tiktoken` rank file -> {token bytes: rank}."""
    import base64
    ranks = {}
    with open(path, "rb") as f:
        for line in f:
            if line.strip():
                tok, rank = line.split()
                ranks[base64.b64decode(tok)] = int(rank)
    return ranks


def _split_by_ranks(ranks, token, max_rank):
    """Re-run BPE on one token's bytes using only ranks below max_rank
    (gpt4.py:11-26): the two parts left are the pair that was merged."""
    parts = [bytes([b]) for b in token]
    while True:
        best = None
        for i in range(len(parts) - 1):
            r = ranks.get(parts[i] + parts[i + 1])
            if r is not None and (best is None or r < best[0]):
                best = (r, i)
        if best is None or (max_rank is not None and best[0] >= max_rank):
            return parts
        i = best[1]
        parts[i:i + 2] = [parts[i] + parts[i + 1]]


def _recover_merges(ranks):
    """mergeable_ranks (bytes -> rank) -> {(id, id): rank}  (gpt4.py:29-46)."""
    merges = {}
    for token, rank in ranks.items():
        if len(token) == 1:
            continue
        pair = _split_by_ranks(ranks, token, rank)
        assert len(pair) == 2
        merges[(ranks[pair[0]], ranks[pair[1]])] = rank
    return merges
