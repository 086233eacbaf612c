"""Test double for minbpe_amd._native.Engine built on the CPU oracle: lets the host side of the
drop-in classes (chunking, de-duplication, special-token splicing, batch offsets, vocab tables,
error mapping) run in the CPU test suite.  Tests only -- the product has no CPU path."""
import numpy as np

import oracle
from minbpe_amd import _native


class OracleEngine:
    def __init__(self):
        self.last_train = None

    # -- training -----------------------------------------------------------------------
    def load_bytes(self, data, offsets=None, weight_exp=None):
        data = bytes(data)
        if weight_exp is not None:
            # a chunk of weight 2^k = 2^k copies (any arrangement that keeps first appearances in
            # order trains alike; adjacent copies do)
            offs = np.asarray(offsets, dtype=np.int64)
            ends = np.append(offs[1:], len(data))
            chunks = []
            for b, e, k in zip(offs, ends, weight_exp):
                chunks += [data[int(b):int(e)]] * (1 << int(k))
            data = b"".join(chunks)
            offsets = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
        self._data, self._offs = data, offsets

    def train(self, num_merges):
        pairs, counts, lens = oracle.train(self._data, num_merges, self._offs, raise_on_empty=False)
        self.last_train = dict(pairs=pairs, counts=counts, lens=lens, iter_ms=None, n_done=len(pairs))
        if len(pairs) < num_merges:
            raise ValueError("max() arg is an empty sequence")
        return self.last_train

    # -- encode / decode -------------------------------------------------------------------
    def encode_batch(self, pairs, merge_ids, data, offsets=None):
        pairs = [tuple(p) for p in np.asarray(pairs).reshape(-1, 2).tolist()]
        ids, out_off = oracle.encode(pairs, bytes(data), offsets)
        if merge_ids is not None and len(pairs):  # rank r -> merge_ids[r]
            lut = np.concatenate([np.arange(256), np.asarray(merge_ids, dtype=np.int64)])
            ids = lut[ids].astype(np.int32)
        return ids, out_off

    def decode_set_vocab(self, blob, offsets):
        self._blob, self._voff = blob, np.asarray(offsets, dtype=np.int64)

    def decode_batch(self, ids, doc_offsets=None):
        ids = np.asarray(ids, dtype=np.int64)
        bad = np.flatnonzero((ids < 0) | (ids >= len(self._voff) - 1))
        if len(bad):
            raise _native.InvalidToken(int(bad[0]))
        parts = [self._blob[self._voff[i]:self._voff[i + 1]] for i in ids]
        out = b"".join(parts)
        if doc_offsets is None:
            return out
        cum = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
        return out, cum[np.asarray(doc_offsets, dtype=np.int64)]
