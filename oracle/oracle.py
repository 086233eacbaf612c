"""ctypes binding of oracle/bpe_oracle.c (the CPU restatement of minbpe's
get_stats / merge / train / encode; see the C file for reference citations).

TEST INFRASTRUCTURE ONLY -- never imported by minbpe_amd/.
Parity status: pinned against fixtures generated from the reference itself
(tests/golden/) -- see tests/test_oracle_golden.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


class OracleEmptyStats(ValueError):
    """The stats dict went empty (reference: ValueError from max(), basic.py:35)."""


def build(force=False):
    """Compile oracle/bpe_oracle.c with gcc (no GPU, no reference needed)."""
    srcs = [os.path.join(_HERE, f) for f in ("bpe_oracle.c", "bpe_fast_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    lib = C.CDLL(_SO)
    p = C.c_void_p
    lib.orc_get_stats.restype = C.c_int64
    lib.orc_get_stats.argtypes = [p, p, C.c_uint64, p, p, p, p, C.c_uint64]
    lib.orc_merge.restype = C.c_uint64
    lib.orc_merge.argtypes = [p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, p]
    lib.orc_merge_chunks.restype = C.c_uint64
    lib.orc_merge_chunks.argtypes = [p, p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32]
    lib.orc_train.restype = C.c_int64
    lib.orc_train.argtypes = [p, C.c_uint64, p, C.c_uint64, C.c_int32, p, p, p, p]
    lib.orc_train_weighted.restype = C.c_int64
    lib.orc_train_weighted.argtypes = [p, C.c_uint64, p, C.c_uint64, p, C.c_int32, p, p, p, p]
    lib.orc_train_fast.restype = C.c_int64
    lib.orc_train_fast.argtypes = [p, C.c_uint64, p, C.c_uint64, C.c_int32, p, p, p, p]
    lib.orc_dedup.restype = C.c_int64
    lib.orc_dedup.argtypes = [p, p, C.c_uint64, p, p, C.c_uint64]
    lib.orc_encode.restype = C.c_int64
    lib.orc_encode.argtypes = [p, C.c_int32, p, C.c_uint64, p, C.c_uint64, p, p]
    lib.orc_encode_ids.restype = C.c_int64
    lib.orc_encode_ids.argtypes = [p, p, C.c_int32, p, C.c_uint64, p, C.c_uint64, p, p]
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _offsets(n, offsets):
    """Normalise chunk offsets to a (n_chunks+1,) uint64 array ending at n."""
    if offsets is None:
        return np.array([0, n], dtype=np.uint64)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    if len(off) == 0 or off[-1] != n:
        off = np.concatenate([off, np.array([n], dtype=np.uint64)])
    return off


def get_stats(ids, offsets=None):
    """Ordered list of ((a, b), count, first_pos) in dict-insertion order."""
    lib = _load()
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    off = _offsets(len(ids), offsets)
    cap = max(len(ids), 1)
    a = np.empty(cap, np.int32)
    b = np.empty(cap, np.int32)
    cnt = np.empty(cap, np.uint64)
    first = np.empty(cap, np.uint64)
    n = lib.orc_get_stats(_ptr(ids), _ptr(off), len(off) - 1, _ptr(a), _ptr(b),
                          _ptr(cnt), _ptr(first), cap)
    if n < 0:
        raise RuntimeError(f"orc_get_stats failed: {n}")
    return [((int(a[i]), int(b[i])), int(cnt[i]), int(first[i])) for i in range(n)]


def merge(ids, pair, idx):
    lib = _load()
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    out = np.empty(max(len(ids), 1), np.int32)
    n = lib.orc_merge(_ptr(ids), len(ids), int(pair[0]), int(pair[1]), int(idx), _ptr(out))
    return out[:n].copy()


def merge_chunks(ids, offsets, pair, idx):
    """merge() on every chunk (regex.py:60).  Returns (new ids, new n_chunks+1 offsets)."""
    lib = _load()
    buf = np.array(ids, dtype=np.int32)
    off = _offsets(len(buf), offsets).copy()
    n = lib.orc_merge_chunks(_ptr(buf), _ptr(off), len(off) - 1, int(pair[0]), int(pair[1]), int(idx))
    return buf[:n].copy(), off


def dedup(data: bytes, offsets):
    """Distinct chunks of a chunk list in order of first appearance: returns
    (data of the distinct chunks, their start offsets, weights[uint64], first_idx[uint64])."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    off = _offsets(len(buf), offsets)
    nc = len(off) - 1
    first = np.empty(max(nc, 1), np.uint64)
    wt = np.empty(max(nc, 1), np.uint64)
    nd = lib.orc_dedup(_ptr(buf) if len(buf) else None, _ptr(off), nc, _ptr(first), _ptr(wt), nc)
    if nd < 0:
        raise RuntimeError(f"orc_dedup failed: {nd}")
    first, wt = first[:nd].copy(), wt[:nd].copy()
    starts = off[first.astype(np.int64)]
    lens = off[first.astype(np.int64) + 1] - starts
    doff = np.zeros(nd, np.uint64)
    if nd > 1:
        np.cumsum(lens[:-1], out=doff[1:])
    out = np.empty(int(lens.sum()), np.uint8)
    # gather the distinct chunks' bytes (vectorised: position j of the output belongs to chunk d)
    if nd:
        owner = np.repeat(np.arange(nd), lens.astype(np.int64))
        out[:] = buf[(starts[owner] + (np.arange(len(out), dtype=np.uint64) - doff[owner])).astype(np.int64)]
    return out.tobytes(), doff, wt, first


def train(data: bytes, num_merges: int, offsets=None, raise_on_empty=True, weights=None):
    """Returns (pairs[(a,b)...], counts[...], lens[...]).  `offsets` = chunk
    start offsets into data (None = one chunk = BasicTokenizer).  `weights` (one uint64 per
    chunk): chunk c stands for weights[c] identical copies (orc_train_weighted)."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    off = _offsets(len(buf), offsets)
    nm = max(num_merges, 1)
    pairs = np.zeros(2 * nm, np.int32)
    counts = np.zeros(nm, np.uint64)
    lens = np.zeros(nm, np.uint64)
    status = C.c_int32(0)
    if weights is not None:
        wt = np.ascontiguousarray(weights, dtype=np.uint64)
        assert len(wt) == len(off) - 1
        done = lib.orc_train_weighted(_ptr(buf) if len(buf) else None, len(buf), _ptr(off),
                                      len(off) - 1, _ptr(wt), num_merges, _ptr(pairs),
                                      _ptr(counts), _ptr(lens), C.byref(status))
    else:
        done = lib.orc_train(_ptr(buf) if len(buf) else None, len(buf), _ptr(off), len(off) - 1,
                             num_merges, _ptr(pairs), _ptr(counts), _ptr(lens),
                             C.byref(status))
    if status.value == -2 and raise_on_empty:
        raise OracleEmptyStats("max() arg is an empty sequence")
    if status.value not in (0, -2):
        raise RuntimeError(f"orc_train failed: {status.value}")
    pl = [(int(pairs[2 * i]), int(pairs[2 * i + 1])) for i in range(done)]
    return pl, [int(c) for c in counts[:done]], [int(x) for x in lens[:done]]


def train_fast(data: bytes, num_merges: int, offsets=None, raise_on_empty=True):
    """train() by the incremental exact trainer (bpe_fast_oracle.c: counts kept current at the merge sites, ties by the
    first live occurrence): the same (pairs, counts, lens) as train() -- pinned to it in tests/test_fast_oracle.py -- in
    minutes where the plain loop needs days (1 GB as one stream, 31,744 merges).  Needs ~28 bytes of memory per input byte
    at vocab 32000."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    off = _offsets(len(buf), offsets)
    nm = max(num_merges, 1)
    pairs = np.zeros(2 * nm, np.int32)
    counts = np.zeros(nm, np.uint64)
    lens = np.zeros(nm, np.uint64)
    status = C.c_int32(0)
    done = lib.orc_train_fast(_ptr(buf) if len(buf) else None, len(buf), _ptr(off), len(off) - 1, num_merges,
                              _ptr(pairs), _ptr(counts), _ptr(lens), C.byref(status))
    if status.value == -2 and raise_on_empty:
        raise OracleEmptyStats("max() arg is an empty sequence")
    if status.value not in (0, -2):
        raise RuntimeError(f"orc_train_fast failed: {status.value}")
    pl = [(int(pairs[2 * i]), int(pairs[2 * i + 1])) for i in range(done)]
    return pl, [int(c) for c in counts[:done]], [int(x) for x in lens[:done]]


def encode(merges, data: bytes, offsets=None, merge_ids=None):
    """merges: list of (a, b) in rank order (an (M, 2) int32 array will do).  merge_ids: the id each
    merge writes (None: 256 + rank).  Returns (ids, out_offsets)."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    off = _offsets(len(buf), offsets)
    m = np.ascontiguousarray(np.array(merges, dtype=np.int32).reshape(-1))
    out = np.empty(max(len(buf), 1), np.int32)
    oo = np.empty(len(off), np.uint64)
    if merge_ids is not None:
        mi = np.ascontiguousarray(merge_ids, dtype=np.int32)
        assert len(mi) == len(merges)
        n = lib.orc_encode_ids(_ptr(m) if len(m) else None, _ptr(mi) if len(mi) else None, len(merges),
                               _ptr(buf) if len(buf) else None, len(buf), _ptr(off),
                               len(off) - 1, _ptr(out), _ptr(oo))
    else:
        n = lib.orc_encode(_ptr(m) if len(m) else None, len(merges),
                           _ptr(buf) if len(buf) else None, len(buf), _ptr(off),
                           len(off) - 1, _ptr(out), _ptr(oo))
    if n < 0:
        raise RuntimeError(f"orc_encode failed: {n}")
    return out[:n].copy(), oo
