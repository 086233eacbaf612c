"""ctypes binding of libbpe_hip.so (include/bpe_hip.h).

The HIP library is the product: there is no CPU fallback.  If the shared
library has not been built, importing this module raises ImportError with the
build command; if no GPU is present, Engine() raises RuntimeError.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MINBPE_AMD_LIB", os.path.join(_HERE, "lib", "libbpe_hip.so"))

BPE_OK = 0
BPE_E_HIP = -1
BPE_E_ARG = -2
BPE_E_EMPTY_STATS = -3
BPE_E_STATE = -4
BPE_E_CAP = -5
BPE_E_LIMIT = -6
BPE_E_INTERNAL = -7

PROF_KINDS = ("widen", "pair_count", "argmax", "merge", "table", "encode", "decode")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -m minbpe_amd.build` "
        "(hipcc --offload-arch=gfx950). minbpe_amd has no CPU fallback.")



def _load_library():
    """PyTorch ships private copies of libamdhip64 / libhsa-runtime64.  Two HIP runtimes cannot
    both bring up the same GPU in one process (the second one reports "No HIP GPUs are
    available"; measured, tools/gpu_probe.sh), and which copy a process ends up with is decided
    by load order: with torch imported first, this library binds to torch's copy (same SONAME)
    and the process has one runtime.  So when torch is installed it is imported before the
    library is loaded -- the order bench.py and the GPU tests run in.  MINBPE_AMD_NO_TORCH=1
    skips this (for processes that will never import torch)."""
    if "torch" not in sys.modules and not os.environ.get("MINBPE_AMD_NO_TORCH"):
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except Exception:
            pass
    return C.CDLL(LIB_PATH)


_lib = _load_library()
_p = C.c_void_p
# bpe_allreduce_fn (include/bpe_hip.h): (user, device_buffer, count, dtype, op, hip_stream) -> 0 on success
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p)
DT_INT32, DT_INT64, OP_SUM, OP_MIN = 0, 1, 0, 1
_u64 = C.c_uint64
_i32 = C.c_int32

_SIGS = {
    "bpe_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "bpe_destroy": (None, [_p]),
    "bpe_last_error": (C.c_char_p, [_p]),
    "bpe_set_stream": (C.c_int, [_p, _p]),
    "bpe_set_option": (C.c_int, [_p, C.c_char_p, C.c_int64]),
    "bpe_load_bytes": (C.c_int, [_p, _p, _u64, _p, _u64]),
    "bpe_load_bytes_weighted": (C.c_int, [_p, _p, _u64, _p, _u64, _p]),
    "bpe_load_ids": (C.c_int, [_p, _p, _u64, _p, _u64]),
    "bpe_get_stats": (C.c_int, [_p, C.POINTER(_u64)]),
    "bpe_read_stats": (C.c_int, [_p, _p, _p, _p, _p, _u64, C.POINTER(_u64)]),
    "bpe_argmax": (C.c_int, [_p, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_u64)]),
    "bpe_merge": (C.c_int, [_p, _i32, _i32, _i32, C.POINTER(_u64)]),
    "bpe_len": (C.c_int, [_p, C.POINTER(_u64)]),
    "bpe_read_ids": (C.c_int, [_p, _p, _u64]),
    "bpe_read_chunk_starts": (C.c_int, [_p, _p, _u64, C.POINTER(_u64)]),
    "bpe_train": (C.c_int, [_p, _i32, _p, _p, _p, _p, C.POINTER(_i32)]),
    "bpe_dp_begin": (C.c_int, [_p, _i32, _i32, _i32]),
    "bpe_dp_buffers": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_u64), C.POINTER(_p), C.POINTER(_u64),
                                 C.POINTER(_p)]),
    "bpe_dp_table_ready": (C.c_int, [_p]),
    "bpe_dp_select": (C.c_int, [_p, _i32]),
    "bpe_dp_merge": (C.c_int, [_p, _i32]),
    "bpe_dp_apply": (C.c_int, [_p, _i32]),
    "bpe_dp_poll": (C.c_int, [_p, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_u64),
                              C.POINTER(_u64), C.POINTER(_i32)]),
    "bpe_dp_end": (C.c_int, [_p]),
    "bpe_comm_available": (C.c_int, []),
    "bpe_comm_unique_id": (C.c_int, [_p]),
    "bpe_comm_init": (C.c_int, [_p, _i32, _i32, _p]),
    "bpe_comm_destroy": (C.c_int, [_p]),
    "bpe_dp_train": (C.c_int, [_p, _i32, _p, _p, _p, C.POINTER(_i32)]),
    "bpe_dp_train_cb": (C.c_int, [_p, _i32, _i32, _i32, C.c_void_p, _p, _p, _p, _p, C.POINTER(_i32)]),
    "bpe_encode_batch": (C.c_int, [_p, _p, _p, _i32, _p, _u64, _p, _u64, _p, _p, C.POINTER(_u64)]),
    "bpe_encode_batch_resident": (C.c_int, [_p, _p, _p, _i32, _p, _u64, _p, _u64, _p, _p, C.POINTER(_u64)]),
    "bpe_decode_set_vocab": (C.c_int, [_p, _p, _p, _i32]),
    "bpe_decode_batch": (C.c_int, [_p, _p, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "bpe_decode_read": (C.c_int, [_p, _p, _u64, _p, _u64, _p]),
    "bpe_prof_reset": (C.c_int, [_p]),
    "bpe_prof_read": (C.c_int, [_p, _p, _p, _p]),
    "bpe_encode_uses_16bit": (C.c_int, [_p, C.c_int32]),
    "bpe_train_stats": (C.c_int, [_p, _p]),
    "bpe_train_stats_ex": (C.c_int, [_p, _p, C.c_int]),
    "bpe_utf8_encode": (C.c_int, [C.c_int, _p, _u64, _p, _u64, C.POINTER(_u64), C.c_int]),
    "bpe_split": (C.c_int, [C.c_int, _p, _u64, _p, _u64, C.POINTER(_u64), C.c_int]),
    "bpe_dedup_chunks": (C.c_int, [_p, _u64, _p, _u64, _p, _p, _p, C.POINTER(_u64), C.POINTER(_u64),
                                  C.POINTER(_u64), C.c_int]),
    "bpe_split_docs": (C.c_int, [C.c_int, _p, _u64, _p, _u64, _p, _u64, C.POINTER(_u64), _p, C.c_int]),
    "bpe_synth_text": (C.c_int, [_p, _u64, _u64]),
    "bpe_version": (C.c_char_p, []),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(_lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def exported_symbols():
    """Names bound above (tests check them against include/bpe_hip.h)."""
    return sorted(_SIGS)


def version():
    return _lib.bpe_version().decode()


def synth_text(n: int, seed: int) -> bytes:
    """Deterministic synthetic UTF-8 text (host only, no GPU needed)."""
    buf = np.empty(n, dtype=np.uint8)
    rc = _lib.bpe_synth_text(buf.ctypes.data_as(_p), n, seed)
    if rc != BPE_OK:
        raise RuntimeError(f"bpe_synth_text failed: {rc}")
    return buf.tobytes()


def utf8_encode(text: str, threads: int = 0):
    """text.encode("utf-8") (basic.py:25, regex.py:44) -- by all host threads for a long text: CPython encodes with one
    thread, 0.7 s per GB of a str with characters beyond the BMP, more than the device side of a whole train().  A
    compact CPython str is a header followed by its code points, 1, 2 or 4 bytes each; the library transcodes them in
    place (bpe_utf8_encode).  The layout is checked, not assumed: interpreter and version, the object's state bits, its
    size, and the first and last code points read through the pointer against the str itself -- anything that does not
    fit (another interpreter, a non-compact or short or ASCII str, a lone surrogate) goes to str.encode.  Returns bytes
    or a uint8 array (both are buffers: what every consumer here takes)."""
    n = len(text)
    if n < (1 << 22) or sys.implementation.name != "cpython" or sys.version_info[:2] != (3, 10) or type(text) is not str:
        return text.encode("utf-8")
    state = C.c_uint8.from_address(id(text) + 32).value  # PyASCIIObject.state: interned:2 kind:3 compact:1 ascii:1 ready:1
    kind, compact, ascii_, ready = (state >> 2) & 7, (state >> 5) & 1, (state >> 6) & 1, (state >> 7) & 1
    if not (compact and ready) or ascii_ or kind not in (1, 2, 4) or sys.getsizeof(text) != 72 + (n + 1) * kind:
        return text.encode("utf-8")  # (an ASCII str encodes by memcpy; a cached utf-8 copy changes the size: both rare here)
    ptr = id(text) + 72  # sizeof(PyCompactUnicodeObject)
    ctype = {1: C.c_uint8, 2: C.c_uint16, 4: C.c_uint32}[kind]
    probe = (ctype * 8).from_address(ptr), (ctype * 8).from_address(ptr + (n - 8) * kind)
    if [ord(c) for c in text[:8]] != list(probe[0]) or [ord(c) for c in text[-8:]] != list(probe[1]):
        return text.encode("utf-8")
    nb = _u64(0)
    if _lib.bpe_utf8_encode(kind, C.c_void_p(ptr), n, None, 0, C.byref(nb), threads) != BPE_OK:
        return text.encode("utf-8")  # (a lone surrogate: str.encode raises the reference's own exception)
    out = np.empty(nb.value, np.uint8)
    if _lib.bpe_utf8_encode(kind, C.c_void_p(ptr), n, _ptr(out), len(out), C.byref(nb), threads) != BPE_OK:
        return text.encode("utf-8")
    return out


def split_offsets(data: bytes, which: int, threads: int = 0):
    """Chunk start offsets of regex.findall(GPT-2 | GPT-4 split pattern) over UTF-8 `data`
    (host only).  which: 2 or 4."""
    buf = np.frombuffer(data, dtype=np.uint8)
    n = _u64(0)
    out = np.empty(len(buf) // 3 + 16, np.uint64)  # GPT-style chunks average > 4 bytes
    rc = _lib.bpe_split(which, _ptr(buf) if len(buf) else None, len(buf), _ptr(out), len(out), C.byref(n), threads)
    if rc == BPE_E_CAP:  # unusually short chunks: now we know the count
        out = np.empty(n.value, np.uint64)
        rc = _lib.bpe_split(which, _ptr(buf), len(buf), _ptr(out), len(out), C.byref(n), threads)
    if rc != BPE_OK:
        raise RuntimeError(f"bpe_split failed: {rc}")
    return out[:n.value].copy() if n.value * 2 < len(out) else out[:n.value]


def split_docs(data: bytes, doc_offsets, which: int, threads: int = 0):
    """split_offsets for documents laid out back to back: every document is split on its own.
    Returns (chunk start offsets, index of each document's first chunk [n_docs + 1])."""
    buf = np.frombuffer(data, dtype=np.uint8)
    doff = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
    first = np.zeros(len(doff) + 1, np.uint64)
    n = _u64(0)
    out = np.empty(len(buf) // 3 + len(doff) + 16, np.uint64)
    args = (which, _ptr(buf) if len(buf) else None, len(buf), _ptr(doff) if len(doff) else None, len(doff))
    rc = _lib.bpe_split_docs(*args, _ptr(out), len(out), C.byref(n), _ptr(first), threads)
    if rc == BPE_E_CAP:
        out = np.empty(n.value, np.uint64)
        rc = _lib.bpe_split_docs(*args, _ptr(out), len(out), C.byref(n), _ptr(first), threads)
    if rc != BPE_OK:
        raise RuntimeError(f"bpe_split_docs failed: {rc}")
    return out[:n.value].copy(), first


def dedup_chunks(data: bytes, offsets, threads: int = 0):
    """Distinct chunks of (data, chunk start offsets) in order of first appearance, one copy
    per set bit of the multiplicity (host only).  Returns (data2, offsets2, weight_exp uint8,
    n_distinct)."""
    buf = np.frombuffer(data, dtype=np.uint8)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = np.empty(max(len(buf), 1), np.uint8)
    ooff = np.empty(max(len(off), 1), np.uint64)
    wexp = np.empty(max(len(off), 1), np.uint8)
    nb, nc, nd = _u64(0), _u64(0), _u64(0)
    rc = _lib.bpe_dedup_chunks(_ptr(buf) if len(buf) else None, len(buf), _ptr(off) if len(off) else None,
                               len(off), _ptr(out), _ptr(ooff), _ptr(wexp), C.byref(nb), C.byref(nc),
                               C.byref(nd), threads)
    if rc != BPE_OK:
        raise RuntimeError(f"bpe_dedup_chunks failed: {rc}")
    return out[:nb.value].tobytes(), ooff[:nc.value].copy(), wexp[:nc.value].copy(), int(nd.value)


class InvalidToken(Exception):
    """bpe_decode_batch met an id outside the vocab table; args[0] = its position."""


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_p)


class Engine:
    """One ctx = one GPU.  Thin, typed wrapper; all compute is in the library."""

    def __init__(self, device: int = 0):
        h = _p()
        rc = _lib.bpe_create(device, C.byref(h))
        if rc != BPE_OK:
            raise RuntimeError(
                "minbpe_amd needs an AMD GPU (gfx950): "
                + (_lib.bpe_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.bpe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- errors ------------------------------------------------------------
    def _check(self, rc):
        if rc == BPE_OK:
            return
        msg = (_lib.bpe_last_error(self._h) or b"").decode()
        if rc == BPE_E_EMPTY_STATS:
            # the reference's failure mode: max() over an empty dict (basic.py:35)
            raise ValueError("max() arg is an empty sequence")
        if rc == BPE_E_ARG:
            raise ValueError(msg)
        raise RuntimeError(f"libbpe_hip error {rc}: {msg}")

    # -- options -------------------------------------------------------------
    def set_option(self, name: str, value: int):
        self._check(_lib.bpe_set_option(self._h, name.encode(), int(value)))

    def set_stream(self, stream_handle: int):
        self._check(_lib.bpe_set_stream(self._h, _p(stream_handle)))

    # -- input -----------------------------------------------------------------
    @staticmethod
    def _offsets(offsets):
        if offsets is None:
            return None, 0
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        return off, len(off)

    def load_bytes(self, data, offsets=None, weight_exp=None):
        """weight_exp (uint8 per chunk): pairs inside chunk c count 2**weight_exp[c] times."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        off, n_off = self._offsets(offsets)
        self._keep = (buf, off)
        if weight_exp is None:
            self._check(_lib.bpe_load_bytes(self._h, _ptr(buf) if len(buf) else None, len(buf),
                                            _ptr(off), n_off))
            return
        w = np.ascontiguousarray(weight_exp, dtype=np.uint8)
        if off is None or len(w) != n_off:
            raise ValueError("weight_exp needs one entry per chunk offset")
        self._check(_lib.bpe_load_bytes_weighted(self._h, _ptr(buf) if len(buf) else None, len(buf),
                                                 _ptr(off), n_off, _ptr(w)))

    def load_ids(self, ids, offsets=None):
        arr = np.ascontiguousarray(ids, dtype=np.int32)
        off, n_off = self._offsets(offsets)
        self._check(_lib.bpe_load_ids(self._h, _ptr(arr) if len(arr) else None, len(arr),
                                      _ptr(off), n_off))

    # -- single steps ---------------------------------------------------------------
    def get_stats(self):
        """[( (a,b), count, first_pos )] in the reference's dict order."""
        n = _u64(0)
        self._check(_lib.bpe_get_stats(self._h, C.byref(n)))
        cap = max(int(n.value), 1)
        a = np.empty(cap, np.int32)
        b = np.empty(cap, np.int32)
        cnt = np.empty(cap, np.uint64)
        first = np.empty(cap, np.uint64)
        got = _u64(0)
        self._check(_lib.bpe_read_stats(self._h, _ptr(a), _ptr(b), _ptr(cnt), _ptr(first), cap,
                                        C.byref(got)))
        k = int(got.value)
        order = np.argsort(first[:k], kind="stable")
        return [((int(a[i]), int(b[i])), int(cnt[i]), int(first[i])) for i in order]

    def argmax(self):
        a, b, cnt = _i32(0), _i32(0), _u64(0)
        self._check(_lib.bpe_argmax(self._h, C.byref(a), C.byref(b), C.byref(cnt)))
        return (a.value, b.value), cnt.value

    def merge(self, pair, idx):
        n = _u64(0)
        self._check(_lib.bpe_merge(self._h, int(pair[0]), int(pair[1]), int(idx), C.byref(n)))
        return n.value

    def __len__(self):
        n = _u64(0)
        self._check(_lib.bpe_len(self._h, C.byref(n)))
        return n.value

    def read_ids(self):
        n = len(self)
        out = np.empty(max(n, 1), np.int32)
        self._check(_lib.bpe_read_ids(self._h, _ptr(out), len(out)))
        return out[:n]

    def read_chunk_starts(self):
        n = len(self)
        out = np.empty(max(n, 1), np.uint64)
        got = _u64(0)
        self._check(_lib.bpe_read_chunk_starts(self._h, _ptr(out), len(out), C.byref(got)))
        return out[:got.value]

    # -- training ---------------------------------------------------------------------
    def train(self, num_merges: int, want_iter_ms: bool = False):
        """Returns dict(pairs, counts, lens, iter_ms, n_done); raises ValueError
        like the reference when the pair table runs empty (basic.py:35)."""
        nm = max(num_merges, 1)
        pairs = np.zeros(2 * nm, np.int32)
        counts = np.zeros(nm, np.uint64)
        lens = np.zeros(nm, np.uint64)
        ms = np.zeros(nm, np.float64) if want_iter_ms else None
        done = _i32(0)
        rc = _lib.bpe_train(self._h, num_merges, _ptr(pairs), _ptr(counts), _ptr(ms), _ptr(lens),
                            C.byref(done))
        d = done.value
        self.last_train = dict(
            pairs=list(map(tuple, pairs[:2 * d].reshape(d, 2).tolist())),
            counts=counts[:d].tolist(), lens=lens[:d].tolist(),
            iter_ms=None if ms is None else ms[:d].copy(), n_done=d)
        self._check(rc)
        return self.last_train

    # -- data-parallel stepping (minbpe_amd/dist.py drives these) ------------------------
    def dp_begin(self, num_merges, rank, nranks):
        self._check(_lib.bpe_dp_begin(self._h, num_merges, rank, nranks))

    def dp_buffers(self):
        """(table_ptr, table_count, delta_ptr, delta_count, tiekey_ptr): raw device
        pointers of the three all-reduce payloads (int32, int32, int64 x 2)."""
        t, d, k = _p(), _p(), _p()
        tc, dc = _u64(0), _u64(0)
        self._check(_lib.bpe_dp_buffers(self._h, C.byref(t), C.byref(tc), C.byref(d), C.byref(dc),
                                        C.byref(k)))
        return t.value, tc.value, d.value, dc.value, k.value

    def dp_table_ready(self):
        self._check(_lib.bpe_dp_table_ready(self._h))

    def dp_select(self, i):
        self._check(_lib.bpe_dp_select(self._h, i))

    def dp_merge(self, i):
        self._check(_lib.bpe_dp_merge(self._h, i))

    def dp_apply(self, i):
        self._check(_lib.bpe_dp_apply(self._h, i))

    def dp_poll(self, i):
        """(pair, count, local_len, status) of iteration i; blocks until reported."""
        a, b, st = _i32(0), _i32(0), _i32(0)
        cnt, ln = _u64(0), _u64(0)
        self._check(_lib.bpe_dp_poll(self._h, i, C.byref(a), C.byref(b), C.byref(cnt), C.byref(ln),
                                     C.byref(st)))
        return (a.value, b.value), cnt.value, ln.value, st.value

    def dp_end(self):
        self._check(_lib.bpe_dp_end(self._h))

    # -- RCCL called from inside the library ---------------------------------------------
    @staticmethod
    def comm_available() -> bool:
        """librccl can be loaded in this process (bpe_comm_available)."""
        return bool(_lib.bpe_comm_available())

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = np.zeros(128, np.uint8)
        rc = _lib.bpe_comm_unique_id(_ptr(buf))
        if rc != BPE_OK:
            raise RuntimeError("bpe_comm_unique_id failed: " + (_lib.bpe_last_error(None) or b"").decode())
        return buf.tobytes()

    def comm_init(self, rank: int, nranks: int, uid: bytes):
        buf = np.frombuffer(uid, np.uint8)
        self._check(_lib.bpe_comm_init(self._h, rank, nranks, _ptr(buf)))

    def comm_destroy(self):
        self._check(_lib.bpe_comm_destroy(self._h))

    def dp_train(self, num_merges: int):
        """Sharded training with the collectives issued by the library (after comm_init).
        Same result dict as train(); lens are global."""
        nm = max(num_merges, 1)
        pairs = np.zeros(2 * nm, np.int32)
        counts = np.zeros(nm, np.uint64)
        lens = np.zeros(nm, np.uint64)
        done = _i32(0)
        rc = _lib.bpe_dp_train(self._h, num_merges, _ptr(pairs), _ptr(counts), _ptr(lens), C.byref(done))
        d = done.value
        self.last_train = dict(
            pairs=list(map(tuple, pairs[:2 * d].reshape(d, 2).tolist())),
            counts=counts[:d].tolist(), lens=lens[:d].tolist(), iter_ms=None, n_done=d)
        self._check(rc)
        return self.last_train

    def dp_train_cb(self, num_merges: int, rank: int, nranks: int, allreduce):
        """Sharded training (the same loop as dp_train) with the collectives handed to `allreduce(ptr, count, dtype,
        op, stream)`: all-reduce `count` elements at device address `ptr` in place across the ranks (dtype DT_INT32 /
        DT_INT64, op OP_SUM / OP_MIN), ordered after the work already on HIP stream `stream`; raise to fail.
        Same result dict as train(); lens are global."""
        nm = max(num_merges, 1)
        pairs = np.zeros(2 * nm, np.int32)
        counts = np.zeros(nm, np.uint64)
        lens = np.zeros(nm, np.uint64)
        done = _i32(0)
        raised = []

        def _cb(_user, buf, count, dtype, op, stream):
            try:
                allreduce(buf, count, dtype, op, stream)
                return 0
            except BaseException as e:  # (must not propagate through the C frames)
                raised.append(e)
                return 1

        cb = ALLREDUCE_FN(_cb)
        rc = _lib.bpe_dp_train_cb(self._h, num_merges, rank, nranks, C.cast(cb, C.c_void_p), None, _ptr(pairs),
                                  _ptr(counts), _ptr(lens), C.byref(done))
        d = done.value
        self.last_train = dict(
            pairs=list(map(tuple, pairs[:2 * d].reshape(d, 2).tolist())),
            counts=counts[:d].tolist(), lens=lens[:d].tolist(), iter_ms=None, n_done=d)
        if raised:
            raise raised[0]
        self._check(rc)
        return self.last_train

    # -- encoding ----------------------------------------------------------------------
    def encode_batch(self, pairs, merge_ids, data, offsets=None):
        """pairs: (M,2) int32 in priority order; merge_ids: (M,) int32 or None.
        Returns (ids ndarray int32, out_offsets ndarray uint64 of n_chunks+1)."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        off, n_off = self._offsets(offsets)
        n_chunks = n_off if off is not None else 1
        pm = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1)
        M = len(pm) // 2
        mi = None if merge_ids is None else np.ascontiguousarray(merge_ids, dtype=np.int32)
        out = np.empty(max(len(buf), 1), np.int32)
        oo = np.zeros(n_chunks + 1, np.uint64)
        n_out = _u64(0)
        self._check(_lib.bpe_encode_batch(self._h, _ptr(pm) if M else None, _ptr(mi), M,
                                          _ptr(buf) if len(buf) else None, len(buf), _ptr(off),
                                          n_off, _ptr(out), _ptr(oo), C.byref(n_out)))
        return out[:n_out.value], oo

    # -- decoding ----------------------------------------------------------------------
    def encode_batch_resident(self, pairs, merge_ids, d_bytes: int, n: int, d_offsets: int, n_chunks: int,
                              d_ids_out: int, d_out_offsets: int) -> int:
        """encode_batch with the batch and its outputs in HBM: d_bytes (n bytes), d_offsets (n_chunks uint64),
        d_ids_out (room for n int32), d_out_offsets (n_chunks + 1 uint64) are device addresses (e.g. torch
        tensor.data_ptr()).  Returns the number of tokens written."""
        p = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1)
        mids = None if merge_ids is None else np.ascontiguousarray(merge_ids, dtype=np.int32)
        total = _u64(0)
        self._check(_lib.bpe_encode_batch_resident(
            self._h, _ptr(p) if len(p) else None, None if mids is None else _ptr(mids), len(p) // 2,
            C.c_void_p(d_bytes), n, C.c_void_p(d_offsets), n_chunks, C.c_void_p(d_ids_out), C.c_void_p(d_out_offsets),
            C.byref(total)))
        return int(total.value)

    def decode_set_vocab(self, blob: bytes, offsets):
        """Vocab table: entry i = blob[offsets[i]:offsets[i+1]]; stays resident in HBM."""
        buf = np.frombuffer(blob, dtype=np.uint8)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(_lib.bpe_decode_set_vocab(self._h, _ptr(buf) if len(buf) else None, _ptr(off),
                                              len(off) - 1))

    def decode_batch(self, ids, doc_offsets=None):
        """ids: int32 table indices.  Returns the concatenated bytes, or (bytes, byte offset of
        each token position in doc_offsets).  An index outside the table raises
        InvalidToken(position)."""
        arr = np.ascontiguousarray(ids, dtype=np.int32)
        nb, bad = _u64(0), _u64(0)
        rc = _lib.bpe_decode_batch(self._h, _ptr(arr) if len(arr) else None, len(arr), C.byref(nb),
                                   C.byref(bad))
        if rc == BPE_E_ARG and bad.value != 0xFFFFFFFFFFFFFFFF:
            raise InvalidToken(int(bad.value))
        self._check(rc)
        out = np.empty(max(nb.value, 1), np.uint8)
        if doc_offsets is None:
            self._check(_lib.bpe_decode_read(self._h, _ptr(out), len(out), None, 0, None))
            return out[:nb.value].tobytes()
        doff = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        boff = np.zeros(len(doff), np.uint64)
        self._check(_lib.bpe_decode_read(self._h, _ptr(out), len(out), _ptr(doff), len(doff), _ptr(boff)))
        return out[:nb.value].tobytes(), boff

    # -- measurement ----------------------------------------------------------------
    def prof_reset(self):
        self._check(_lib.bpe_prof_reset(self._h))

    def train_stats(self):
        """dict(dense, sparse, index_builds, slots, lean, deferred, chained, steps, selections) of the last train()
        (bpe_train_stats_ex): passes by kind, merges done by lean iterations / chain steps, merges handed back to the
        general path, merges that needed no selection of their own, chain steps, chain steps that selected; slot_ids = ids per
        slot the stream ended in (1024, or 256 once it was re-packed for sparse passes); fused_steps = chain steps that were
        one launch (k_step.hip)."""
        out = np.zeros(11, np.uint64)
        self._check(_lib.bpe_train_stats_ex(self._h, _ptr(out), 11))
        return dict(dense=int(out[0]), sparse=int(out[1]), index_builds=int(out[2]), slots=int(out[3]),
                    lean=int(out[4]), deferred=int(out[5]), chained=int(out[6]), steps=int(out[7]),
                    selections=int(out[8]), slot_ids=int(out[9]), fused_steps=int(out[10]))

    def prof_read(self):
        k = len(PROF_KINDS)
        ms = np.zeros(k, np.float64)
        launches = np.zeros(k, np.uint64)
        nbytes = np.zeros(k, np.uint64)
        self._check(_lib.bpe_prof_read(self._h, _ptr(ms), _ptr(launches), _ptr(nbytes)))
        return {name: dict(ms=float(ms[i]), launches=int(launches[i]), alg_bytes=int(nbytes[i]))
                for i, name in enumerate(PROF_KINDS)}
