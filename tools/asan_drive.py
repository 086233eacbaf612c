import ctypes as C, numpy as np, random, sys
lib = C.CDLL(sys.argv[1])
p, u64 = C.c_void_p, C.c_uint64
lib.bpe_synth_text.argtypes = [p, u64, u64]
lib.bpe_split.argtypes = [C.c_int, p, u64, p, u64, C.POINTER(u64), C.c_int]
lib.bpe_split_docs.argtypes = [C.c_int, p, u64, p, u64, p, u64, C.POINTER(u64), p, C.c_int]
lib.bpe_dedup_chunks.argtypes = [p, u64, p, u64, p, p, p, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.c_int]
def ptr(a): return a.ctypes.data_as(p) if len(a) else None
rng = random.Random(1)
def synth(n, seed):
    b = np.empty(n, np.uint8); assert lib.bpe_synth_text(ptr(b), n, seed) == 0; return b
def split(buf, which, threads):
    n = u64(0); out = np.empty(len(buf) + 8, np.uint64)
    rc = lib.bpe_split(which, ptr(buf), len(buf), ptr(out), len(out), C.byref(n), threads); assert rc == 0, rc
    return out[:n.value].copy()
cases = [synth(n, s) for n, s in [(0, 1), (1, 2), (7, 3), (1000, 4), (300_000, 5), (3_000_000, 6)]]
# adversarial bytes: random (invalid UTF-8 too), truncated sequences, long whitespace, apostrophes at the end
for L in (1, 2, 3, 5, 64, 4096):
    cases.append(np.frombuffer(bytes(rng.randrange(256) for _ in range(L)), np.uint8))
cases += [np.frombuffer(s, np.uint8) for s in [b"'", b"a'", b"'l", b"'\xc5", b"\xf0\x9f", b" \n \n  ", b"\xe2\x80", b"12345678901", b"x" * 100 + b"'"]]
for buf in cases:
    for which in (2, 4):
        for T in (1, 3, 16):
            offs = split(buf, which, T)
            # docs at arbitrary cut points
            cuts = sorted(set([0] + [rng.randrange(len(buf) + 1) for _ in range(5)])) if len(buf) else [0]
            doff = np.array([c for c in cuts if c <= len(buf)], np.uint64)
            n = u64(0); out = np.empty(len(buf) + len(doff) + 8, np.uint64); first = np.zeros(len(doff) + 1, np.uint64)
            rc = lib.bpe_split_docs(which, ptr(buf), len(buf), ptr(doff), len(doff), ptr(out), len(out), C.byref(n), ptr(first), T)
            assert rc == 0, rc
            ob = np.empty(max(len(buf), 1), np.uint8); oo = np.empty(max(len(offs), 1), np.uint64); ow = np.empty(max(len(offs), 1), np.uint8)
            nb, nc, nd = u64(0), u64(0), u64(0)
            rc = lib.bpe_dedup_chunks(ptr(buf), len(buf), ptr(offs), len(offs), ptr(ob), ptr(oo), ptr(ow), C.byref(nb), C.byref(nc), C.byref(nd), T)
            assert rc == 0, rc
print("asan drive ok", len(cases), "inputs")
