#!/usr/bin/env python3
"""configs[2]-sized run: chunked (RegexTokenizer-style) training on 1 GB of
synthetic text, vocab 32000, one GPU.  The stream (4 GB of ids) is far beyond the
256 MiB Infinity Cache, so this is the run whose GB/s is an HBM number.
Chunks are cut before every space/newline with numpy (the host `regex` split runs
at 5 MB/s and is not part of the measured path).  Checks the size-independent
invariant len[i-1] - len[i] == count[i] for every a != b merge."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minbpe_amd
from minbpe_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
t0 = time.time()
data = minbpe_amd.synth_text(n, 2)
arr = np.frombuffer(data, dtype=np.uint8)
cut = np.flatnonzero((arr == 32) | (arr == 10)).astype(np.uint64)
offs = np.unique(np.concatenate([np.zeros(1, np.uint64), cut]))
print(f"generated {n} bytes, {len(offs)} chunks in {time.time()-t0:.1f}s", flush=True)
eng = Engine(0)
t0 = time.time()
eng.load_bytes(data, offs)
print(f"upload {time.time()-t0:.2f}s", flush=True)
eng.set_option("profile", 1)
eng.prof_reset()
nm = vocab - 256
t0 = time.time()
res = eng.train(nm)
dt = time.time() - t0
prof = eng.prof_read()["merge"]
lens = np.array([n] + res["lens"], dtype=np.int64)
cnt = np.array(res["counts"], dtype=np.int64)
same = np.array([a == b for a, b in res["pairs"]])
removed = lens[:-1] - lens[1:]
assert np.all(removed[~same] == cnt[~same]), "len[i-1]-len[i] != count[i] for some a != b merge"
assert np.all(removed[same] <= cnt[same]) and np.all(removed > 0)
print(json.dumps({"workload": f"chunked train, {n} B synthetic, vocab {vocab}", "merges": len(res["pairs"]),
                  "seconds": round(dt, 3), "merges_per_s": round(len(res["pairs"]) / dt, 1),
                  "merge_alg_GBps": round(prof["alg_bytes"] / (prof["ms"] * 1e-3) / 1e9, 1),
                  "merge_ms_total": round(prof["ms"], 1), "final_len": int(lens[-1]),
                  "invariant_len_drop_equals_count": True}))
