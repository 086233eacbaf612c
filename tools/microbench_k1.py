#!/usr/bin/env python3
"""Pair-count (get_stats) kernels on a fixed stream: GB/s of ids read (4 B each)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minbpe_amd
from minbpe_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
data = minbpe_amd.synth_text(n, 1)
eng = Engine(0)
eng.set_option("profile", 2)
eng.load_bytes(data)
ref = None
for k1, label in ((0, "one L2 atomic per position"), (1, "LDS hash cache"), (2, "dense 16-bit LDS table (byte stream)")):
    eng.set_option("k1", k1)
    eng.load_bytes(data)
    ts = []
    for r in range(reps):
        eng.prof_reset()
        pair, cnt = eng.argmax()
        ts.append(eng.prof_read()["pair_count"]["ms"])
    if ref is None: ref = (pair, cnt)
    assert (pair, cnt) == ref
    t = np.median(ts)
    print(f"k1={k1} {label:40s} n={n}: median {t*1e3:9.1f} us  {4*n/t/1e6:8.1f} GB/s", flush=True)
# after some merges (ids >= 256): general kernel only
eng.set_option("k1", 2); eng.set_option("mode", 1)
eng.load_bytes(data); eng.train(512)
for k1 in (0, 1):
    eng.set_option("k1", k1)
    ts = []
    for r in range(reps):
        eng.prof_reset(); res = eng.argmax(); ts.append(eng.prof_read()["pair_count"]["ms"])
    m = len(eng)
    print(f"k1={k1} after 512 merges n={m}: median {np.median(ts)*1e3:9.1f} us  {4*m/np.median(ts)/1e6:8.1f} GB/s  argmax={res}", flush=True)
