#!/usr/bin/env python3
"""Pair-count (get_stats) kernels on a fixed stream: GB/s of ids read (4 B each), on the byte
stream and mid-training (after 512 and 3840 merges of cfg2: vocab 768 / 4096)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minbpe_amd
from minbpe_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
data = minbpe_amd.synth_text(n, 1)
eng = Engine(0)
eng.set_option("profile", 2)
out = []


def measure(label):
    ref = None
    for k1, name in ((0, "one L2 atomic per position"), (1, "LDS cache, 16 Ki 8-byte slots, 4 probes"),
                     (2, "default (dense 16-bit LDS table on bytes, else as k1=1)"),
                     (3, "as 2, 32 Ki exact 4-byte slots for general streams")):
        eng.set_option("k1", k1)
        ts = []
        for r in range(reps):
            eng.prof_reset()
            res = eng.argmax()
            ts.append(eng.prof_read()["pair_count"]["ms"])
        ref = ref or res
        assert res == ref, (res, ref)
        t = float(np.median(ts))
        m = len(eng)
        out.append({"stream": label, "ids": m, "k1": k1, "kernel": name, "median_us": round(t * 1e3, 1),
                    "GBps": round(4 * m / t / 1e6, 1)})
        print(out[-1], flush=True)


eng.load_bytes(data)
measure("bytes")
for nm in (512, 3840):
    eng.set_option("k1", 2)
    eng.load_bytes(data)
    eng.train(nm)
    measure(f"after {nm} merges")
print(json.dumps(out))
