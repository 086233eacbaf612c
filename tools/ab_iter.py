#!/usr/bin/env python3
"""Per-iteration device time of train() (hipEvents between iterations), binned."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minbpe_amd
from minbpe_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
data = minbpe_amd.synth_text(n, 1)
eng = Engine(0)
eng.load_bytes(data)
eng.train(64)
res = eng.train(nm, want_iter_ms=True)
ms = res["iter_ms"] * 1e3
cnt = np.array(res["counts"])
same = np.array([a == b for a, b in res["pairs"]])
print("lib", os.environ.get("MINBPE_AMD_LIB", "default"), "total ms", ms.sum() / 1e3, "a==b merges", int(same.sum()))
edges = [0, 10, 100, 500, 1000, 2000, 3000, nm]
for lo, hi in zip(edges[:-1], edges[1:]):
    if lo >= nm: break
    sl = slice(lo, min(hi, nm))
    print(f"iters {lo:5d}-{hi:5d}: mean {ms[sl].mean():8.1f} us   a!=b mean {ms[sl][~same[sl]].mean():8.1f}   a==b mean {(ms[sl][same[sl]].mean() if same[sl].any() else 0):8.1f} (n={int(same[sl].sum())})  count~{int(cnt[sl].mean())}")
