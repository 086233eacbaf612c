#!/usr/bin/env python3
"""Per-iteration device time of train() on one of bench.py's workloads (hipEvents between
iterations), binned by iteration range, plus the per-class device-time breakdown.

    python tools/iter_profile.py regex1g [name=value engine options ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from minbpe_amd import Engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "regex1g"
wl = dict(bench.WORKLOADS[name])
data, offs, prep = bench.make_input(wl)
nm = int(os.environ.get("ITERS", wl["vocab"] - 256))
eng = Engine(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
eng.load_bytes(data, offs)
eng.train(min(64, nm))
eng.set_option("profile", 2)
eng.prof_reset()
res = eng.train(nm)
bd = eng.prof_read()
eng.set_option("profile", 0)
res = eng.train(nm, want_iter_ms=True)
ms = res["iter_ms"] * 1e3
if os.environ.get("ITER_NPY"):  # per-merge device time in full (float32 us): equal neighbours = one chain step's merges
    np.save(os.environ["ITER_NPY"], ms.astype(np.float32))
cnt = np.array(res["counts"])
lens = np.array(res["lens"])
same = np.array([a == b for a, b in res["pairs"]])
out = {"workload": name, "options": sys.argv[2:], "total_ms": round(float(ms.sum()) / 1e3, 1),
       "a_eq_b_merges": int(same.sum()), "passes": eng.train_stats(),
       "device_ms_by_class": {k: round(v["ms"], 1) for k, v in bd.items() if v["ms"]}, "bins": [],
       "first_iters": [[int(round(float(m))), int(c), int(l)] for m, c, l in zip(ms[:160], cnt[:160], lens[:160])]}
edges = [0, 10, 100, 300, 1000, 2000, 4000, 8000, 16000, 24000, nm]
for lo, hi in zip(edges[:-1], edges[1:]):
    if lo >= nm:
        break
    sl = slice(lo, min(hi, nm))
    ne, eq = ms[sl][~same[sl]], ms[sl][same[sl]]
    out["bins"].append({"iters": [lo, min(hi, nm)], "mean_us": round(float(ms[sl].mean()), 1),
                        "a_ne_b_us": round(float(ne.mean()), 1) if len(ne) else None,
                        "a_eq_b_us": round(float(eq.mean()), 1) if len(eq) else None, "n_eq": int(len(eq)),
                        "count": int(cnt[sl].mean()), "len_M": round(float(lens[sl].mean()) / 1e6, 1)})
print(json.dumps(out))
for b in out["bins"]:
    print(b, file=sys.stderr)
eng.close()
