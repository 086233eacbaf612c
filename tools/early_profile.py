#!/usr/bin/env python3
"""Device time of the first merges of a bench workload (the dense passes) under several engine option
sets, and that every set produces the same merges:
    python tools/early_profile.py regex1g 1000 "dense_prefetch=1" "dense_prefetch=1 rep_max=5"
(the empty set = library defaults always runs first)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from minbpe_amd import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "regex1g"
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
data, offs, _ = bench.make_input(dict(bench.WORKLOADS[name]))
ref = None
for opts in [""] + sys.argv[3:]:
    eng = Engine(0)
    for kv in opts.split():
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.load_bytes(data, offs)
    eng.train(32)
    try:
        res = eng.train(nm, want_iter_ms=True)
    except Exception as e:
        res = eng.last_train
    ms = res["iter_ms"] * 1e3
    row = {"options": opts, "merges": len(ms), "total_ms": round(float(ms.sum()) / 1e3, 2)}
    for lo, hi in ((0, 10), (10, 100), (100, 300), (300, 1000)):
        if hi <= len(ms):
            row[f"{lo}-{hi}"] = round(float(ms[lo:hi].mean()), 1)
    key = (res["pairs"], res["counts"], res["lens"])
    if ref is None:
        ref = key
    row["same_merges_as_default"] = key == ref
    print(json.dumps(row), flush=True)
    eng.close()
