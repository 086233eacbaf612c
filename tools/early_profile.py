#!/usr/bin/env python3
"""Device time of the first merges of regex1g (the atomic-heavy dense passes), with engine options."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from minbpe_amd import Engine
nm = 1000
data, offs, _ = bench.make_input(dict(bench.WORKLOADS["regex1g"]))
for opts in ([], ["rep_max=5"], ["rep_max=2"], ["rep_max=0"]):
    eng = Engine(0)
    for kv in opts + sys.argv[1:]:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.load_bytes(data, offs)
    eng.train(32)
    try:
        res = eng.train(nm, want_iter_ms=True)
    except Exception as e:
        res = eng.last_train
    ms = res["iter_ms"] * 1e3
    row = {"options": opts + sys.argv[1:], "merges": len(ms)}
    for lo, hi in ((0, 10), (10, 100), (100, 300), (300, 1000)):
        if hi <= len(ms):
            row[f"{lo}-{hi}"] = round(float(ms[lo:hi].mean()), 1)
    print(json.dumps(row), flush=True)
    eng.close()
