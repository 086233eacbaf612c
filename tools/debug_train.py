#!/usr/bin/env python3
"""debug: train a file (or synthetic text) with engine options, compare with the oracle, print statistics"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from minbpe_amd import Engine
path, nm = sys.argv[1], int(sys.argv[2])
data = open(path, "rb").read()
exp = oracle.train(data, nm)
for opts in sys.argv[3:]:
    eng = Engine(0)
    for kv in opts.split(","):
        if kv:
            k, v = kv.split("=")
            eng.set_option(k, int(v))
    eng.load_bytes(data)
    try:
        res = eng.train(nm)
        ok = res["pairs"] == exp[0] and res["counts"] == exp[1] and res["lens"] == exp[2]
        first_bad = next((i for i in range(len(res["pairs"])) if res["pairs"][i] != exp[0][i] or res["counts"][i] != exp[1][i]), None)
        print(opts or "default", "ok" if ok else f"MISMATCH at {first_bad}", eng.train_stats())
    except Exception as e:
        print(opts or "default", "ERROR", e)
        print("   oracle merges 5..14:", list(zip(exp[0][5:15], exp[1][5:15])))
    eng.close()
