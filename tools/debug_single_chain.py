#!/usr/bin/env python3
"""bpe_train (single GPU) on the three-letter / zero-byte chunk corpus of the sharded tie test, by engine option."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from minbpe_amd import Engine

rng = np.random.default_rng(9)
chunks = [b" " + bytes(97 + rng.integers(0, 3, size=rng.integers(1, 6))) for _ in range(3000)]
data = b"".join(chunks)
offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
nm = 400
exp = oracle.train(data, nm, offs, raise_on_empty=False)
print("oracle merges", len(exp[0]), "bytes", len(data))
eng = Engine(0)
base = {"chain": 1, "chain_extend": 1, "lean": 1, "sparse": 1, "aa_sparse": 1, "fuse_load": 1, "chain_dense": 1, "lean_backoff": 1}
for opts in ({}, {"chain": 0}, {"chain_extend": 0}, {"lean": 2}, {"lean": 2, "lean_backoff": 0}, {"sparse": 2}, {"sparse": 0}, {"chain_dense": 0},
             {"fuse_load": 0}, {"lean": 0}):
    for k, v in {**base, **opts}.items():
        eng.set_option(k, v)
    eng.load_bytes(data, offs)
    try:
        res = eng.train(nm)
    except ValueError:
        res = eng.last_train
    bad = next((i for i in range(min(len(res["pairs"]), len(exp[0]))) if res["pairs"][i] != exp[0][i] or res["counts"][i] != exp[1][i]), None)
    print(opts, "n", len(res["pairs"]), "first_bad", bad, eng.train_stats())
    if bad is not None:
        lo = max(0, bad - 2)
        print("   exp", list(zip(exp[0][lo:bad + 3], exp[1][lo:bad + 3])))
        print("   got", list(zip(res["pairs"][lo:bad + 3], res["counts"][lo:bad + 3])))
