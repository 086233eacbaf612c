#!/usr/bin/env python3
"""Where the sharded chain steps first leave the oracle's merge list, by engine option (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from minbpe_amd import _native as native
import test_gpu_parity as T

rng = np.random.default_rng(9)
chunks = [b" " + bytes(97 + rng.integers(0, 3, size=rng.integers(1, 6))) for _ in range(3000)]
data = b"".join(chunks)
offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
nm = 400
exp = oracle.train(data, nm, offs, raise_on_empty=False)
print("oracle merges", len(exp[0]))
variants = [(3, ()), (1, ()), (2, ()), (3, (("dp_kcap", 1),)), (3, (("chain_extend", 0),)), (3, (("aa_sparse", 0),)),
            (3, (("sparse", 0),)), (3, (("sparse", 2),)), (3, (("chain", 0),)), (3, (("lean_backoff", 0),))]

for world, opts in variants:
    try:
        out, errs, stats = T._chain_ranks(native, chunks, nm, world, opts)
    except BaseException as e:
        print(world, opts, "EXC", type(e).__name__, str(e)[:200])
        continue
    res = out[0]
    bad = next((i for i in range(min(len(res["pairs"]), len(exp[0]))) if res["pairs"][i] != exp[0][i] or res["counts"][i] != exp[1][i]), None)
    agree = all(o["pairs"] == res["pairs"] for o in out)
    print(world, opts, "n", len(res["pairs"]), "first_bad", bad, "ranks_agree", agree, stats[0])
    if bad is not None:
        lo = max(0, bad - 3)
        print("   exp", list(zip(exp[0][lo:bad + 3], exp[1][lo:bad + 3])))
        print("   got", list(zip(res["pairs"][lo:bad + 3], res["counts"][lo:bad + 3])))
