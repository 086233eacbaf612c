#!/usr/bin/env python3
"""debug build (-DBPE_DP_DEBUG): the selection state of every sharded chain step, from the SUM payload's padding"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from minbpe_amd import _native as native
from minbpe_amd.dist import _DevicePtr
rng = np.random.default_rng(9)
chunks = [b" " + bytes(97 + rng.integers(0, 3, size=rng.integers(1, 6))) for _ in range(3000)]
data = b"".join(chunks)
offs = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
nm = 400
exp = oracle.train(data, nm, offs, raise_on_empty=False)
eng = native.Engine(0)
eng.load_bytes(data, offs)
dev = torch.device("cuda", 0)
log = []
def ar(ptr, count, dtype, op, stream):
    t = torch.as_tensor(_DevicePtr(ptr, count, "<i8" if dtype == 1 else "<i4"), device=dev)
    torch.cuda.synchronize()
    h = t.cpu().numpy()
    if dtype == 1 and count == 98:
        ks = [int(x) for x in h[2:] if x != 0x7FFFFFFFFFFFFFFF]
        log.append(("MIN", int(h[0]), int(h[1]), ks))
    elif dtype == 0 and count > 64 and count != 65536 and count != 4 * 704 + 64:
        tl = h[count - 64:].astype(np.uint32)
        pr = lambda v: (int(v) >> 16, int(v) & 0xFFFF)
        K = int(tl[18])
        log.append(("SUM", dict(iter=int(tl[32]), mode=int(tl[33]), K=K, tl_n=int(tl[35]), M=int(tl[36]), skip=int(tl[37]),
                                defer=int(tl[38]), gap=int(tl[39]), batch=[pr(v) for v in tl[40:40 + min(K, 8)]],
                                chain=[pr(v) for v in tl[48:48 + min(int(tl[35]), 16)]])))
try:
    res = eng.dp_train_cb(nm, 0, 1, ar)
except ValueError:
    res = eng.last_train
bad = next((i for i in range(min(len(res["pairs"]), len(exp[0]))) if res["pairs"][i] != exp[0][i]), None)
print("first_bad", bad, "n", len(res["pairs"]))
for e in log:
    if e[0] == "SUM" and not (270 <= e[1]["iter"] <= 305):
        continue
    print(e)
print("expected 285..305:", list(enumerate(exp[0]))[285:305])
