#!/usr/bin/env python3
"""A/B of option chain_levels (a batch that goes on into a tied level below the maximum) on a GPT-4-split text: the
merges, counts and lengths must be identical with the option on and off; prints steps and time of both.
    python tools/ab_levels.py [bytes] [seed] [merges]"""
import json
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minbpe_amd
from minbpe_amd import Engine

nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nm = int(sys.argv[3]) if len(sys.argv) > 3 else 31744
data = minbpe_amd.synth_text(nbytes, seed)
offs = minbpe_amd.split_offsets(data, 4)
eng = Engine(0)
eng.load_bytes(data, offs)
out = {}
res = {}
runs = [(0, 1), (1, 1), (1, 0)] if os.environ.get("AB_NOLIST") else [(0, 1), (1, 1)]
for lv, use_list in runs:
    eng.set_option("chain_levels", lv)
    eng.set_option("chain_list", use_list)
    t0 = time.perf_counter()
    try:
        r = eng.train(nm)
    except Exception as e:  # noqa: BLE001
        out[f"levels{lv}"] = f"failed: {type(e).__name__}: {e}"
        break
    dt = time.perf_counter() - t0
    key = lv if use_list else 2
    res[key] = r
    out[f"levels{lv}_list{use_list}"] = {"s": round(dt, 4), "stats": eng.train_stats(), "merges": len(r["pairs"])}
if 2 in res and 0 in res:
    out["nolist_identical"] = bool(res[0]["pairs"] == res[2]["pairs"] and res[0]["counts"] == res[2]["counts"] and res[0]["lens"] == res[2]["lens"])
if 0 in res and 1 in res:
    same = res[0]["pairs"] == res[1]["pairs"] and res[0]["counts"] == res[1]["counts"] and res[0]["lens"] == res[1]["lens"]
    out["identical"] = bool(same)
    if not same:
        n = min(len(res[0]["pairs"]), len(res[1]["pairs"]))
        k = next((i for i in range(n) if res[0]["pairs"][i] != res[1]["pairs"][i] or res[0]["counts"][i] != res[1]["counts"][i]), n)
        out["first_difference"] = {"merge": k, "off": [res[0]["pairs"][k:k + 3], res[0]["counts"][k:k + 3]],
                                   "on": [res[1]["pairs"][k:k + 3], res[1]["counts"][k:k + 3]]}
print(json.dumps(out))
eng.close()
