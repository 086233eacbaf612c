#!/usr/bin/env python3
"""A/B of engine option sets on one of bench.py's workloads, ONE upload for all of them:
    python tools/ab_opts.py regex1g "" "chain_kcap=8" "chain_kcap=8 chain_prefetch=0"
Per set: wall time of a plain train() (best of REPS), the step statistics, the kernel-class breakdown (profile 2),
per-phase device time (want_iter_ms) and the parity verdict against the committed golden digests of the workload
(bench.parity_report).  Options are reset to OPT_RESET's values between sets ("name=value ...": the defaults of the
options any set touches)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from minbpe_amd import Engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "regex1g"
sets = sys.argv[2:] or [""]
wl = dict(bench.WORKLOADS[name])
if os.environ.get("AB_BYTES"):
    wl["bytes"] = int(os.environ["AB_BYTES"])
if os.environ.get("AB_SEED"):
    wl["seed"] = int(os.environ["AB_SEED"])
reps = int(os.environ.get("REPS", 2))
nm = int(os.environ.get("ITERS", wl["vocab"] - 256))
data, offs, prep = bench.make_input(wl)
sha = hashlib.sha256(data).hexdigest()
eng = Engine(0)
eng.load_bytes(data, offs)
reset = dict(kv.split("=") for kv in os.environ.get("OPT_RESET", "").split())
touched = {}
first = None
edges = [0, 300, 1000, 2000, 4000, 8000, 16000, 24000, nm]
for s in sets:
    for k, v in touched.items():
        eng.set_option(k, int(v))
    for kv in s.split():
        k, v = kv.split("=")
        touched.setdefault(k, reset.get(k, "0"))
        eng.set_option(k, int(v))
    out = {"options": s, "workload": name, "bytes": wl["bytes"], "merges": nm}
    try:
        eng.train(min(nm, 2048))  # warm
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            res = eng.train(nm)
            best = min(best, time.perf_counter() - t0)
        out["s_per_train"] = round(best, 4)
        out["merges_per_s"] = round(nm / best, 1)
        out["stats"] = eng.train_stats()
        if nm == wl["vocab"] - 256:
            out["parity"] = bench.parity_report(name, wl, sha, offs, res)
        if first is None:
            first = res
        else:
            out["same_as_first_set"] = bool(res["pairs"] == first["pairs"] and res["counts"] == first["counts"]
                                            and res["lens"] == first["lens"])
        if not os.environ.get("AB_FAST"):
            eng.set_option("profile", 2)
            eng.prof_reset()
            eng.train(nm)
            bd = eng.prof_read()
            eng.set_option("profile", 0)
            out["device_ms_by_class"] = {k: round(v["ms"], 1) for k, v in bd.items() if v["ms"]}
            r2 = eng.train(nm, want_iter_ms=True)
            ms = np.asarray(r2["iter_ms"]) * 1e3
            ph = []
            for lo, hi in zip(edges[:-1], edges[1:]):
                if lo >= nm:
                    break
                sl = ms[lo:min(hi, nm)]
                nsteps = int((np.diff(sl) != 0).sum()) + 1
                ph.append({"merges": [lo, min(hi, nm)], "ms": round(float(sl.sum()) / 1e3, 1), "steps": nsteps})
            out["phases"] = ph
    except Exception as e:  # noqa: BLE001
        out["failed"] = f"{type(e).__name__}: {e}"
    print(json.dumps(out), flush=True)
eng.close()
