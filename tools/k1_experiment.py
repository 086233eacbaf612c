#!/usr/bin/env python3
"""get_stats kernels of the literal path (mode=0) on the headline input, timed at several depths of training:
median device time of the pair-count pass per k1 variant (1 = k_pair_count_lds, the default of general streams; 3 =
k_pair_count_h32; profiles/r6_ap_pair_count_phases.patch adds 4 = its phased form and the timing-only 7 / 8 / 9: misses
dropped / and no flush / loads only), every variant's arg-max checked against the first one's."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from minbpe_amd import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "regex1g"
depths = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "64,512,1024").split(",")]
variants = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,3").split(",")]
reps = int(os.environ.get("REPS", 5))
data, offs, _ = bench.make_input(dict(bench.WORKLOADS[name]))
eng = Engine(0)
eng.set_option("mode", 0)
eng.set_option("profile", 2)
eng.load_bytes(data, offs)
done = 0
for d in depths:
    eng.set_option("k1", 2)
    if d < done:
        eng.load_bytes(data, offs)
        done = 0
    # (train() restarts from the bytes: train d merges in one go)
    eng.load_bytes(data, offs)
    eng.train(d)
    done = d
    ref = None
    for k1 in variants:
        eng.set_option("k1", k1)
        ts = []
        for r in range(reps):
            eng.prof_reset()
            try:
                res = eng.argmax()
            except ValueError:
                res = ((-1, -1), 0)
            ts.append(eng.prof_read()["pair_count"]["ms"])
        if ref is None:
            ref = res
        t = float(np.median(ts))
        m = len(eng)
        print(json.dumps({"workload": name, "after_merges": d, "ids": m, "k1": k1, "median_us": round(t * 1e3, 1),
                          "min_us": round(min(ts) * 1e3, 1), "GBps": round(4 * m / t / 1e6, 1),
                          "frac_of_8TBps": round(4 * m / t / 1e6 / 8000, 3), "argmax": [int(res[0][0]), int(res[0][1]), int(res[1])],
                          "same_argmax_as_first_variant": bool(res == ref) if k1 != 7 else None}), flush=True)
