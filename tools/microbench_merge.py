#!/usr/bin/env python3
"""A/B the merge pass on a fixed stream: three-pass vs single-pass look-back with
different poll back-offs.  Times come from hipEvents around the pass (profile=1).
usage: python tools/microbench_merge.py [n_bytes] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minbpe_amd
from minbpe_amd import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 22_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
data = minbpe_amd.synth_text(n, 5)
eng = Engine(0)
eng.set_option("profile", 1)


def run(label, merge, tune=1, pair=(1, 2)):
    eng.set_option("merge", merge)
    eng.set_option("lb_tune", tune)
    eng.load_bytes(data)
    eng.merge(pair, 300)  # warm
    ts = []
    for r in range(reps):
        eng.load_bytes(data)
        eng.prof_reset()
        eng.merge(pair, 300)
        p = eng.prof_read()["merge"]
        ts.append(p["ms"] * 1e3)
    ts = np.array(ts)
    nl = len(eng)
    print(f"{label:34s} n={n} -> {nl}: median {np.median(ts):7.1f} us  min {ts.min():7.1f} us  "
          f"phys {4*(n+nl)/np.median(ts)/1e3:7.1f} GB/s", flush=True)


for pair in ((1, 2), (101, 32)):
    print("pair", pair)
    run("three-pass", 0, pair=pair)
    run("lookback sleep x0", 1, 0, pair=pair)
    run("lookback sleep x1", 1, 1, pair=pair)
    run("lookback sleep x4", 1, 4, pair=pair)
    run("lookback sleep x16", 1, 16, pair=pair)
    run("lookback NO WAIT (wrong result)", 1, 0x100, pair=pair)
eng.set_option("lb_tune", 1)
