#!/usr/bin/env python3
"""train the first N merges of a bench workload once (for rocprofv3 counter passes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minbpe_amd import Engine
name, nm = sys.argv[1], int(sys.argv[2])
data, offs, _ = bench.make_input(dict(bench.WORKLOADS[name]))
eng = Engine(0)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
eng.load_bytes(data, offs)
res = eng.train(nm)
print(len(res["pairs"]), res["lens"][-1])
eng.close()
