#!/usr/bin/env python3
"""bpe_load_bytes of the headline input (1 GB of text + 1.4 GB of chunk offsets) with and without the pinned staging ring
(option pinned_upload): wall time of the call, the stream synchronised."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minbpe_amd import Engine
data, offs, _ = bench.make_input(dict(bench.WORKLOADS["regex1g"]))
eng = Engine(0)
eng.load_bytes(data, offs)  # (allocations)
for rep in range(3):
    for opt in (1, 0):
        eng.set_option("pinned_upload", opt)
        t0 = time.perf_counter()
        eng.load_bytes(data, offs)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.load_bytes(data)
        dt1 = time.perf_counter() - t0
        print(json.dumps({"pinned_upload": opt, "text_and_offsets_s": round(dt, 4), "GBps": round((len(data) + 8 * len(offs)) / dt / 1e9, 1),
                          "text_only_s": round(dt1, 4), "text_only_GBps": round(len(data) / dt1 / 1e9, 1), "nproc": os.cpu_count()}), flush=True)
