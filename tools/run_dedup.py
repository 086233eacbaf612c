#!/usr/bin/env python3
"""Chunked training with and without chunk de-duplication on the same text:
same merges and counts required, timings of each leg printed as one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minbpe_amd
from minbpe_amd import _native as native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
nm = vocab - 256
data = native.synth_text(n, 1)
t0 = time.time(); offs = native.split_offsets(data, 4); t_split = time.time() - t0
eng = native.Engine(0)
eng.load_bytes(data[:1_000_000], offs[offs < 1_000_000]); eng.train(64)  # warm
t0 = time.time(); eng.load_bytes(data, offs); plain = eng.train(nm); t_plain = time.time() - t0
t0 = time.time(); d2, o2, w, nd = native.dedup_chunks(data, offs); t_dedup = time.time() - t0
t0 = time.time(); eng.load_bytes(d2, o2, w); ded = eng.train(nm); t_wtrain = time.time() - t0
same = plain["pairs"] == ded["pairs"] and plain["counts"] == ded["counts"]
print(json.dumps(dict(bytes=n, chunks=len(offs), distinct=nd, weighted_chunks=len(o2), weighted_bytes=len(d2),
                      merges=nm, split_s=round(t_split, 3), plain_upload_train_s=round(t_plain, 3),
                      dedup_host_s=round(t_dedup, 3), weighted_upload_train_s=round(t_wtrain, 3),
                      same_merges_and_counts=same, host_threads=os.cpu_count())))
sys.exit(0 if same else 1)
