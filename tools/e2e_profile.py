#!/usr/bin/env python3
"""Where the wall time of RegexTokenizer().train(text_1GB, 32000) goes (host phases around the device's 0.6 s)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import minbpe_amd
from minbpe_amd import _native
from minbpe_amd.tokenizer import engine
wl = dict(bench.WORKLOADS["regex1g"])
data = bench.synth_cached(wl["bytes"], wl["seed"]) if hasattr(bench, "synth_cached") else minbpe_amd.synth_text(wl["bytes"], wl["seed"])
text = bytes(data).decode("utf-8")
out = {}
for rep in range(2):
    t = time.perf_counter(); d0 = text.encode("utf-8"); out["str_encode_s"] = round(time.perf_counter() - t, 3)
    t = time.perf_counter(); d = _native.utf8_encode(text); out["utf8_encode_s"] = round(time.perf_counter() - t, 3); out["equal"] = bytes(d[:1000000]) == d0[:1000000] and len(d) == len(d0); del d0
    t = time.perf_counter(); offs = _native.split_offsets(d, 4); out["split_s"] = round(time.perf_counter() - t, 3)
    eng = engine()
    t = time.perf_counter(); eng.load_bytes(d, offs); out["load_bytes_s"] = round(time.perf_counter() - t, 3)
    t = time.perf_counter(); res = eng.train(31744); out["train_s"] = round(time.perf_counter() - t, 3)
    t = time.perf_counter()
    merges, vocab = {}, {i: bytes([i]) for i in range(256)}
    for i, pair in enumerate(res["pairs"]):
        merges[pair] = 256 + i
        vocab[256 + i] = vocab[pair[0]] + vocab[pair[1]]
    out["dicts_s"] = round(time.perf_counter() - t, 3)
    tok = minbpe_amd.RegexTokenizer(); tok.dedup = False
    t = time.perf_counter(); tok.train(text, 32000); out["whole_call_s"] = round(time.perf_counter() - t, 3)
    out["threads"] = os.cpu_count()
    print(json.dumps(out), flush=True)
