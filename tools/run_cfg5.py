#!/usr/bin/env python3
"""configs[4] shape: batch-encode 1M documents on one GPU with a trained vocabulary
(tiktoken's cl100k ranks are not available offline -- SURVEY 8c -- so the merges are
our own: trained here on 50 MB, vocab 16384).  Documents = synthetic text cut at
blank lines; chunks cut at spaces/newlines with numpy (the host `regex` split is not
the measured path).  Reports docs/s and tokens/s for bpe_encode_batch (H2D of the
bytes and D2H of the ids included), and checks a sample against the CPU oracle."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minbpe_amd
from minbpe_amd import Engine
import oracle

n_docs_target = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng = Engine(0)
train = minbpe_amd.synth_text(50_000_000, 4)
arr = np.frombuffer(train, np.uint8)
toffs = np.unique(np.concatenate([np.zeros(1, np.uint64), np.flatnonzero((arr == 32) | (arr == 10)).astype(np.uint64)]))
eng.load_bytes(train, toffs)
t0 = time.time(); res = eng.train(16384 - 256); print(f"trained {len(res['pairs'])} merges in {time.time()-t0:.2f}s", flush=True)
pairs = np.array(res["pairs"], np.int32)

# ~1M documents: synth text has a blank line every ~100 tokens (~590 B)
nbytes = int(n_docs_target * 600)
text = minbpe_amd.synth_text(min(nbytes, 2_000_000_000), 5)
a = np.frombuffer(text, np.uint8)
n_docs = int(np.count_nonzero((a[:-1] == 10) & (a[1:] == 10))) + 1
offs = np.unique(np.concatenate([np.zeros(1, np.uint64), np.flatnonzero((a == 32) | (a == 10)).astype(np.uint64)]))
print(f"{len(text)} bytes, {n_docs} documents, {len(offs)} chunks", flush=True)
eng.encode_batch(pairs, None, text[:1_000_000], offs[offs < 1_000_000])  # warm
ts = []
for r in range(3):
    t0 = time.time()
    ids, out_off = eng.encode_batch(pairs, None, text, offs)
    ts.append(time.time() - t0)
dt = min(ts)
# spot-check against the oracle on the first 2 MB
m = int(np.searchsorted(offs, 2_000_000))
exp_ids, exp_off = oracle.encode(res["pairs"], text[:int(offs[m])], offs[:m])
assert np.array_equal(ids[:len(exp_ids)], exp_ids) and np.array_equal(out_off[:m], exp_off[:m])
print(json.dumps({"workload": f"encode {n_docs} docs ({len(text)} B, {len(offs)} chunks), vocab 16384",
                  "seconds": round(dt, 3), "docs_per_s": round(n_docs / dt), "tokens_per_s": round(len(ids) / dt),
                  "MB_per_s": round(len(text) / dt / 1e6, 1), "tokens": int(len(ids)),
                  "oracle_check": "first 2 MB identical"}))
