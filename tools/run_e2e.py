#!/usr/bin/env python3
"""End to end through the drop-in classes (what a minbpe user calls): RegexTokenizer.train /
encode / decode on 100 MB of synthetic text, wall clock including the native pre-split, the
H2D upload and all host bookkeeping."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minbpe_amd
from minbpe_amd import RegexTokenizer, BasicTokenizer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
text = minbpe_amd.synth_text(n, 1).decode("utf-8")
out = {}
for cls in (RegexTokenizer, BasicTokenizer):
    tok = cls()
    tok.train(text[:1_000_000], 300)  # warm (context, allocations)
    t0 = time.time(); tok.train(text, 4096); t_train = time.time() - t0
    sample = text[:20_000_000]
    t0 = time.time(); ids = tok.encode(sample); t_enc = time.time() - t0
    t0 = time.time(); back = tok.decode(ids); t_dec = time.time() - t0
    assert back == sample
    import numpy as np
    arr = np.asarray(ids, dtype=np.int32)
    tok.decode_batch(arr[:1000])  # installs the vocab table
    t0 = time.time(); raw = tok.decode_batch(arr); t_decb = time.time() - t0
    assert raw == sample.encode("utf-8")
    out[cls.__name__] = dict(train_s=round(t_train, 3), merges=len(tok.merges), encode_20MB_s=round(t_enc, 3),
                             tokens=len(ids), decode_s=round(t_dec, 3),
                             decode_batch_s=round(t_decb, 4))
print(json.dumps(out))
