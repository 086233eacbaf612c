#!/usr/bin/env python3
"""What the resident stream looks like after a train() that ran out of pairs (ValueError) -- error-path check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from minbpe_amd import Engine
eng = Engine(0)
for data, offs, nm in ((b"ab", None, 12), (b"ab", np.array([0, 1], dtype=np.uint64), 12), (b"abcabcab", None, 12),
                       (b"abcabcab", np.array([0, 3, 6], dtype=np.uint64), 12), (b"xyxyxyxyzz" * 300, None, 40)):
    for fuse in (1, 0):
        eng.set_option("fuse_load", fuse)
        eng.load_bytes(data, offs)
        before = eng.read_ids().tolist()[:12]
        try:
            res = eng.train(nm)
            err = None
        except ValueError as e:
            res, err = eng.last_train, "ValueError"
        print(dict(n=len(data), offs=None if offs is None else offs.tolist(), fuse=fuse, done=res["n_done"], err=err,
                   lens=res["lens"][-2:], len_now=len(eng), ids=eng.read_ids().tolist()[:12], before=before,
                   starts=eng.read_chunk_starts().tolist()[:6]))
