#!/usr/bin/env python3
"""CPU model of a chain step's batch rule on a known merge sequence: how many steps a training run needs
as a function of the batch cap, and what ends a batch.

A batch = consecutive merges of the reference's order that (1) have a != b, (2) share no token with an earlier merge
of the batch (neither its a, b nor the id it creates), (3) [level] have the same count as the batch's first merge --
the engine also goes below the maximum while the next level is unambiguous, so "any" drops this condition: the two
variants bracket the engine --, (4) number at most K.  An a == b merge is a unit of its own (general path).

    python tools/batch_model.py --make seq.json      # the headline's merge sequence: the weighted oracle on the distinct
                                                     # chunks of the 1 GB input (4-5 minutes of one core)
    python tools/batch_model.py seq.json > profiles/r4_batch_model.json
seq.json = {"pairs": [[a, b], ...], "counts": [...]} (the oracle's output: oracle.train(...)).
"""
import json
import sys


def steps(pairs, counts, K, level, lo=0, hi=None):
    hi = len(pairs) if hi is None else hi
    i, n, why, sizes = lo, 0, {"cap": 0, "token": 0, "aa": 0, "level": 0, "end": 0}, {}
    while i < hi:
        a, b = pairs[i]
        n += 1
        if a == b:
            why["aa"] += 1
            sizes[0] = sizes.get(0, 0) + 1
            i += 1
            continue
        used = {a, b, 256 + i}
        j = i + 1
        while True:
            if j >= hi:
                why["end"] += 1
                break
            if j - i >= K:
                why["cap"] += 1
                break
            x, y = pairs[j]
            if x == y:
                why["aa"] += 1
                break
            if level and counts[j] != counts[i]:
                why["level"] += 1
                break
            if x in used or y in used:
                why["token"] += 1
                break
            used |= {x, y, 256 + j}
            j += 1
        sizes[j - i] = sizes.get(j - i, 0) + 1
        i = j
    return n, why, sizes


def make(path):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import minbpe_amd
    import oracle
    data = minbpe_amd.synth_text(1_000_000_000, 2)
    offs = minbpe_amd.split_offsets(data, 4)
    d2, o2, wt, _ = oracle.dedup(data, offs)
    del data, offs
    pairs, counts, lens = oracle.train(d2, 31744, o2, weights=wt)
    json.dump({"pairs": pairs, "counts": counts, "lens": lens}, open(path, "w"))


def main():
    if sys.argv[1] == "--make":
        return make(sys.argv[2])
    d = json.load(open(sys.argv[1]))
    pairs, counts = [tuple(p) for p in d["pairs"]], d["counts"]
    M = len(pairs)
    out = {"merges": M, "a_eq_b": sum(1 for a, b in pairs if a == b), "model": __doc__.split("\n\n")[1], "by_cap": {}}
    for K in (1, 2, 4, 8, 16, 32, 64):
        row = {}
        for name, level in (("same_level", True), ("any_level", False)):
            n, why, sizes = steps(pairs, counts, K, level)
            row[name] = {"steps": n, "merges_per_step": round(M / n, 2), "batch_ended_by": why,
                         "batch_sizes": {str(k): v for k, v in sorted(sizes.items())}}
        out["by_cap"][str(K)] = row
    # by phase of the run, cap 8 against cap 32
    edges = [0, 300, 1000, 2000, 4000, 8000, 16000, 24000, M]
    out["by_phase"] = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        out["by_phase"].append({"merges": [lo, hi], **{f"steps_cap{K}_{nm}": steps(pairs, counts, K, lv, lo, hi)[0]
                                                        for K in (8, 32) for nm, lv in (("same_level", True), ("any_level", False))}})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
