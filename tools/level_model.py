#!/usr/bin/env python3
"""CPU study for the next engine step: how many chain steps the headline run would need if a batch could go
on INTO A TIED LEVEL below the maximum, under a rule the device can check before it merges.

Replays the reference's merges on the DISTINCT chunks of the 1 GB input (weights = multiplicities; the same statistics
and order as the full chunk list, tests/test_dedup.py) with incremental pair counts, checks every merge and count
against the oracle's sequence, and at every step boundary evaluates three batch rules on the live pair table:

  engine   what k_chain_sel does today: the pairs tied at the maximum in order (prefix without a shared token, a != b),
           then, if the whole list was taken, the levels below while each has exactly ONE pair (no shared token)
  tied     the same, but a level with SEVERAL pairs may be entered when a conservative bound holds: every count(L, a_j)
           and count(b_j, R) of the batch's pairs is below the level (what one would check with row / column maxima)
  engine+list, free+list   the same two rules with the engine's two kinds of step: only a FULL step walks below its list's
           level; a step that leaves pairs of a level behind is followed by LIST steps that take what is left of that list
           and go no further (the engine: 8,401 steps, 4,799 of them FULL; the branch with the walk into tied levels: 7,177 /
           3,144)
  free1    "free" with at most ONE tied level per step (the walk ends with the first level of several pairs it enters: the
           first form of the branch's k_chain_sel)
  free     no bound at all: walk the levels from the top, inside a level the pairs in the reference's order (known here
           from the sequence; on the device: the index-based tie-break), stop at the first pair that shares a token with
           the batch or has a == b.  This IS exact -- a created pair reaches a level only by taking over, in place, a
           pair of that level that shares a token with the batch, where the walk stops anyway: tests/test_level_model.py
           pins it against the reference semantics -- so "free" is the rule to build, "tied" a weaker form of it
  free_s2, free_rx   (round 6, RULES=free,free_s2,free_rx) "free" with a weaker clash rule: only a pair that could OVERLAP a site
           of the batch stops the walk -- (x, y) with x a second token or y a first token of the batch (a chain a b d), and
           a == b; free_s2 still keeps first tokens distinct (the merge pass looks a pair up by its first token)

    python tools/level_model.py seq.json [cap ...] > profiles/r4_level_model.json     (seq.json: tools/batch_model.py --make)
"""
import json
import os
import sys
import time

from sortedcontainers import SortedList

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_chunks():
    import minbpe_amd
    import oracle
    data = minbpe_amd.synth_text(1_000_000_000, 2)
    offs = minbpe_amd.split_offsets(data, 4)
    d2, o2, wt, _ = oracle.dedup(data, offs)
    del data, offs
    o2 = [int(x) for x in o2] + [len(d2)]
    return [list(d2[o2[i]:o2[i + 1]]) for i in range(len(o2) - 1)], [int(w) for w in wt]


class Table:
    """pair -> count, with per-row / per-column multisets of counts (the maxima the device keeps or can scan for) and the
    chunks every pair occurs in"""

    def __init__(self):
        self.cnt = {}
        self.where = {}
        self.row = {}
        self.col = {}
        self.levels = SortedList()  # multiset of all counts

    def add(self, p, w, ci):
        old = self.cnt.get(p, 0)
        new = old + w
        self._move(p, old, new)
        if ci is not None:
            self.where.setdefault(p, set()).add(ci)

    def _move(self, p, old, new):
        a, b = p
        if old:
            self.row[a].remove(old)
            self.col[b].remove(old)
            self.levels.remove(old)
        if new:
            self.cnt[p] = new
            self.row.setdefault(a, SortedList()).add(new)
            self.col.setdefault(b, SortedList()).add(new)
            self.levels.add(new)
        else:
            self.cnt.pop(p, None)
            self.where.pop(p, None)

    def rowmax_excl(self, a, excl):
        """largest count in row a, leaving out one occurrence of each count in `excl`"""
        return _top_excl(self.row.get(a), excl)

    def colmax_excl(self, b, excl):
        return _top_excl(self.col.get(b), excl)


def _top_excl(sl, excl):
    if not sl:
        return 0
    excl = sorted(excl, reverse=True)
    i = len(sl) - 1
    for e in excl:
        if i >= 0 and sl[i] == e:
            i -= 1
    return sl[i] if i >= 0 else 0


def merge_chunk(t, chunk, w, ci, a, b, z):
    """rewrite one chunk, keeping the table current (counts only: `where` may keep stale chunk indices, harmless)"""
    out, i, n = [], 0, len(chunk)
    # remove all pairs of the old chunk, add all pairs of the new one: simple and exact
    for x, y in zip(chunk, chunk[1:]):
        t.add((x, y), -w, None)
    while i < n:
        if i + 1 < n and chunk[i] == a and chunk[i + 1] == b:
            out.append(z)
            i += 2
        else:
            out.append(chunk[i])
            i += 1
    for x, y in zip(out, out[1:]):
        t.add((x, y), w, ci)
    return out


def main():
    seq = json.load(open(sys.argv[1]))
    caps = [int(x) for x in sys.argv[2:]] or [8, 16]
    pairs, counts = [tuple(p) for p in seq["pairs"]], seq["counts"]
    M = len(pairs)
    t0 = time.time()
    chunks, wts = load_chunks()
    t = Table()
    for ci, (c, w) in enumerate(zip(chunks, wts)):
        for x, y in zip(c, c[1:]):
            t.add((x, y), w, ci)
    print(f"table built: {len(t.cnt)} pairs, {time.time() - t0:.0f} s", file=sys.stderr)

    def batch_at(i, cap, rule, list_level=None):
        """number of merges a step starting at merge i takes under `rule` (the table is in the state before merge i);
        list_level: the step is a LIST step -- it takes its pairs off the list of the pairs at that count and goes no
        further (rules ending in "+list" model the engine's two kinds of step; the others let every step walk on)"""
        a, b = pairs[i]
        if a == b:
            return 1
        top = counts[i]
        used = {a, b}
        firsts, seconds = {a}, {b}
        mode = None
        batch = [pairs[i]]
        j = i + 1
        level = top
        in_tied = False
        while j < M and len(batch) < cap:
            x, y = pairs[j]
            c = t.cnt.get((x, y), 0)
            if rule == "free_s2":    # a second token may be shared: (a, b), (c, b) never overlap and leave each other's counts alone
                clash = x in firsts or x in seconds or y in firsts
            elif rule == "free_rx":  # only a chain a b d stops the walk (and a == b): first tokens may be shared too
                clash = x in seconds or y in firsts
            elif rule == "free_e2":  # a batch shares EITHER second tokens OR first tokens (whichever it meets first)
                clash = x in seconds or y in firsts
                if not clash and x in firsts:
                    clash = mode == "s"
                    if not clash:
                        mode = "f"
                if not clash and y in seconds:
                    clash = mode == "f"
                    if not clash:
                        mode = "s"
            else:
                clash = x in used or y in used
            if x == y or clash or c != counts[j]:  # (c != counts[j]: a pair the batch creates or changes)
                break
            if c < level:  # a level below
                if list_level is not None or in_tied:
                    break
                if rule in ("free", "free1", "free_s2", "free_rx", "free_e2"):
                    if rule == "free1" and t.levels.count(c) > 1:
                        in_tied = True  # (one tied level per step: the walk ends with it)
                else:
                    # the level must be the next one down among the pairs outside the batch ...
                    k = len(t.levels) - 1 - len(batch)  # the batch's own counts sit on top
                    nxt = t.levels[k] if k >= 0 else 0
                    if nxt != c:
                        break  # (cannot happen if the sequence is consistent; keeps the model honest)
                    n_at = t.levels.count(c)
                    if n_at > 1:
                        if rule == "engine":
                            break
                        # tied: no created pair may reach this level
                        bc = [t.cnt[p] for p in batch]
                        ok = True
                        for (pa, pb) in batch:
                            # row pb = (b_j, R), column pa = (L, a_j); a batch pair itself may sit in those lines
                            ex_r = [t.cnt[p] for p in batch if p[0] == pb]
                            ex_c = [t.cnt[p] for p in batch if p[1] == pa]
                            if t.rowmax_excl(pb, ex_r) >= c or t.colmax_excl(pa, ex_c) >= c or t.cnt.get((pb, pa), 0) >= c:
                                ok = False
                                break
                        if not ok:
                            break
                    elif rule == "engine":
                        # a single pair at the level: the engine also needs every entry hidden behind the maxima of the rows it
                        # took from to rank below it -- implied here (the sequence says this pair IS next)
                        pass
                level = c
            batch.append((x, y))
            used |= {x, y}
            firsts.add(x)
            seconds.add(y)
            j += 1
        return len(batch)

    rules = tuple(os.environ.get("RULES", "engine,tied,free,free1,engine+list,free+list,free1+list").split(","))
    res = {f"{r}_cap{cap}": {"steps": 0, "next": 0, "by_phase": {}, "list": None, "full": 0} for r in rules for cap in caps}
    edges = [0, 300, 1000, 2000, 4000, 8000, 16000, 24000, M]
    for i in range(M):
        a, b = pairs[i]
        for key, st in res.items():
            if st["next"] == i:
                r, cap = key.split("_cap")
                if r.endswith("+list"):
                    # the engine's two kinds of step: a FULL step walks (rule), a LIST step only takes what is left of the
                    # list = the pairs at the count of the last pair the previous step took, while any are left
                    lv = st["list"] if (st["list"] is not None and t.cnt.get(pairs[i], 0) == st["list"]) else None
                    k = batch_at(i, int(cap), r[:-5], lv)
                    st["full"] += lv is None
                    last = counts[i + k - 1]
                    st["list"] = last if (i + k < M and counts[i + k] == last and pairs[i][0] != pairs[i][1]) else None
                else:
                    k = batch_at(i, int(cap), r)
                st["steps"] += 1
                st["next"] = i + k
                ph = max(e for e in edges[:-1] if e <= i)
                st["by_phase"][str(ph)] = st["by_phase"].get(str(ph), 0) + 1
        got = t.cnt.get((a, b), 0)
        if got != counts[i] or got != t.levels[-1]:
            raise SystemExit(f"merge {i}: pair {a, b} has count {got}, maximum {t.levels[-1]}, the oracle says {counts[i]}")
        z = 256 + i
        for ci in list(t.where.get((a, b), ())):
            c = chunks[ci]
            if any(c[q] == a and c[q + 1] == b for q in range(len(c) - 1)):
                chunks[ci] = merge_chunk(t, c, wts[ci], ci, a, b, z)
        if i % 2000 == 0:
            print(f"merge {i}: {time.time() - t0:.0f} s, " + ", ".join(f"{k} {v['steps']}" for k, v in res.items()), file=sys.stderr)
    out = {"merges": M, "rules": __doc__.split("\n\n")[2], "result": {k: {"steps": v["steps"], "merges_per_step": round(M / v["steps"], 2),
                                                                          **({"full_steps": v["full"]} if k.split("_cap")[0].endswith("+list") else {}),
                                                                          "steps_by_first_merge_of_phase": v["by_phase"]} for k, v in res.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
