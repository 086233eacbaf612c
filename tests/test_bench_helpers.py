"""CPU: the pure helpers bench.py builds its line from -- the full-length invariants, the comparison
with committed oracle digests, the library source hash that gates the PMC traffic file."""
import json

import pytest
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from helpers import checkpoint_digests  # noqa: E402


def _res(pairs, counts, lens):
    return {"pairs": pairs, "counts": counts, "lens": lens}


def test_invariants_accept_a_consistent_run_and_reject_broken_ones():
    good = _res([(1, 2), (3, 3), (4, 5)], [10, 8, 8], [90, 85, 77])  # a == b removes <= count
    assert bench.invariants(good, 100)["len_drop_equals_count_and_counts_monotone"]
    assert not bench.invariants(_res([(1, 2)], [10], [91]), 100)["len_drop_equals_count_and_counts_monotone"]
    assert not bench.invariants(_res([(1, 2), (3, 4)], [5, 6], [95, 89]), 100)["len_drop_equals_count_and_counts_monotone"]
    assert not bench.invariants(_res([(3, 3)], [4], [95]), 100)["len_drop_equals_count_and_counts_monotone"]


def test_parity_report_against_committed_digests(monkeypatch):
    pairs = [(i, i + 1) for i in range(40)]
    counts = [100 - i for i in range(40)]
    lens = [1000 - 10 * i for i in range(40)]
    digs = checkpoint_digests(pairs, counts, lens, 8)
    entry = {"bytes": 123, "seed": 9, "data_sha256": "ab", "done": 40, "step": 8,
             "digests": [list(d) for d in digs]}
    monkeypatch.setattr(bench, "golden_entry", lambda name: entry)
    wl = {"bytes": 123, "seed": 9}
    rep = bench.parity_report("x", wl, "ab", None, _res(pairs, counts, lens))
    assert rep["equal"] is True and rep["merges_checked"] == 40
    bad_counts = list(counts)
    bad_counts[20] += 1
    rep = bench.parity_report("x", wl, "ab", None, _res(pairs, bad_counts, lens))
    assert rep["equal"] is False and rep["first_bad_checkpoint"] == 24
    # another input (sha differs): nothing is claimed
    rep = bench.parity_report("x", wl, "cd", None, _res(pairs, counts, lens))
    assert rep["golden"] is None and rep["equal"] is None


def test_pmc_profile_is_tied_to_library_sources():
    """bench.py attaches profiles/r2_<workload>_pmc.json only when its recorded hash equals the hash of
    the sources being run; the file must carry what that needs."""
    h = bench.source_hash()
    assert len(h) == 16 and h == bench.source_hash()
    with open(os.path.join(ROOT, "profiles", "r2_regex1g_pmc.json")) as f:
        pmc = json.load(f)
    assert pmc["workload"] == "regex1g" and pmc["launches"] > 0 and pmc["hbm_bytes_total"] > 0
    assert len(pmc["source_hash"]) == 16
    cal = pmc["calibration_k_widen"]  # reads n bytes, writes 4n: the x2 fetch correction holds
    n = cal["n_input_bytes"] * cal["calls"]
    assert abs(cal["fetch_bytes_x2"] / n - 1.0) < 0.02 and abs(cal["write_bytes"] / (4 * n) - 1.0) < 0.03


def test_round4_pmc_profile_matches_the_sources_and_its_own_check():
    """The committed round-4 profile is of the library sources in the tree (so bench.py attaches it), the hash leaves
    the host-only translation units out, and the profile's self-check holds: k_load_count reads n + 8 B per chunk and
    writes 4n bytes per train (32 B x the size-weighted request counters; profiles/r4_pmc_calibration.json)."""
    csrc = os.path.join(ROOT, "minbpe_amd", "csrc")
    for f in bench.HOST_ONLY_SOURCES:
        assert os.path.exists(os.path.join(csrc, f)), f
    with open(os.path.join(ROOT, "profiles", "r4_regex1g_pmc.json")) as f:
        pmc = json.load(f)
    if pmc["source_hash"] != bench.source_hash():  # (mid-development: bench.py then prints the algorithmic figure, labelled)
        pytest.skip("device sources changed after the committed PMC pass: tools/gpu_final.sh makes a new one")
    assert pmc["trains"] == 3 and pmc["merges"] == 3 * 31744 and pmc["launches"] > 0
    first = pmc["check_on_the_first_pass"]
    n, chunks = first["n_input_bytes"] * first["calls"], 170_679_779 * first["calls"]
    assert abs(first["read_bytes"] / (n + 8 * chunks) - 1.0) < 0.03
    assert abs(first["write_bytes"] / (4 * n) - 1.0) < 0.01
    # per train on both sides (round 3's line set per-launch bytes against per-merge time)
    per_train = pmc["all_kernels_hbm_bytes_total"] / pmc["trains"]
    assert 1.5e12 < per_train < 3e12
    for name in ("r4_cfg2_pmc.json", "r4_encode_pmc.json"):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            assert json.load(f)["source_hash"] == bench.source_hash(), name


def test_round6_pmc_profiles_match_the_sources_and_their_own_check():
    """The committed round-6 profiles (bench.PROFILE_ROUND) are of the library sources in the tree -- so bench.py attaches
    their traffic --, the headline's self-check holds (k_load_count reads n + 8 B per chunk and writes 4n per train),
    kernels of the 256-id slot geometry are told apart ("@256"), the profile names its dominant kernel with that kernel's
    OWN fraction of the HBM peak and its atomics rate against a timed ceiling, and the committed bench line carries
    the parity verdict where the driver keeps it (the first 120 characters of config.workload)."""
    assert bench.PROFILE_ROUND == "r6"
    path = os.path.join(ROOT, "profiles", "r6_regex1g_pmc.json")
    if not os.path.exists(path):
        pytest.skip("no round-6 PMC pass committed yet: tools/gpu_final_r6.sh makes one")
    with open(path) as f:
        pmc = json.load(f)
    if pmc["source_hash"] != bench.source_hash():  # (mid-development: bench.py then prints the algorithmic figure, labelled)
        pytest.skip("device sources changed after the committed PMC pass: tools/gpu_final_r6.sh makes a new one")
    assert pmc["trains"] == 3 and pmc["merges"] == 3 * 31744 and pmc["launches"] > 0
    first = pmc["check_on_the_first_pass"]
    n, chunks = first["n_input_bytes"] * first["calls"], 170_679_779 * first["calls"]
    assert abs(first["read_bytes"] / (n + 8 * chunks) - 1.0) < 0.03
    assert abs(first["write_bytes"] / (4 * n) - 1.0) < 0.01
    assert any(k.endswith("@256") for k in pmc["merge_kernels"])
    per_train = pmc["all_kernels_hbm_bytes_total"] / pmc["trains"]
    assert 0.5e12 < per_train < 1.6e12
    dk = pmc["dominant_kernel"]
    assert dk["name"] in pmc["kernel_table"] and 0 < dk["frac_of_hbm_peak"] < 1 and dk["avg_us"] > 0
    ra = pmc["roofline_atomics"]
    assert ra["peak"] > 0 and ra["achieved"] > 0 and abs(ra["frac"] - ra["achieved"] / ra["peak"]) < 1e-3
    bl = os.path.join(ROOT, "profiles", "r6_final_bench.json")
    if not os.path.exists(bl):
        pytest.skip("no round-6 bench line committed yet")
    with open(bl) as f:
        line = json.loads(f.readline())
    assert line["source_hash"] == bench.source_hash() and line["n_gpus"] == 1 and line["unit"] == "merges/s"
    assert line["roofline"]["traffic_source"].startswith("profiles/r6_regex1g_pmc.json")
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    # per train on both sides: bytes per train / pass time per train
    rf = line["roofline"]
    assert abs(rf["achieved"] - rf["traffic_per_train"] / (rf["pass_ms_per_train"] * 1e-3) / 1e9) < 0.02 * rf["achieved"]
    assert line["cpu_baseline"]["cores"] == 1 and "pyref.py" in line["cpu_baseline"]["sample"]
    assert line["cpu_baseline"]["c_port"]["kind"] == "port"
    assert len(line["config"]["workload"]) < 120
    assert line["config"]["workload"].startswith("parity 31744/31744 merges = oracle: True")
    assert line["config"]["parity_equal"] is True and line["value"] is not None
    assert line["roofline_atomics"]["peak"] == ra["peak"]


def test_parity_failures_finds_every_false_verdict_and_nothing_else():
    line = {"parity": {"equal": True, "goldens": [{"equal": True}]}, "config": {"parity_equal": True},
            "secondary": {"cfg2": {"parity": {"equal": None}}, "encode": {"parity": {"equal_oracle": True}}}}
    assert bench.parity_failures(line) == []
    line["secondary"]["encode"]["parity"]["equal_oracle"] = False
    line["parity"]["goldens"][0]["equal"] = False
    assert sorted(bench.parity_failures(line)) == ["parity.goldens.0.equal", "secondary.encode.parity.equal_oracle"]


def test_parity_report_walks_every_committed_golden_without_tripping():
    """The headline's parity check looks at every entry of big_golden.json that could be of its input -- the sharded
    jobs' entries (per-shard digests, no data_sha256) and the 3.9 GB one among them -- and claims nothing for an input
    whose bytes it does not know."""
    for name in ("regex1g", "basic1g", "cfg2"):
        wl = dict(bench.WORKLOADS[name])
        rep = bench.parity_report(name, wl, "0" * 64, None, _res([(1, 2)] * 4, [9, 8, 7, 6], [90, 80, 70, 60]))
        assert rep["golden"] is None and rep["equal"] is None and rep["merges_checked"] == 0
    with open(os.path.join(ROOT, "tests", "golden", "big_golden.json")) as f:
        g = json.load(f)
    for k in ("regex1g_dp2_w", "regex1g_dp4_w", "regex1g_dp8_w"):
        e = g[k]
        assert e["world"] == len(e["shard_sha256"]) == int(k[len("regex1g_dp"):-2]) and e["done"] == 31744
        assert e["shard_sha256"][0] == g["regex1g"]["data_sha256"]  # rank 0's shard is the single-GPU headline input
    assert g["regex3p9g_w"]["bytes"] == 3_900_000_000 and g["regex3p9g_w"]["done"] == 31744


def test_workloads_name_the_baseline_configs():
    w = bench.WORKLOADS
    assert w["regex1g"]["bytes"] == 1_000_000_000 and w["regex1g"]["vocab"] == 32000 and w["regex1g"]["chunked"]
    assert w["basic1g"]["bytes"] == 1_000_000_000 and not w["basic1g"]["chunked"]
    assert w["cfg2"]["bytes"] == 100_000_000 and w["cfg2"]["vocab"] == 4096
    assert w["regex1g_dedup"]["dedup"] and w["regex1g_dedup"]["seed"] == w["regex1g"]["seed"]


class _BenchDouble:
    """fake_engine.OracleEngine + the measurement plumbing bench.py calls: lets run_train_workload's assembly of the
    line (parity report, invariants, roofline with and without a PMC profile, secondary or not) run on the CPU"""

    def __init__(self):
        from fake_engine import OracleEngine
        self._e = OracleEngine()
        self.opts = {}

    def load_bytes(self, data, offsets=None, weight_exp=None):
        self._e.load_bytes(data, offsets, weight_exp)

    def train(self, n):
        return self._e.train(n)

    def set_option(self, k, v):
        self.opts[k] = v

    def prof_reset(self):
        pass

    def prof_read(self):
        return {k: {"ms": ms, "launches": 7, "alg_bytes": 7_000_000}
                for k, ms in (("widen", 0.1), ("pair_count", 0.2), ("argmax", 0.5), ("merge", 2.0), ("table", 0.4))}

    def train_stats(self):
        return {"dense": 1, "sparse": 2}


@pytest.mark.parametrize("secondary", [False, True])
@pytest.mark.parametrize("mode", [-1, 0])
def test_run_train_workload_assembles_its_line_on_a_test_double(secondary, mode):
    wl = dict(bytes=60_000, seed=5, vocab=256 + 40, chunked=True, desc="RegexTokenizer.train (test double)")
    r, data, offs, res = bench.run_train_workload("unit", wl, _BenchDouble(), 2, 1, lambda: None, lambda dt: dt, mode,
                                                  secondary=secondary)
    json.dumps(r)  # the line must serialise
    assert len(res["pairs"]) == 40 and r["invariants"]["len_drop_equals_count_and_counts_monotone"]
    assert r["parity"]["golden"] is None and r["parity"]["equal"] is None  # (not an input any golden knows)
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["traffic"] is None and rf["equivalent_work_GBps"] > 0
    if secondary:  # no PMC pass of this workload: no fraction is claimed
        assert rf["frac"] is None and rf["achieved"] is None and rf["achieved_kind"].startswith("none")
    else:          # the headline without one: the algorithmic figure, labelled
        assert rf["frac"] is not None and rf["achieved_kind"].startswith("algorithmic")
    assert r["whole_iteration"]["device_ms_per_train"] == pytest.approx(3.2) and r["whole_iteration"]["frac"] is None


def test_main_prints_one_conforming_line_at_n1_on_a_test_double(monkeypatch, capsys):
    """bench.main() at N = 1 (a small --bytes / --vocab, no secondaries) with the engine replaced by the test double:
    the keys the driver's contract names are there, the CPU baseline leg runs, the line is one JSON object."""
    import torch
    import minbpe_amd

    class Double(_BenchDouble):
        def __init__(self, device=0):
            super().__init__()

        def close(self):
            pass

    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(minbpe_amd, "Engine", Double)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--bytes", "50000", "--vocab", str(256 + 30), "--steps", "2", "--warmup", "1",
                                      "--cpu-iters", "3", "--cpu-bytes", "20000"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_FORCE_DP"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(out) == 1
    line = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["vs_baseline"] is None
    assert line["unit"] == "merges/s" and line["dtype"] == "int32" and line["data"] == "synthetic"
    assert line["value"] == pytest.approx(30 / (line["ms_per_step"] * 1e-3), rel=1e-3)
    assert "workload" in line["config"] and "model" not in line["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    cpu = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cpu) and cpu["kind"] == "port" and cpu["cores"] == 1
    # the figure is minbpe's pure-Python loop (north_star), the C port rides along
    assert "pyref.py" in cpu["sample"] and cpu["value"] > 0 and cpu["c_port"]["value"] > cpu["value"]
    assert len(line["config"]["workload"]) < 120 and line["config"]["workload"].startswith("parity")
    assert line["config"]["parity_equal"] is None and "parity_failures" not in line
    assert line["secondary"] == {}
