"""CPU, world_size 2, gloo: bench.py's N > 1 leg run end to end on test doubles -- argument handling, per-rank input,
the sharded training through dist.train_sharded (the per-merge protocol, BPE_DIST=steps) over tests/cpu_shard.py, the
barrier / max-over-ranks timing, the line's value (what all ranks processed per second) next to the job's own rate,
ranks_agree, equals_single_gpu, the oracle-golden lookup, the CPU baseline, one JSON line from rank 0.

Everything that stands in for the GPU is patched in HERE (the product and bench.py have no CPU path): torch.cuda calls
become no-ops, "cuda" tensors become CPU tensors, the process group is gloo, minbpe_amd.Engine is an oracle-backed
double and dist.GpuShard the numpy shard.  What this guards is bench.py's own Python on a path no single-GPU box runs."""
import io
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_job_golden(world, nbytes, seed, merges):
    """what tests/golden/gen_big_golden.py::main_sharded makes for the 1 GB shards, for shards of a few kB: the plain
    oracle on the shards back to back (every shard split on its own)"""
    import hashlib
    import numpy as np
    import minbpe_amd
    import oracle
    from helpers import checkpoint_digests
    parts, offl, shas, base = [], [], [], 0
    for r in range(world):
        d = minbpe_amd.synth_text(nbytes, seed + r)
        o = np.ascontiguousarray(minbpe_amd.split_offsets(d, 4), dtype=np.uint64)
        shas.append(hashlib.sha256(d).hexdigest())
        parts.append(d)
        offl.append(o + np.uint64(base))
        base += len(d)
    pairs, counts, lens = oracle.train(b"".join(parts), merges, np.concatenate(offl))
    return {"bytes": nbytes, "seed": seed, "merges": merges, "chunked": True, "weighted": True, "world": world,
            "shard_sha256": shas, "done": len(pairs), "step": 8, "digests": checkpoint_digests(pairs, counts, lens, 8)}


def _worker(rank, world, port, argv, out_q, tmpdir, golden=None, path="steps"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", BPE_DIST=path, BENCH_DP_CHECK="1", TMPDIR=tmpdir)
    import torch.distributed as dist
    import minbpe_amd
    import minbpe_amd.dist as mdist
    from cpu_shard import CpuShard
    from fake_engine import OracleEngine

    class Double(OracleEngine):  # + the measurement plumbing the sharded leg touches
        def __init__(self, device=0):
            super().__init__()

        def set_option(self, k, v):
            pass

        def prof_reset(self):
            pass

        def prof_read(self):
            return {"merge": {"ms": 1.5, "launches": 3, "alg_bytes": 1000}}

        def train_stats(self):
            return {"steps": 0}

        def close(self):
            pass

        # BPE_DIST=native: dist.init_native_comm hands the library's communicator id around (rank 0's 128 bytes reach
        # every rank), then the whole sharded loop is ONE call -- here the per-merge protocol on the numpy shard
        def comm_available(self):
            return True

        def comm_unique_id(self):
            return bytes(range(1, 129))

        def comm_init(self, r, w, raw):
            assert (r, w) == (rank, world) and raw == bytes(range(1, 129))
            self.comm_ready = True

        def dp_train(self, n):
            assert self.comm_ready
            return mdist.train_sharded(CpuShard(self._data, self._offs), mdist.TorchComm(), n)

    # -- the GPU's stand-ins --------------------------------------------------------------------------------
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    real_tensor = torch.tensor
    torch.tensor = lambda *a, **k: real_tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, **k: real_init("gloo", rank=rank, world_size=world)
    minbpe_amd.Engine = Double
    mdist.GpuShard = lambda eng, local_rank: CpuShard(eng._data, eng._offs)
    import bench
    if golden is not None:  # a committed oracle answer for this very job (as regex1g_dp2_w is for the 1 GB shards)
        real_entry = bench.golden_entry
        bench.golden_entry = lambda name: golden if name == f"regex1g_dp{world}_w" else real_entry(name)
    sys.argv = ["bench.py"] + argv
    buf, real_stdout = io.StringIO(), sys.stdout
    sys.stdout = buf
    try:
        bench.main()
        out_q.put((rank, "ok", buf.getvalue()))
    except SystemExit as e:  # (bench.py exits 3 after printing its line when a comparison with the oracle came out False)
        out_q.put((rank, "ok" if not e.code else f"exit {e.code}", buf.getvalue()))
    except BaseException as e:  # noqa: BLE001 (the parent reports it)
        import traceback
        out_q.put((rank, "failed", f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        sys.stdout = real_stdout
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("golden,path", [("none", "steps"), ("right", "steps"), ("wrong", "steps"), ("right", "native")])
def test_sharded_bench_leg_runs_on_doubles_at_world_2(tmp_path, golden, path):
    world = 2
    argv = ["--gpus", str(world), "--bytes", "40000", "--vocab", str(256 + 24), "--steps", "1", "--warmup", "1",
            "--cpu-iters", "3", "--cpu-bytes", "20000"]
    g = None
    if golden != "none":
        g = _tiny_job_golden(world, 40000, 2, 24)  # (bench.py's regex1g workload: seed 2 + rank)
        if golden == "wrong":
            g["digests"][-1][1] = "0" * 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, q, str(tmp_path), g, path)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, status, text in outs:
        # a job whose merges differ from the oracle's is reported by rank 0's exit code as well as in its line
        assert status == ("exit 3" if (golden == "wrong" and rank == 0) else "ok"), f"rank {rank}: {status}: {text}"
    assert outs[1][2].strip() == ""  # rank 1 prints nothing
    lines = [ln for ln in outs[0][2].splitlines() if ln.strip()]
    assert len(lines) == 1  # ONE JSON line from rank 0
    line = json.loads(lines[0])
    # (N > 1: `value` counts a merge once per shard it is applied to, and the unit says so)
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["unit"] == "shard-merges/s"
    assert ("parity_failures" in line) == (golden == "wrong")
    assert line["steps"] == 1 and line["warmup"] == 1 and line["higher_is_better"] is True
    # the units ALL ranks processed per second = world x the job's own rate
    assert line["value"] == pytest.approx(world * line["job_merges_per_s"], rel=1e-3)
    assert line["job_merges_per_s"] == pytest.approx(24 / (line["ms_per_step"] * 1e-3), rel=1e-3)
    chk = line["sharded_check"]
    assert chk["ranks_agree"] is True and chk["equals_single_gpu"] is True
    if golden == "none":
        assert chk["oracle"].startswith("no committed oracle digest")  # (40 kB shards: no committed golden knows them)
    else:  # the job's merges against the oracle's answer for the shards back to back
        assert chk["oracle"]["merges_checked"] == 24 and chk["oracle"]["equal"] is (golden == "right")
        assert ("first_bad_checkpoint" in chk["oracle"]) == (golden == "wrong")
    assert line["config"]["parallelism"].startswith("dp2") and "sharded over 2 GPUs" in line["config"]["workload"]
    assert ("bpe_dp_train" in line["config"]["parallelism"]) == (path == "native")
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["frac"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["compare_with"] == "job_merges_per_s"
