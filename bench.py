#!/usr/bin/env python3
"""bench.py -- headline benchmark (BASELINE.json: "BPE merges/sec + pair-count
GB/s vs HBM roofline").

    python bench.py [--gpus N --steps K --warmup W]

Workload (N=1): BASELINE.json configs[1] -- BasicTokenizer.train on 100 MB of
synthetic UTF-8 (synth_text(100_000_000, seed=1)), vocab 4096 = 3840 merges, one
MI355X.  A "step" is one complete train() over the stream: widen the resident
bytes to ids, then 3840 x (pair statistics, arg-max with the reference's
tie-break, merge).  The bytes are uploaded once, before the timed region (the
PCIe-inclusive rate is noted in DESIGN.md, never reported as `value`).

Prints ONE JSON line on rank 0.  `roofline` is the dominant kernel class,
timed with hipEvents on the library's own stream during the timed steps;
`cpu_baseline` is the CPU oracle (a C port of the reference's loop, one thread)
on the first 50 iterations of the same stream, on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s)
HBM_COPY_GBPS = 6290.0  # measured float4 copy, same guide


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=100_000_000, help="stream size per GPU")
    ap.add_argument("--vocab", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", type=int, default=int(os.environ.get("BPE_MODE", "-1")),
                    help="-1 library default | 0 recount | 1 delta")
    ap.add_argument("--cpu-iters", type=int, default=50,
                    help="oracle iterations for cpu_baseline, ~10 s of one host core at 100 MB (0 = skip); "
                         "whether the GPU's first merges equal them is reported")
    args = ap.parse_args()

    import torch  # device sync + torch.distributed (RCCL) plumbing only
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    force_dp = os.environ.get("BENCH_FORCE_DP") == "1"  # exercise the sharded path on one GPU
    if world > 1 or force_dp:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import minbpe_amd
    from minbpe_amd import Engine

    num_merges = args.vocab - 256
    eng = Engine(local_rank)
    if args.mode >= 0:
        eng.set_option("mode", args.mode)
    if "BPE_MERGE" in os.environ:  # experiments: 0 three-pass | 1 single-pass look-back
        eng.set_option("merge", int(os.environ["BPE_MERGE"]))
    if world == 1 and not force_dp:
        # configs[1]: BasicTokenizer.train, one unchunked stream, one GPU
        data = minbpe_amd.synth_text(args.bytes, args.seed)
        eng.load_bytes(data)  # H2D once, outside the timed region
        step = lambda: eng.train(num_merges)
    else:
        # BasicTokenizer's single stream does not shard; N > 1 runs the
        # chunked (RegexTokenizer-style) training sharded by chunks, `bytes` per GPU
        # (weak scaling), with the two per-merge all-reduces over RCCL.  Chunks are
        # cut before every space/newline (a vectorised stand-in for the regex split,
        # which runs at 5 MB/s on the host and is not part of the timed path).
        import numpy as np
        from minbpe_amd.dist import GpuShard, TorchComm, train_sharded
        data = minbpe_amd.synth_text(args.bytes, args.seed + rank)
        arr = np.frombuffer(data, dtype=np.uint8)
        cut = np.flatnonzero((arr == 32) | (arr == 10)).astype(np.uint64)
        offs = np.unique(np.concatenate([np.zeros(1, np.uint64), cut]))
        eng.load_bytes(data, offs)
        from minbpe_amd.dist import init_native_comm
        comm = TorchComm()
        # default: the library issues its own RCCL all-reduces (bpe_dp_train); BPE_DIST=torch, or a
        # failed communicator set-up on any rank, falls back to the torch.distributed driver
        dist_path = "torch.distributed"
        if os.environ.get("BPE_DIST", "native") == "native" and init_native_comm(eng, comm):
            dist_path = "librccl (in-library loop)"
            step = lambda: eng.dp_train(num_merges)
        else:
            shard = GpuShard(eng, local_rank)
            step = lambda: train_sharded(shard, comm, num_merges)

    def barrier():
        if world > 1 or force_dp:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # one untimed step with events around every kernel class: the breakdown
    eng.set_option("profile", 2)
    eng.prof_reset()
    step()
    breakdown = eng.prof_read()
    # timed steps: events only around the dominant kernel (two records per iteration)
    eng.set_option("profile", 1)
    eng.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.prof_read()
    if world > 1 or force_dp:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # N > 1: the ranks must agree, and the sharded result must be the single-GPU result on the
    # concatenation of all shards (rank 0 re-trains it unsharded, outside the timed region).
    dp_check = None
    if world > 1 or force_dp:
        import hashlib
        digest = int.from_bytes(hashlib.sha256(repr((res["pairs"], res["counts"], res["lens"])).encode())
                                .digest()[:7], "big")
        lo = torch.tensor([digest], dtype=torch.int64, device="cuda")
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dp_check = {"ranks_agree": bool(lo.item() == hi.item()), "equals_single_gpu": None}
        if rank == 0 and os.environ.get("BENCH_DP_CHECK", "1") == "1":
            try:
                parts, offl, base = [], [], 0
                for r in range(world):
                    d = minbpe_amd.synth_text(args.bytes, args.seed + r)
                    a = np.frombuffer(d, dtype=np.uint8)
                    cuts = np.flatnonzero((a == 32) | (a == 10)).astype(np.uint64)
                    o = np.unique(np.concatenate([np.zeros(1, np.uint64), cuts]))
                    parts.append(d)
                    offl.append(o + np.uint64(base))
                    base += len(d)
                eng2 = Engine(local_rank)
                eng2.load_bytes(b"".join(parts), np.concatenate(offl))
                single = eng2.train(num_merges)
                eng2.close()
                dp_check["equals_single_gpu"] = bool(
                    single["pairs"] == res["pairs"] and single["counts"] == res["counts"]
                    and single["lens"] == res["lens"])
            except Exception as e:  # the bench line must still come out
                dp_check["equals_single_gpu"] = f"not checked: {type(e).__name__}: {e}"

    # Weak scaling: every rank runs num_merges merge passes over its own `bytes`.  The whole-job
    # aggregate is therefore merge passes summed over ranks (= merges/s at N=1); the plain rate of
    # the one sharded job is reported next to it as job_merges_per_s.
    merges_total = num_merges * args.steps * world
    value = merges_total / dt
    job_merges_per_s = num_merges * args.steps / dt

    # dominant kernel class by device time -> roofline (timed live in the timed region)
    hot = max(("pair_count", "merge", "widen"), key=lambda k: breakdown[k]["ms"])
    hp = prof[hot] if prof[hot]["launches"] else breakdown[hot]
    achieved = hp["alg_bytes"] / (hp["ms"] * 1e-3) / 1e9 if hp["ms"] > 0 else 0.0
    # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this
    # process; they come from the committed rocprofv3 --pmc passes of this same command
    # (profiles/, FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950).
    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "r1_final_cfg2_pmc_merge_slot.json")
    if (hot == "merge" and world == 1 and not force_dp and args.bytes == 100_000_000 and args.vocab == 4096
            and args.mode < 0 and "BPE_MERGE" not in os.environ and os.path.exists(pmc_file)):
        with open(pmc_file) as f:
            traffic = int(json.load(f)["hbm_bytes_per_launch"])
        traffic_src = "profiles/r1_final_cfg2_pmc_merge_slot.json"
    roofline = {
        "bound": "hbm", "kernel": {"merge": "k_merge_slot (merge + pair-table delta)",
                                   "pair_count": "k_pair_count", "widen": "k_widen"}[hot],
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "frac_of_measured_copy": round(achieved / HBM_COPY_GBPS, 4),  # vs the 6.29 TB/s float4 copy
        "traffic": traffic,
        "traffic_source": traffic_src,
        "launches": hp["launches"], "avg_launch_ms": round(hp["ms"] / max(hp["launches"], 1), 5),
        "alg_bytes_per_launch": hp["alg_bytes"] // max(hp["launches"], 1),
        "note": "achieved = algorithmic bytes (4 B per id read or written by get_stats+merge, "
                "SURVEY 8d) / hipEvent time; traffic = measured HBM bytes per launch",
    }
    # the two figures the metric names, over the whole timed region
    pc, mg = breakdown["pair_count"], breakdown["merge"]
    alg_bytes_step = pc["alg_bytes"] + mg["alg_bytes"]  # sum of B_i = 4(2N_i + N_{i+1})
    extra = {
        "pair_count_GBps": round(pc["alg_bytes"] / (pc["ms"] * 1e-3) / 1e9, 1) if pc["ms"] else None,
        "merge_GBps": round(mg["alg_bytes"] / (mg["ms"] * 1e-3) / 1e9, 1) if mg["ms"] else None,
        # whole-iteration algorithmic rate, wall clock of the timed region (includes every small kernel and gap)
        "iter_GBps_wall": round(alg_bytes_step * args.steps / dt / 1e9, 1),
        "device_ms_per_step": {k: round(v["ms"], 3) for k, v in breakdown.items()},
        "final_len": res["lens"][-1] if res["lens"] else len(data),
    }

    cpu_baseline = None
    if rank == 0 and args.cpu_iters > 0 and world == 1 and not force_dp:
        import oracle
        t0 = time.perf_counter()
        cp, _, _ = oracle.train(data, args.cpu_iters)
        ct = time.perf_counter() - t0
        cpu_baseline = {
            "value": round(args.cpu_iters / ct, 4), "unit": "merges/s", "cores": 1, "kind": "port",
            # parity at the bench's full size, reported rather than asserted so that the line always comes out
            "gpu_first_merges_equal": bool(cp == res["pairs"][:args.cpu_iters]),
            "sample": f"first {args.cpu_iters} iterations (get_stats+max+merge) of the same "
                      f"{args.bytes}-byte stream, oracle/bpe_oracle.c, single thread",
        }

    if rank == 0:
        print(json.dumps({
            "metric": "BPE merges/sec", "value": round(value, 2), "unit": "merges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": (f"BasicTokenizer.train, {args.bytes} B synthetic UTF-8, " if world == 1 else
                                    f"chunked (RegexTokenizer-style) train sharded over {world} GPUs, "
                                    f"{args.bytes} B synthetic UTF-8 per GPU, ") +
                                   f"vocab {args.vocab} ({num_merges} merges), bit-exact vs oracle",
                       "mode": "recount" if args.mode == 0 else ("delta" if args.mode == 1 else "default"),
                       "parallelism": f"dp{world} (chunk shards; per-merge all-reduce of tie key + table deltas; "
                                                      f"collectives via {dist_path})"
                                      if (world > 1 or force_dp) else "single"},
            "job_merges_per_s": round(job_merges_per_s, 2),
            "value_definition": "merge passes per second summed over GPUs (each rank merges its own shard); "
                                "equals job_merges_per_s x n_gpus",
            "sharded_check": dp_check,
            "roofline": roofline, "cpu_baseline": cpu_baseline, **extra,
        }))
    eng.close()
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
