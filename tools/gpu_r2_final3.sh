#!/bin/bash
# Round 2, final validation, part 3: smoke(), and the sharded (bpe_dp_* + RCCL) bench path at world 1.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log | cut -c1-300
BENCH_FORCE_DP=1 timeout -k 5 600 python bench.py --steps 1 --warmup 0 --secondary none --cpu-iters 0 > gpurun_out/bench_dp1.json 2> gpurun_out/bench_dp1.err; echo "dp bench rc=$?"; cut -c1-1500 gpurun_out/bench_dp1.json; tail -3 gpurun_out/bench_dp1.err | cut -c1-300
