#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
BENCH_FORCE_DP=1 BENCH_DP_CHECK=0 timeout -k 5 300 python bench.py --steps 1 --warmup 0 --secondary none --cpu-iters 0 > gpurun_out/bench_dp1b.json 2> gpurun_out/bench_dp1b.err; echo "dp bench rc=$?"; cut -c1-2500 gpurun_out/bench_dp1b.json; grep -v "amdgpu.ids\|socket.cpp" gpurun_out/bench_dp1b.err | tail -5 | cut -c1-300
