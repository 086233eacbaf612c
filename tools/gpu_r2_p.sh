#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 600 python -X faulthandler -m pytest tests -m gpu -q --maxfail=12 -k "dp or sharded or lockstep" > gpurun_out/pytest_dp.log 2>&1
echo "pytest dp rc=$?"; tail -25 gpurun_out/pytest_dp.log | cut -c1-300
BENCH_FORCE_DP=1 timeout -k 5 300 python bench.py --steps 1 --warmup 1 --secondary none > gpurun_out/bench_dp.json 2> gpurun_out/bench_dp.err; echo "bench dp rc=$?"; cut -c1-1200 gpurun_out/bench_dp.json; tail -3 gpurun_out/bench_dp.err
