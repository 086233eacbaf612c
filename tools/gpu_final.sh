#!/bin/bash
# Last GPU call of a round: all GPU tests (no -x: every failure is listed), smoke, the default
# bench line, the sharded path forced onto one GPU (with its single-GPU cross-check), and the
# end-to-end timings.  Every step is bounded; nothing reads stdin.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 150 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
timeout -k 5 60 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout -k 5 90 python bench.py > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_full.log
BENCH_FORCE_DP=1 timeout -k 5 60 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_dp.log 2>&1; echo "bench_dp rc=$?"; tail -1 gpurun_out/bench_dp.log
timeout -k 5 90 python tools/run_e2e.py > gpurun_out/e2e.log 2>&1; echo "e2e rc=$?"; tail -1 gpurun_out/e2e.log
