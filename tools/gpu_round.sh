#!/bin/bash
# One gpurun call: parity tests, smoke, benches, optional rocprof kernel trace.
# Every step is bounded by `timeout -k`; nothing here reads stdin.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
if [ -z "$SKIP_TESTS" ]; then
timeout -k 5 ${TEST_TIMEOUT:-500} python -X faulthandler -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout -k 5 200 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
fi
if [ -n "$BENCH_SMALL" ]; then
timeout -k 5 300 python bench.py --bytes 10000000 --vocab 1024 --steps 2 --warmup 1 > gpurun_out/bench_small.log 2>&1; echo "bench_small rc=$?"; tail -2 gpurun_out/bench_small.log
fi
if [ -n "$BENCH_FULL" ]; then
timeout -k 5 600 python bench.py ${BENCH_ARGS} > gpurun_out/bench_full.log 2>&1; echo "bench_full rc=$?"; tail -2 gpurun_out/bench_full.log
fi
if [ -n "$BENCH_RECOUNT" ]; then
timeout -k 5 600 python bench.py --mode 0 --steps 1 --warmup 1 > gpurun_out/bench_recount.log 2>&1; echo "bench_recount rc=$?"; tail -2 gpurun_out/bench_recount.log
fi
if [ -n "$PROF" ]; then
rm -rf gpurun_out/prof
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o run -- python bench.py --steps 1 --warmup 0 --cpu-iters 0 ${PROF_ARGS} > gpurun_out/prof_bench.log 2>&1
echo "prof rc=$?"; tail -2 gpurun_out/prof_bench.log
ls -la gpurun_out/prof
fi
