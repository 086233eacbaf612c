#!/bin/bash
# One gpurun call: parity tests, smoke, a short bench, a rocprof kernel trace.
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -30 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --bytes ${BENCH_BYTES:-10000000} --vocab ${BENCH_VOCAB:-1024} --steps 2 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench_small.log
if [ -n "$BENCH_FULL" ]; then
timeout 900 python bench.py --steps 1 --warmup 1 2>&1 | tail -3 | tee gpurun_out/bench_full.log
fi
if [ -n "$PROF" ]; then
rm -rf gpurun_out/prof; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o run -- python bench.py --bytes ${PROF_BYTES:-100000000} --vocab ${PROF_VOCAB:-512} --steps 1 --warmup 0 --cpu-iters 0 > gpurun_out/prof_bench.log 2>&1
tail -3 gpurun_out/prof_bench.log
find gpurun_out/prof -name "*kernel_stats*" | head; cat $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | head -30
fi
