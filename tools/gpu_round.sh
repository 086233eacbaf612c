#!/bin/bash
# One gpurun call: parity tests, smoke, a short bench, a rocprof kernel trace.
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --bytes ${BENCH_BYTES:-10000000} --vocab ${BENCH_VOCAB:-1024} --steps 2 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench_small.log
