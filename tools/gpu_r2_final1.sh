#!/bin/bash
# Round 2, final validation, part 1: every GPU test, then the PMC + kernel-trace passes of the
# headline workload, then a kernel trace of the encode workload.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/pytest_gpu.log 2>&1
RC=$?; echo "pytest rc=$RC"; tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
if [ $RC -ne 0 ]; then echo "tests failed: stopping here"; exit 1; fi
bash tools/gpu_pmc.sh regex1g

rm -rf gpurun_out/prof_enc
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_enc -o run -- python bench.py --workload encode --steps 2 --warmup 1 --cpu-iters 0 > gpurun_out/prof_enc.log 2>&1; echo "enc rc=$?"
DB=$(find gpurun_out/prof_enc -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > gpurun_out/r2_encode_kernel_stats.csv && rm -rf gpurun_out/prof_enc
head -8 gpurun_out/r2_encode_kernel_stats.csv | cut -c1-150
