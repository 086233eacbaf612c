#!/bin/bash
# Round 2, call A: baseline of the round-1 kernels on the 1 GB / vocab-32000 target.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
nproc; grep -m1 "model name" /proc/cpuinfo
timeout -k 5 600 python -X faulthandler -m pytest tests -m gpu -x -q -k "not cfg2_all and not cfg3_shape" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
timeout -k 5 400 python tools/iter_profile.py regex1g > gpurun_out/iter_regex1g.json 2> gpurun_out/iter_regex1g.err; echo "iter rc=$?"; cat gpurun_out/iter_regex1g.err | tail -12
timeout -k 5 600 python bench.py --steps 1 --warmup 0 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_a.json; tail -3 gpurun_out/bench_a.err
rm -rf gpurun_out/prof_a
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_a -o run -- python bench.py --steps 1 --warmup 0 --secondary none --cpu-iters 0 > gpurun_out/prof_a.log 2>&1
echo "prof rc=$?"; ls gpurun_out/prof_a | head; 
DB=$(find gpurun_out/prof_a -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/prof_a_kernel_stats.csv; rm -rf gpurun_out/prof_a; fi
find gpurun_out/prof_a -name "*stats*.csv" 2>/dev/null | head
head -20 gpurun_out/prof_a_kernel_stats.csv
