#!/bin/bash
# the literal path's kernels: parity of the histogram variants, their timing at three depths of the 1 GB input, and a
# kernel trace of 256 recount iterations (mode=0) cut per kernel
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "recount_histogram_variants or long_runs_cross or get_stats_argmax_merge" 2>&1 | tail -5
REPS=${REPS:-5} timeout -k 5 500 python tools/k1_experiment.py regex1g "${DEPTHS:-64,512,1024}" "${VARIANTS:-1,3}" > gpurun_out/r6_k1_experiment.jsonl 2> gpurun_out/r6_k1_experiment.err; echo "k1 rc=$?"
cut -c1-260 gpurun_out/r6_k1_experiment.jsonl; tail -3 gpurun_out/r6_k1_experiment.err
rm -rf gpurun_out/m0_prof
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/m0_prof -o run -- python tools/train_n.py regex1g ${M0_MERGES:-256} mode=0 ${M0_OPTS} > gpurun_out/m0_prof.log 2>&1; echo "prof rc=$?"
db=$(ls gpurun_out/m0_prof/*/*.db gpurun_out/m0_prof/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db > gpurun_out/r6_mode0_kernel_stats.csv && head -14 gpurun_out/r6_mode0_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/m0_prof
