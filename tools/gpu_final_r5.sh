#!/bin/bash
# Round-5 evidence, one gpurun call per PART (1: GPU suite + the PMC passes; 2: the bench line, phases, the 3.9 GB run).
# Every step is bounded by `timeout -k`; outputs under gpurun_out/r5_*; nothing reads stdin.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
export TAG=r5
case "$1" in
1)
  timeout -k 5 600 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r5_final_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_final_pytest_gpu.log | cut -c1-200
  bash tools/gpu_pmc.sh regex1g 2>&1 | tail -6 | cut -c1-400
  SKIP_KT=1 bash tools/gpu_pmc.sh basic1g 2>&1 | tail -4 | cut -c1-400
  SKIP_KT=1 bash tools/gpu_pmc.sh encode 2>&1 | tail -4 | cut -c1-400
  ;;
2)
  cp gpurun_out/r5_regex1g_pmc.json gpurun_out/r5_basic1g_pmc.json gpurun_out/r5_encode_pmc.json profiles/ 2>/dev/null
  timeout -k 5 900 python bench.py > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err; echo "bench rc=$?"; cut -c1-700 gpurun_out/r5_final_bench.json; tail -2 gpurun_out/r5_final_bench.err | cut -c1-300
  ITER_NPY=gpurun_out/r5_final_regex1g_iter_us.npy timeout -k 5 300 python tools/iter_profile.py regex1g > gpurun_out/r5_final_regex1g_iter_profile.json 2> gpurun_out/r5_iter.err; echo "iter rc=$?"
  rm -rf gpurun_out/r5_prof; timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r5_prof -o run -- python tools/train_n.py regex1g 31744 > gpurun_out/r5_prof.log 2>&1; echo "prof rc=$?"
  db=$(ls gpurun_out/r5_prof/*/*.db gpurun_out/r5_prof/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > gpurun_out/r5_final_regex1g_kernel_stats_one_train.csv && python tools/rocpd_phases.py $db 0 123 261 453 746 1301 2483 3529 1073741824 > gpurun_out/r5_final_regex1g_phases_by_step.json
  rm -rf gpurun_out/r5_prof
  ;;
3)
  timeout -k 5 900 python bench.py --bytes 3900000000 --steps 2 --warmup 1 --cpu-iters 0 --secondary none > gpurun_out/r5_big_3p9gb_bench.json 2> gpurun_out/r5_big.err; echo "big rc=$?"; cut -c1-900 gpurun_out/r5_big_3p9gb_bench.json; tail -2 gpurun_out/r5_big.err | cut -c1-300
  ;;
esac
