#!/bin/bash
# Round-6 evidence, one gpurun call per PART.  Every step is bounded by `timeout -k`; outputs under gpurun_out/r6_*;
# nothing reads stdin.
#   1: the GPU suite, the timed atomic / launch / barrier ceilings, the PMC passes (regex1g with its kernel trace, basic1g, encode)
#   2: (after part 1's summaries were copied into profiles/) the bench line, per-step device time, one train under the
#      kernel trace cut into phases, the phase stamps of the one-launch step (option fuse_step=1)
#   3: the sharded loop at world 1, the 3.9 GB input
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
export TAG=r6
# step numbers at merges 0 / 300 / 1000 / 2000 / 4000 / 8000 / 16000 / 24000 of the headline run (from r6_final_regex1g_iter_us.npy:
# equal neighbours = one step; 0 123 261 453 746 1301 2483 3529 before batches shared second tokens)
EDGES="0 107 226 393 657 1168 2258 3266"
case "$1" in
1)
  timeout -k 5 700 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r6_final_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6_final_pytest_gpu.log | cut -c1-200
  [ -x tools/atomic_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/atomic_peak.hip -o tools/atomic_peak
  timeout 120 tools/atomic_peak > gpurun_out/r6_atomic_peak.json; echo "atomic_peak rc=$?"
  [ -x tools/launch_gap ] && timeout 100 tools/launch_gap > gpurun_out/r6_launch_gap.json
  bash tools/gpu_pmc.sh regex1g 2>&1 | tail -6 | cut -c1-400
  SKIP_KT=1 bash tools/gpu_pmc.sh basic1g 2>&1 | tail -4 | cut -c1-400
  SKIP_KT=1 bash tools/gpu_pmc.sh encode 2>&1 | tail -4 | cut -c1-400
  ;;
2)
  timeout -k 5 1100 python bench.py > gpurun_out/r6_final_bench.json 2> gpurun_out/r6_final_bench.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/r6_final_bench.json; tail -2 gpurun_out/r6_final_bench.err | cut -c1-300
  ITER_NPY=gpurun_out/r6_final_regex1g_iter_us.npy timeout -k 5 300 python tools/iter_profile.py regex1g > gpurun_out/r6_final_regex1g_iter_profile.json 2> gpurun_out/r6_iter.err; echo "iter rc=$?"
  rm -rf gpurun_out/r6_prof; timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r6_prof -o run -- python tools/train_n.py regex1g 31744 > gpurun_out/r6_prof.log 2>&1; echo "prof rc=$?"
  db=$(ls gpurun_out/r6_prof/*/*.db gpurun_out/r6_prof/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > gpurun_out/r6_final_regex1g_kernel_stats_one_train.csv && python tools/rocpd_phases.py $db $EDGES 1073741824 > gpurun_out/r6_final_regex1g_phases_by_step.json
  rm -rf gpurun_out/r6_prof
  BPE_STEP_STAMPS=gpurun_out/r6_stamps.bin timeout -k 5 300 python tools/train_n.py regex1g 31744 fuse_step=1 > gpurun_out/r6_stamps.log 2>&1; python tools/step_stamps.py gpurun_out/r6_stamps.bin > gpurun_out/r6_step_stamps_fused.json; rm -f gpurun_out/r6_stamps.bin
  OPT_RESET="fuse_step=0" REPS=2 timeout -k 5 400 python tools/ab_opts.py regex1g "" "fuse_step=1" > gpurun_out/r6_ab_fuse_step.jsonl 2> gpurun_out/r6_ab.err; echo "ab rc=$?"
  ;;
2b)
  rm -rf gpurun_out/r6_prof; timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r6_prof -o run -- python tools/train_n.py regex1g 31744 > gpurun_out/r6_prof.log 2>&1; echo "prof rc=$?"
  db=$(ls gpurun_out/r6_prof/*/*.db gpurun_out/r6_prof/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > gpurun_out/r6_final_regex1g_kernel_stats_one_train.csv && python tools/rocpd_phases.py $db $EDGES 1073741824 > gpurun_out/r6_final_regex1g_phases_by_step.json
  rm -rf gpurun_out/r6_prof
  ;;
3)
  BENCH_FORCE_DP=1 timeout -k 5 600 python bench.py --steps 3 --warmup 1 --cpu-iters 0 > gpurun_out/r6_bench_dp1_world1.json 2> gpurun_out/r6_dp1.err; echo "dp1 rc=$?"; cut -c1-600 gpurun_out/r6_bench_dp1_world1.json
  timeout -k 5 900 python bench.py --bytes 3900000000 --steps 2 --warmup 1 --cpu-iters 0 --secondary none > gpurun_out/r6_big_3p9gb_bench.json 2> gpurun_out/r6_big.err; echo "big rc=$?"; cut -c1-900 gpurun_out/r6_big_3p9gb_bench.json; tail -2 gpurun_out/r6_big.err | cut -c1-300
  ;;
esac
