cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
rm -rf gpurun_out/m0_prof
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/m0_prof -o run -- python tools/train_n.py regex1g 256 mode=0 merge=1 > gpurun_out/m0_prof.log 2>&1; echo "prof rc=$?"
db=$(ls gpurun_out/m0_prof/*/*.db gpurun_out/m0_prof/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db > gpurun_out/r6_mode0_lookback_kernel_stats.csv && head -8 gpurun_out/r6_mode0_lookback_kernel_stats.csv | cut -c1-60,150-260
rm -rf gpurun_out/m0_prof
