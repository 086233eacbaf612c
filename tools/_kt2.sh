cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -rf gpurun_out/kt_u
timeout -k 5 300 rocprofv3 --kernel-trace -d gpurun_out/kt_u -o run -- python tools/train_n.py regex1g 31744 $KT_OPTS > gpurun_out/kt_u.log 2>&1; echo "kt rc=$?"
db=$(ls gpurun_out/kt_u/*/*.db gpurun_out/kt_u/*.db 2>/dev/null | head -1)
python tools/rocpd_phases.py $db 0 123 261 453 746 1301 2483 3529 1073741824 > gpurun_out/${TAG}_phases_by_step.json
python tools/rocpd_stats.py $db > gpurun_out/${TAG}_kernel_stats_one_train.csv
rm -rf gpurun_out/kt_u
