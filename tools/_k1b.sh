cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
REPS=5 timeout -k 5 500 python tools/k1_experiment.py regex1g "64,1024" "7,8,9" 2>&1 | cut -c1-230 | tail -8
