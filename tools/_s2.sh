cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tie_heavy or chain_step_options or train_golden or full12 or weighted or dist or shard" 2>&1 | tail -6
REPS=2 timeout -k 5 400 python tools/ab_opts.py regex1g "" > gpurun_out/r6_ar_s2.jsonl 2> gpurun_out/r6_ar_s2.err; echo "ab rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r6_ar_s2.jsonl'):
    d=json.loads(l); print({k:d[k] for k in d if k in ('options','best_s','merges_per_s','parity','stats','phase_ms')})
P
tail -2 gpurun_out/r6_ar_s2.err | cut -c1-300
