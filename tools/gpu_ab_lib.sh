#!/bin/bash
# same-box A/B of two BUILDS of the library: minbpe_amd/lib/libbpe_hip_base.so (kept from an earlier build) against the
# tree's libbpe_hip.so, ab_opts on $WL (default regex1g) under each, alternating; plus a parity subset under the new one.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
TAG=${TAG:-r6_ab_lib}
WL=${WL:-regex1g}
if [ -z "$SKIP_TESTS" ]; then
(timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py -m gpu -x -q -k "${TESTK:-train_synth_2mb or tie_heavy or (full8r and (1-1 or 1-7)) or long_runs}") > gpurun_out/${TAG}_pytest.log 2>&1; tail -2 gpurun_out/${TAG}_pytest.log
fi
: > gpurun_out/${TAG}.jsonl
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export MINBPE_AMD_LIB=$GRAFT_REPO_ROOT/minbpe_amd/lib/libbpe_hip_base.so; else unset MINBPE_AMD_LIB; fi
  REPS=2 timeout -k 5 400 python tools/ab_opts.py $WL "$OPTS" 2> gpurun_out/${TAG}.err | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); d['lib']='$lib'; print(json.dumps(d))
    print('$lib', d.get('s_per_train'), d.get('merges_per_s'), (d.get('parity') or {}).get('equal'), [p['ms'] for p in d.get('phases',[])], file=sys.stderr)
" >> gpurun_out/${TAG}.jsonl
done
done
tail -2 gpurun_out/${TAG}.err
