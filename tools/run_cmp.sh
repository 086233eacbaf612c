cd $GRAFT_REPO_ROOT; exec </dev/null
echo "== MJ4 lookback"; timeout -k 5 120 python bench.py --steps 2 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['device_ms_per_step'])"
echo "== MJ4 three-pass"; BPE_MERGE=0 timeout -k 5 120 python bench.py --steps 2 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['device_ms_per_step'])"
export MINBPE_AMD_LIB=$PWD/minbpe_amd/lib/libbpe_hip_mj8.so
echo "== MJ8 lookback"; timeout -k 5 120 python bench.py --steps 2 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['device_ms_per_step'])"
echo "== MJ8 three-pass"; BPE_MERGE=0 timeout -k 5 120 python bench.py --steps 2 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['device_ms_per_step'])"
