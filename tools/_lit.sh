cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 600 python bench.py --steps 2 --warmup 1 --cpu-iters 0 --secondary literal_path > gpurun_out/r6_aq_bench_literal.json 2> gpurun_out/r6_aq_bench_literal.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6_aq_bench_literal.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['workload'][:60])
print(json.dumps(d['secondary'],indent=0)[:3000])
P
tail -3 gpurun_out/r6_aq_bench_literal.err | cut -c1-300
