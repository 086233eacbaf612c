cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; exec </dev/null
timeout 400 python -m pytest tests -m gpu -x -q -k "encode_batch_resident or dp_chain or dp_dense or dp_native or encode_batch_vs" > gpurun_out/r4_z_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r4_z_pytest.log
timeout 500 python bench.py --workload encode --steps 2 --warmup 1 > gpurun_out/r4_z_encode.json 2> gpurun_out/r4_z_encode.err; echo "encode rc=$?"; python - <<'P'
import json
b=json.loads(open("gpurun_out/r4_z_encode.json").readline())
e=b["encode"]
print({k:e[k] for k in ("docs_per_s_device","device_ms_per_step","ms_per_step","device_resident_batch","parity")})
print(e.get("cl100k_sized",{}).get("device_resident_batch"), e.get("cl100k_sized",{}).get("device_ms_per_step"))
P
tail -3 gpurun_out/r4_z_encode.err
