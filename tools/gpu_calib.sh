#!/bin/bash
# Counter calibration: tools/pmc_calib (kernels with known byte counts) under the three counter sets, each its own
# rocprofv3 run (kernel trace only).  Summaries land in gpurun_out/${TAG}_pmc_calibration.json.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
TAG=${TAG:-r4}
[ -x tools/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib || exit 1
rm -rf gpurun_out/cal_f gpurun_out/cal_w gpurun_out/cal_r
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/cal_f -o run -- tools/pmc_calib > gpurun_out/${TAG}_calib_requested.json 2> gpurun_out/cal_f.log; echo "fetch rc=$?"
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/cal_w -o run -- tools/pmc_calib > /dev/null 2> gpurun_out/cal_w.log; echo "write rc=$?"
timeout -k 5 200 rocprofv3 --pmc ${RAW_COUNTERS:-TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_WRITE_ATOMIC_32B TCC_EA0_RDREQ_128B} --kernel-trace -d gpurun_out/cal_r -o run -- tools/pmc_calib > /dev/null 2> gpurun_out/cal_r.log; echo "raw rc=$?"
tail -3 gpurun_out/cal_r.log | cut -c1-300
python tools/pmc_calib_summary.py gpurun_out/${TAG}_calib_requested.json gpurun_out/cal_f gpurun_out/cal_w gpurun_out/cal_r > gpurun_out/${TAG}_pmc_calibration.json; echo "summary rc=$?"
cat gpurun_out/${TAG}_pmc_calibration.json | cut -c1-4000
rm -rf gpurun_out/cal_f gpurun_out/cal_w gpurun_out/cal_r
