# em_coded_kernel alone (tools/check_em_coded.py: every launch the same work) + the FETCH_SIZE / WRITE_SIZE calibration for its access
# widths (tools/ubench_fetch.hip: 4, 8 and 16 bytes per lane).   gpurun -- 'bash tools/prof_em_coded.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 $R/tools/ubench_fetch.hip -o $R/gpurun_out/ubench_fetch || exit 1
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fcal_fetch -- $R/gpurun_out/ubench_fetch > $R/gpurun_out/fcal_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/fcal_write -- $R/gpurun_out/ubench_fetch > $R/gpurun_out/fcal_write.log 2>&1 || exit 1
for V in fused single; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/emc_${V}_kt -- python3 $R/tools/check_em_coded.py 10000000 1000 10 12 $V > $R/gpurun_out/emc_${V}_kt.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/emc_${V}_fetch -- python3 $R/tools/check_em_coded.py 10000000 1000 10 12 $V > $R/gpurun_out/emc_${V}_fetch.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/emc_${V}_write -- python3 $R/tools/check_em_coded.py 10000000 1000 10 12 $V > $R/gpurun_out/emc_${V}_write.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/emc_${V}_sq -- python3 $R/tools/check_em_coded.py 10000000 1000 10 12 $V > $R/gpurun_out/emc_${V}_sq.log 2>&1 || exit 1
  grep "^{" $R/gpurun_out/emc_${V}_kt.log | tail -1 > $R/gpurun_out/emc_${V}_result.json
done
python3 -c "import sys, json; sys.path.insert(0, '$R'); from wgsassign_amd import _lib; l = _lib.load(); print(json.dumps({'build_id': l.wgs_build_id().decode(), 'kernels_id': l.wgs_kernels_id().decode()}))" > $R/gpurun_out/emc_ids.json
echo profiled em_coded
