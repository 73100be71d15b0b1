# usage: bash tools/gpu_prof.sh <prefix> <program args...>   (rocprofv3 passes of one command; run on the GPU box)
set -o pipefail
P=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
# which code is being profiled: the loaded library's ids (tools/summarize_profile.py copies them into pmc_summary.json)
python3 -c "import sys, json; sys.path.insert(0, '$R'); from wgsassign_amd import _lib; l = _lib.load(); print(json.dumps({'build_id': l.wgs_build_id().decode(), 'kernels_id': l.wgs_kernels_id().decode()}))" > $R/gpurun_out/${P}_ids.json || exit 1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_kt -- python3 "$@" > $R/gpurun_out/${P}_kt.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${P}_sq -- python3 "$@" > $R/gpurun_out/${P}_sq.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_fetch -- python3 "$@" > $R/gpurun_out/${P}_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_write -- python3 "$@" > $R/gpurun_out/${P}_write.log 2>&1 || exit 1
echo "profiled $P"
