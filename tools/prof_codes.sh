set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/codes_kt -- python3 $R/tools/check_codes.py 10000000 1000 10 > $R/gpurun_out/codes_kt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/codes_sq -- python3 $R/tools/check_codes.py 10000000 1000 10 > $R/gpurun_out/codes_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/codes_sq2 -- python3 $R/tools/check_codes.py 10000000 1000 10 > $R/gpurun_out/codes_sq2.log 2>&1 || exit 1
echo profiled
