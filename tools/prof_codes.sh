# rocprofv3 passes over tools/check_codes.py (the class-coded kernels at 10M x 1000 x K=10): kernel trace, the issue counters, and the
# LDS counters that say what bounds score_coded_kernel (bank conflicts, LDS-array cycles, address / data FIFO stalls).
#   gpurun -- 'bash tools/prof_codes.sh [prefix]'      then: python tools/summarize_profile.py <prefix> <profiles/name> "<config>"
set -o pipefail
P=${1:-codes}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
python3 -c "import sys, json; sys.path.insert(0, '$R'); from wgsassign_amd import _lib; l = _lib.load(); print(json.dumps({'build_id': l.wgs_build_id().decode(), 'kernels_id': l.wgs_kernels_id().decode()}))" > $R/gpurun_out/${P}_ids.json || exit 1
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_WAIT_[A-Z_]*" | sort -u > $R/gpurun_out/${P}_counters_avail.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_kt -- python3 $R/tools/check_codes.py 10000000 1000 10 > $R/gpurun_out/${P}_kt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${P}_sq -- python3 $R/tools/check_codes.py 10000000 1000 10 > $R/gpurun_out/${P}_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/${P}_sq2 -- python3 $R/tools/check_codes.py 10000000 1000 10 > $R/gpurun_out/${P}_sq2.log 2>&1 || { tail -5 $R/gpurun_out/${P}_sq2.log; echo "(LDS counter pass failed: see ${P}_counters_avail.txt)"; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_fetch -- python3 $R/tools/check_codes.py 10000000 1000 10 > $R/gpurun_out/${P}_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_write -- python3 $R/tools/check_codes.py 10000000 1000 10 > $R/gpurun_out/${P}_write.log 2>&1 || exit 1
echo profiled $P
