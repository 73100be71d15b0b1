# rocprofv3 counter passes of the class encoder alone (tools/bench_encode.py); run on the GPU box: bash tools/prof_encode.sh [snps inds pops]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-10000000}; N=${2:-1000}; K=${3:-10}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/enc_sq -- python3 $R/tools/bench_encode.py $M $N $K 1 > $R/gpurun_out/enc_sq.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/enc_sq2 -- python3 $R/tools/bench_encode.py $M $N $K 1 > $R/gpurun_out/enc_sq2.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/enc_fetch -- python3 $R/tools/bench_encode.py $M $N $K 1 > $R/gpurun_out/enc_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/enc_write -- python3 $R/tools/bench_encode.py $M $N $K 1 > $R/gpurun_out/enc_write.log 2>&1 || exit 1
echo profiled
