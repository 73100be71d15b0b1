# rocprofv3 counter pass of the leave-one-out re-fits, through the codes and over the float32 slabs (tools/probe_loo.py); on the GPU box:
#   bash tools/prof_loo.sh [snps inds pops]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-2000000}; N=${2:-500}; K=${3:-8}
export TMPDIR=/tmp
export PROBE_LOO_LEGS=coded_warm,float32_again
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/loo_sq -- python3 $R/tools/probe_loo.py $M $N $K > $R/gpurun_out/loo_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/loo_sq2 -- python3 $R/tools/probe_loo.py $M $N $K > $R/gpurun_out/loo_sq2.log 2>&1 || exit 1
python3 $R/tools/sum_counters.py $R/gpurun_out/loo_sq $R/gpurun_out/loo_sq2 > $R/gpurun_out/loo_counters.txt
echo profiled
