set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_assign_k.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider > gpurun_out/r2_t1.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/r2_t1.log
if [ $rc -ge 124 ]; then exit $rc; fi
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r2a_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/r2a_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/r2a_sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/r2a_sq2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2a_kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > $R/gpurun_out/r2a_kt.log 2>&1 || exit 1
echo done
