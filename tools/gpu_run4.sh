set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_bench_contract.py tests/test_gpu_c_abi.py tests/test_gpu_cli.py tests/test_gpu_fisher.py tests/test_gpu_log.py tests/test_gpu_rmse.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider > gpurun_out/r2_t4.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/r2_t4.log
if [ $rc -ge 124 ]; then exit $rc; fi
for w in em af score all; do python -X faulthandler tools/dbg/dbg_exit.py $w > gpurun_out/r2_exit_$w.log 2>&1; echo "exit $w rc=$?"; tail -12 gpurun_out/r2_exit_$w.log; done
bash tools/gpu_prof.sh r2b $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu || exit 1
bash tools/gpu_prof.sh r2c $GRAFT_REPO_ROOT/tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 3 || exit 1
