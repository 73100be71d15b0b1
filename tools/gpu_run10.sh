set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/r2_t10.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r2_t10.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 3 > gpurun_out/r2_paths10_p3.json 2> gpurun_out/r2_paths10.err; cat gpurun_out/r2_paths10_p3.json
timeout -k 10 300 python tools/bench_paths.py --snps 5000000 --inds 180 --pops 5 --loo > gpurun_out/r2_paths10_5M.json 2>> gpurun_out/r2_paths10.err; cat gpurun_out/r2_paths10_5M.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r2_bench10.json 2> gpurun_out/r2_bench10.err; cut -c1-1500 gpurun_out/r2_bench10.json | tr ',' '\n' | grep -E "ms_per_step|frac|kernel_ms|value" | head -12
