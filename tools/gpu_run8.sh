set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r2_t8.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r2_t8.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 3 > gpurun_out/r2_paths8_p3.json 2> gpurun_out/r2_paths8.err; cat gpurun_out/r2_paths8_p3.json
timeout -k 10 300 python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 1 --exact-parts > gpurun_out/r2_paths8_p1.json 2>> gpurun_out/r2_paths8.err; cat gpurun_out/r2_paths8_p1.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r2_bench8.json 2> gpurun_out/r2_bench8.err; cut -c1-1200 gpurun_out/r2_bench8.json
