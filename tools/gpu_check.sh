# The GPU-box recipe of this repository (run through gpurun): all GPU tests, the bench line, the whole-path timings.
#   gpurun --timeout 1200 -- 'bash tools/gpu_check.sh'
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/check_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/check_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err || exit 1
cat gpurun_out/check_bench.json
timeout -k 10 300 python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 3 > gpurun_out/check_paths_c4.json 2> gpurun_out/check_paths.err || exit 1
cat gpurun_out/check_paths_c4.json
python __graft_entry__.py --smoke 2>&1 | tail -2
