set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/check_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/check_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/gpu_prof.sh r2d $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu || exit 1
timeout -k 10 300 python tools/bench_paths.py --snps 6250000 --inds 2000 --pops 20 > gpurun_out/r2_paths13_c5.json 2> gpurun_out/r2_paths13.err; cut -c1-700 gpurun_out/r2_paths13_c5.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; cut -c1-3000 gpurun_out/check_bench.json
