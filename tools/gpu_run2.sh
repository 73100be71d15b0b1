set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/r2_t2.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/r2_t2.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 3 > gpurun_out/r2_paths_c4_p3.json 2> gpurun_out/r2_paths_c4_p3.err || exit 1
timeout -k 10 300 python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 1 --exact-parts > gpurun_out/r2_paths_c4_p1.json 2> gpurun_out/r2_paths_c4_p1.err || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err || exit 1
cat gpurun_out/r2_paths_c4_p3.json gpurun_out/r2_paths_c4_p1.json gpurun_out/r2_bench2.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o gpurun_out/ubench_f64 && timeout -k 10 120 ./gpurun_out/ubench_f64 > gpurun_out/r2_ubench.log 2>&1; cat gpurun_out/r2_ubench.log
