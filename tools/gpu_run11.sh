set -o pipefail
mkdir -p gpurun_out
bash tools/gpu_prof.sh r2d $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu || exit 1
bash tools/gpu_prof.sh r2e $GRAFT_REPO_ROOT/tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 3 || exit 1
timeout -k 10 300 python tools/bench_paths.py --snps 6250000 --inds 2000 --pops 20 > gpurun_out/r2_paths11_c5.json 2> gpurun_out/r2_paths11_c5.err; cat gpurun_out/r2_paths11_c5.json
timeout -k 10 300 python tools/bench_paths.py --snps 600000 --inds 80 --pops 5 --loo > gpurun_out/r2_paths11_600k.json 2>> gpurun_out/r2_paths11_c5.err; cat gpurun_out/r2_paths11_600k.json
timeout -k 10 300 python tools/bench_paths.py --snps 1000000 --inds 200 --pops 5 --loo > gpurun_out/r2_paths11_c2.json 2>> gpurun_out/r2_paths11_c5.err; cat gpurun_out/r2_paths11_c2.json
