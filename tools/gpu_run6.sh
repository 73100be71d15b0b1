set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r2_t6.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r2_t6.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench6.json 2> gpurun_out/r2_bench6.err; echo "bench rc=$?"; cat gpurun_out/r2_bench6.json; tail -3 gpurun_out/r2_bench6.err
WGS_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --snps 1250000 --no-cpu --no-assign > gpurun_out/r2_bench6_shard.json 2> gpurun_out/r2_bench6_shard.err; echo "bench shard rc=$?"; cat gpurun_out/r2_bench6_shard.json; tail -3 gpurun_out/r2_bench6_shard.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --snps 1250000 --no-cpu --no-assign > gpurun_out/r2_bench6_shard_local.json 2>&1; cat gpurun_out/r2_bench6_shard_local.json
