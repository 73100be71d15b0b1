set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/r2_t3.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/r2_t3.log
if [ $rc -ge 124 ]; then exit $rc; fi
for cfg in "1000000 200 5" "2000000 500 8" "10000000 1000 10"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_paths.py --snps $1 --inds $2 --pops $3 > gpurun_out/r2_paths_$1.json 2> gpurun_out/r2_paths_$1.err || exit 1
  cat gpurun_out/r2_paths_$1.json
done
timeout -k 10 300 python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo --partitions 3 > gpurun_out/r2_paths_c4_p3b.json 2> gpurun_out/r2_paths_c4_p3b.err || exit 1
cat gpurun_out/r2_paths_c4_p3b.json
