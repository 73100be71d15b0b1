set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/check_fast_mode.py > gpurun_out/r2_fast.json 2> gpurun_out/r2_fast.err; echo "fast rc=$?"; cat gpurun_out/r2_fast.json; tail -5 gpurun_out/r2_fast.err
timeout -k 10 300 python tools/bench_reader.py --inds 2000 --sites 3000 --device > gpurun_out/r2_reader.json 2> gpurun_out/r2_reader.err; echo "reader rc=$?"; cat gpurun_out/r2_reader.json; tail -3 gpurun_out/r2_reader.err
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_multirank.py -q -m gpu -p no:cacheprovider > gpurun_out/r2_t5.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_t5.log
