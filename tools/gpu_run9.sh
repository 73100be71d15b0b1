set -o pipefail
mkdir -p gpurun_out
for f in "" "--bgzf"; do
timeout -k 10 300 python tools/bench_reader.py --inds 2000 --sites 4000 --device $f > gpurun_out/r2_reader9$f.json 2> gpurun_out/r2_reader9$f.err; echo "reader $f rc=$?"; cat gpurun_out/r2_reader9$f.json; tail -3 gpurun_out/r2_reader9$f.err
done
timeout -k 10 300 python tools/bench_reader.py --inds 100 --sites 100000 --device > gpurun_out/r2_reader9_n100.json 2>&1; cat gpurun_out/r2_reader9_n100.json
timeout -k 10 300 python tools/bench_reader.py --inds 100 --sites 100000 --device --bgzf > gpurun_out/r2_reader9_n100b.json 2>&1; cat gpurun_out/r2_reader9_n100b.json
