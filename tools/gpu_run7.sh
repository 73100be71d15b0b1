set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
C=$R/wgsassign_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_rcp.hip -o gpurun_out/ubench_rcp && timeout -k 10 120 ./gpurun_out/ubench_rcp > gpurun_out/r2_rcp.log 2>&1; cat gpurun_out/r2_rcp.log
# variant A: one Newton refinement in the EM quotient
/opt/rocm/bin/hipcc $FLAGS -DWGS_DIV_NR1 -c $C/em_kernels.hip -o /tmp/em_nr1.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libexpA.so $C/api.o /tmp/em_nr1.o $C/assign_kernels.o $C/beagle_kernels.o $C/rccl_comm.o $C/reader.o -lz -lpthread -ldl || exit 1
# variant B: K = 10 in one register pass
/opt/rocm/bin/hipcc $FLAGS -DWGS_EXPERIMENT_KB10 -c $C/assign_kernels.hip -o /tmp/ak_kb10.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libexpB.so $C/api.o $C/em_kernels.o /tmp/ak_kb10.o $C/beagle_kernels.o $C/rccl_comm.o $C/reader.o -lz -lpthread -ldl || exit 1
for v in base A B; do
  if [ $v = base ]; then unset WGSASSIGN_LIB_PATH; else export WGSASSIGN_LIB_PATH=/tmp/libexp$v.so; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r2_exp_$v.json 2> gpurun_out/r2_exp_$v.err || { echo "bench $v failed"; tail -5 gpurun_out/r2_exp_$v.err; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_exp_$v.json"))
print("$v", "step_ms", round(d["ms_per_step"],3), "em_kernel_ms", d["roofline"]["kernel_ms_avg"], "frac", d["roofline"]["frac"], "assign_ms", d["extra"]["assign"]["kernel_ms"], "checksum", d["extra"]["assign"]["checksum"], "ssq", d["extra"]["ssq_last"][0])
PY
done
export WGSASSIGN_LIB_PATH=/tmp/libexpA.so
timeout -k 10 600 python -m pytest tests/test_gpu_log.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider > gpurun_out/r2_t7_nr1.log 2>&1; echo "NR1 parity rc=$?"; tail -5 gpurun_out/r2_t7_nr1.log
timeout -k 10 300 python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 --loo > gpurun_out/r2_exp_A_loo.json 2>&1; cat gpurun_out/r2_exp_A_loo.json | cut -c1-900
unset WGSASSIGN_LIB_PATH
python __graft_entry__.py --smoke 2>&1 | tail -3
