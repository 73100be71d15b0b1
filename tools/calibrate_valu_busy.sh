# Calibration of the "VALU busy" figure the profiles quote (SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)):
# the same counters for kernels that do nothing but issue one kind of VALU instruction back to back (tools/ubench_f64.hip,
# 4 waves per SIMD).  Run on the GPU box:  bash tools/calibrate_valu_busy.sh   ->  gpurun_out/calib/ + calib_valu.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 $R/tools/ubench_f64.hip -o $R/gpurun_out/ubench_f64 || exit 1
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/calib -- $R/gpurun_out/ubench_f64 > $R/gpurun_out/calib.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, json, collections
f = sorted(glob.glob("$R/gpurun_out/calib/*/*_counter_collection.csv"))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["v_mul_f64", "v_add_f64", "v_fma_f64", "v_rcp_f64", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_div_fixup_f64", "v_div_fmas_f64",
         "v_div_scale_f64", "v_log_f32", "v_add_f32", "v_fma_f32", "v_mov_b32", "v_ldexp_f64", "v_frexp_mant_f64", "v_rndne_f64",
         "v_max_f64", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"]
out = {}
for (k, grid), cs in sorted(agg.items()):
    i = int(k.split("<")[1].split(">")[0])
    act, ins, gui = (sum(cs[c]) / len(cs[c]) for c in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"))
    waves_per_simd = grid // 256 // 256
    out.setdefault(names[i], {})["%dw" % waves_per_simd] = {"valu_busy_frac": round(act * 4.0 / (1024.0 * gui / 8.0), 4),
                                                           "active_quadcycles_per_instruction": round(act / ins, 3)}
json.dump(out, open("$R/gpurun_out/calib_valu.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
