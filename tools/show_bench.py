#!/usr/bin/env python3
"""A readable digest of a bench.py JSON line: python tools/show_bench.py gpurun_out/bench.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline: %.4g %s, %.3f ms/step, roofline frac %.4f (%s), n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["n_gpus"]))
c = d["extra"].get("coded")
if c:
    for k in ("fit_cold", "fit_warm", "fit_direct"):
        print("  coded.%s: %s" % (k, {a: b for a, b in c[k].items() if a != "iterations"}))
    for k in ("amortised_ms_per_iteration", "steady_state_sweep", "pop_like_cold", "class_codes", "identical_frequencies"):
        if k in c:
            print("  coded.%s: %s" % (k, c[k]))
a = d["extra"].get("assign")
if a:
    print("  assign: %.4g SNPs/s, kernel %.2f ms; coded %s; fast %s" % (a["value"], a["kernel_ms"], a.get("coded"), a.get("fast_mode", {}).get("kernel_ms")))
p = d["extra"].get("paths") or {}
keep = ("seconds", "codes_built", "identical", "kernel_ms_warm", "em_seconds", "class_codes_cold", "warm_sweep_kernels_ms", "float32_sweep_kernels_ms", "cold_sweep_kernels_ms")
for k, v in p.items():
    if isinstance(v, dict):
        print("==", k)
        for kk, vv in v.items():
            if isinstance(vv, dict):
                print("   ", kk, {x: y for x, y in vv.items() if x.startswith(keep)})
            elif kk.startswith(keep) or kk in ("classes_per_snp_mean_max", "hash_slots_per_snp", "snps_per_scoring_table", "uncoded_snp_share", "em_table_rows", "em_direct_tile_share", "class_codes"):
                print("   ", kk, vv)
if d.get("cpu_baseline"):
    print("cpu:", d["cpu_baseline"]["value"], "cores", d["cpu_baseline"]["cores"], "| paths total s:", p.get("seconds_total"))
