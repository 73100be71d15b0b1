#!/usr/bin/env python3
"""Renders the whole-path section of a bench.py line (extra.coded, extra.paths) as a markdown table:
   python tools/paths_md.py profiles/r04_bench_final.json > profiles/r04_whole_paths.md"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ex = d["extra"]
rows = []


def ms(x):
    return "%.1f ms" % (x * 1e3) if x < 0.1 else "%.3f s" % x


c = ex.get("coded")
if c:
    rows.append(("10M x 1000 x K=10 `--get_reference_af` (%d iterations)" % c["fit_cold"]["iterations"][0],
                 ms(c["fit_cold"]["seconds"]),
                 ms(c["fit_warm"]["seconds"]), ms(c["fit_direct"]["seconds"]), "built inside the fit: %s (%.1f ms)%s" % (c["fit_cold"]["codes_built_inside_the_fit"], c["fit_cold"].get("of_which_codes_build_ms", 0),
                                                            ("; the codes' hipMalloc took %.0f ms on the helper thread, the fit did not wait: the second fit builds them in %s" %
                                                             (c["fit_cold"]["codes_memory_arrived_after_the_fit_hipMalloc_ms"], ms(c["fit_second_builds_the_codes"]["seconds"])))
                                                            if "codes_memory_arrived_after_the_fit_hipMalloc_ms" in c["fit_cold"] else "")))
    if "pop_like_cold" in c:
        a = ex["assign"]
        rows.append(("... `--get_pop_like` alone", ms(c["pop_like_cold"]["seconds"]), "%.1f ms (kernel)" % a["coded"]["kernel_ms"], ms(a["seconds"]), "build %.1f ms" % c["pop_like_cold"]["of_which_codes_build_ms"]))
for name, v in (ex.get("paths") or {}).items():
    if not isinstance(v, dict):
        continue
    fit = v if "seconds_cold" in v else v.get("get_reference_af")
    if fit:
        cc = fit.get("class_codes") or {}
        cold = ms(fit["seconds_cold"])
        note = "codes built inside the cold fit: %s%s; iterations %s; identical %s" % (fit["codes_built_inside_the_cold_fit"], (" (%.1f ms)" % cc["build_ms"]) if cc.get("available") else "",
                                                                                         fit["iterations"][0], fit["identical_frequencies"])
        if "codes_memory_arrived_after_the_cold_fit_hipMalloc_ms" in fit:
            note += "; the codes' hipMalloc took %.0f ms (helper thread), the cold fit did not wait; the second fit, building them: %s" % (
                fit["codes_memory_arrived_after_the_cold_fit_hipMalloc_ms"], ms(fit["seconds_second_fit_building_the_codes"]))
        mal = fit.get("of_which_hipMalloc_seconds", {})

        def cell(seconds, key):
            """a figure with the driver's share of it taken out where that share is more than 5 ms (VRAM an earlier process used)"""
            m = mal.get(key, 0.0)
            return ms(seconds) if m < 5e-3 else "%s + %s of hipMalloc" % (ms(seconds - m), ms(m))
        rows.append((name + " fit", cell(fit["seconds_cold"], "cold"), cell(fit["seconds_warm"], "warm"), cell(fit["seconds_float32"], "float32"), note))
    pl = v.get("get_pop_like")
    if pl:
        cc = pl.get("class_codes_cold") or {}
        rows.append(("... `--get_pop_like` alone", ms(pl["seconds_cold"]), "%.1f ms (kernel)" % pl["kernel_ms_warm"], ms(pl["seconds_float32"]),
                     "cold build %s; after the fit %s; identical %s" % (("%.1f ms" % cc["build_ms"]) if cc.get("available") else "none", ms(pl["seconds_after_the_fit"]), pl["identical_sums"])))
    for k in ("loo", "loo_partition_sites_3"):
        if k in v:
            rows.append(("... `--%s`" % k.replace("_partition_sites_3", " --partition_sites 3"), ms(v[k]["seconds"]), "", "", "EM phase %s (%d re-fits, %s)%s" % (ms(v[k]["em_seconds"]), v[k]["re_fits"], v[k].get("em_sweep_kernel", "em_sweep_group_kernel<exact>"),
                                                                            ("; of which hipMalloc of the re-fits' buffers %s" % ms(v[k]["of_which_hipMalloc_seconds"])) if v[k].get("of_which_hipMalloc_seconds", 0) > 2e-3 else "")))
print("# Whole paths on one MI355X (`bench.py`, %s)\n" % d["config"]["workload"])
print("Headline: %.4g %s, %.2f ms per iteration over the float32 matrix, %.4f of the HBM peak (`%s`).\n" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"]))
print("cold = nothing built for the matrix before the call (the class codes, where the cost models want them, are built inside it -- unless the driver takes longer than 3 ms to hand out their memory: then the call runs over the float32 slabs and the NEXT call builds them, see the notes); warm = codes present; float32 = `WGSASSIGN_CODES=0`.\n")
print("| path | cold | warm | float32 slabs | notes |\n|---|---|---|---|---|")
for r in rows:
    print("| " + " | ".join(str(x) for x in r) + " |")
rg = (ex.get("paths") or {}).get("realistic_gl_2Mx1000_K10")
if rg:
    print("\nQuality-dependent likelihoods (%s): classes per SNP mean / max %s, %s hash slots per SNP, %s SNPs per scoring table, uncoded SNPs %.4f %%, EM table rows %s." %
          (rg["generator"], rg["classes_per_snp_mean_max"], rg["hash_slots_per_snp"], rg["snps_per_scoring_table"], 100 * (rg["uncoded_snp_share"] or 0), rg["em_table_rows"]))
    cr = rg.get("class_rich_uniform_Q20_40")
    if cr:
        print("\nClass-rich matrix (qualities uniform over eight values Q20..Q40: %s classes per SNP in the sample, not coded): `--get_pop_like` exact %.1f ms, float32 mode %.1f ms "
              "(max relative deviation of the sums %.1e): the default stays exact and pays %.2fx (%s)." %
              (cr["classes_per_snp_in_the_sample"], cr["get_pop_like_exact"]["kernel_ms"], cr["get_pop_like_float32_mode"]["kernel_ms"],
               cr["get_pop_like_float32_mode"]["max_rel_dev_of_sums_vs_exact"], cr["price_of_exact"], cr["rule"]))


def cm_rows():
    out = []

    def add(name, cm, key):
        if not cm:
            return
        e = cm.get(key, {})
        out.append((name, "yes" if cm["builds_the_codes"] else "no", "%s / %s" % (cm["saving_cold_ms"]["predicted"], cm["saving_cold_ms"]["measured"]),
                    "%s / %s" % (cm["saving_warm_ms"]["predicted"], cm["saving_warm_ms"]["measured"]), "%s / %s" % (e.get("cold"), e.get("warm")),
                    ("had it built: cold %s / %s, warm %s / %s; the no was right: %s" % (cm["had_it_built"]["saving_cold_ms"]["predicted"], cm["had_it_built"]["saving_cold_ms"]["measured"],
                                                                                         cm["had_it_built"]["saving_warm_ms"]["predicted"], cm["had_it_built"]["saving_warm_ms"]["measured"],
                                                                                         cm["had_it_built"]["the_no_was_right"])) if "had_it_built" in cm else ""))
    if c:
        add("10M x 1000 x K=10 fit", c.get("cost_model"), "abs_error_share_of_float32_fit")
    for name, v in (ex.get("paths") or {}).items():
        if not isinstance(v, dict):
            continue
        fit = v if "seconds_cold" in v else v.get("get_reference_af")
        if fit:
            add(name + " fit", fit.get("cost_model"), "abs_error_share_of_float32_fit")
        if v.get("get_pop_like"):
            add("... `--get_pop_like`", v["get_pop_like"].get("cost_model"), "abs_error_share_of_float32_sweep")
    return out


rows = cm_rows()
if rows:
    print("\n## The cost models' predictions beside the measurements (ms saved against the float32 path: predicted / measured)\n")
    print("| path | model builds the codes | saving cold | saving warm | abs. error as share of the float32 time (cold / warm) | |\n|---|---|---|---|---|---|")
    for r in rows:
        print("| " + " | ".join(str(x) for x in r) + " |")
sp = ex.get("shard_projection")
if sp:
    print("\n## Projection of the SNP-sharded path from one GPU (%s)\n" % sp["note"])
    print("one tagged all-reduce on a one-rank RCCL communicator: %s us\n" % sp["collectives"].get("allreduce_us"))
    print("| leg | whole matrix | shard of 2 | shard of 4 | shard of 8 | collectives at 8 | speedup 2 / 4 / 8 | >= 6x at 8 | headroom per collective (us) |\n|---|---|---|---|---|---|---|---|---|")
    legs = [k for k in sp["N=8"] if isinstance(sp["N=8"][k], dict)]
    for k in legs:
        r = [sp["N=%d" % N].get(k) for N in (2, 4, 8)]
        if not all(r):
            continue
        print("| %s | %s | %s | %s | %s | %d all-reduces, %d broadcasts | %s / %s / %s | %s | %s |" % (
            k, ms(sp["seconds_on_the_whole_matrix"][k]), ms(r[0]["shard_seconds"]), ms(r[1]["shard_seconds"]), ms(r[2]["shard_seconds"]),
            r[2]["allreduces"], r[2]["broadcasts"], r[0]["speedup"], r[1]["speedup"], r[2]["speedup"], r[2]["holds_6x"], r[2]["headroom_us_per_collective"]))
