#!/usr/bin/env python3
"""gpurun_out/{fcal_*, emc_fused_*, emc_single_*} (tools/prof_em_coded.sh) -> profiles/<name>/summary.json: em_coded_kernel alone, every
launch the same work, with its HBM traffic from the counters corrected by the factors MEASURED for its access widths (4, 8 and 16
bytes per lane: tools/ubench_fetch.hip reads a known 8 GiB).
   python tools/summarize_em_coded_prof.py r05_em_coded"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    return sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)[-1]


def counters(dirname, match):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(dirname + "/*/*_counter_collection.csv"))):
        if match in r["Kernel_Name"]:
            out[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in out.items()}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "r05_em_coded"
    out_dir = os.path.join(ROOT, "profiles", name)
    os.makedirs(out_dir, exist_ok=True)
    cal = {}
    known = 8 << 30
    for (k, c), (n, v) in counters("fcal_fetch", "stream_read").items():
        width = 4 * int(k.split("<")[1].split(">")[0])
        cal["%d_bytes_per_lane" % width] = {"FETCH_SIZE_KiB": v, "bytes_read": known, "factor_bytes_per_reported_byte": round(known / (v * 1024.0), 4)}
    wcal = [v for (_, c), (n, v) in counters("fcal_write", "stream_read").items()]
    res = {"calibration": {"reads": cal, "WRITE_SIZE_KiB_for_8_MiB_written": wcal[0] if wcal else None,
                           "note": "FETCH_SIZE reports half of the bytes of a coalesced streaming read at 4, 8 and 16 bytes per lane alike; WRITE_SIZE is exact"},
           "kernel": "em_coded_kernel<4, 2, 24>", "config": "10000000 x 1000, K=10, 12 EM iterations per fit, class codes present"}
    factor = sum(c["factor_bytes_per_reported_byte"] for c in cal.values()) / max(1, len(cal))
    for v in ("fused", "single"):
        r = json.load(open(os.path.join(ROOT, "gpurun_out", "emc_%s_result.json" % v)))
        leg = r["two_iterations_per_launch" if v == "fused" else "one_iteration_per_launch"]
        f = [x for (k, c), x in counters("emc_%s_fetch" % v, "em_coded_kernel").items() if c == "FETCH_SIZE"][0]
        w = [x for (k, c), x in counters("emc_%s_write" % v, "em_coded_kernel").items() if c == "WRITE_SIZE"][0]
        kt = None
        for row in csv.DictReader(open(newest("emc_%s_kt/*/*_kernel_stats.csv" % v))):
            if "em_coded_kernel" in row["Name"]:
                kt = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6}
        traffic = f[1] * 1024.0 * factor + w[1] * 1024.0
        sq = {c: round(x[1]) for (k, c), x in counters("emc_%s_sq" % v, "em_coded_kernel").items()}
        res[v] = {"launches_profiled": f[0], "kernel_trace": kt, "kernel_ms_per_launch_hip_events": leg["kernel_ms_per_launch"],
                  "FETCH_SIZE_KiB": f[1], "WRITE_SIZE_KiB": w[1], "traffic_bytes_per_launch": traffic,
                  "algorithmic_bytes_per_launch": leg["algorithmic_bytes_per_launch"],
                  "traffic_over_algorithmic": round(traffic / leg["algorithmic_bytes_per_launch"], 3),
                  "hbm_frac_of_its_own_bytes": leg["hbm_frac_of_its_own_bytes"], "sq": sq}
        if "SQ_ACTIVE_INST_VALU" in sq and "GRBM_GUI_ACTIVE" in sq:
            res[v]["valu_busy_frac"] = round(sq["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * sq["GRBM_GUI_ACTIVE"] / 8.0), 4)
    ids = os.path.join(ROOT, "gpurun_out", "emc_ids.json")
    if os.path.exists(ids):
        res.update(json.load(open(ids)))
    res["why_traffic_exceeds_the_algorithmic_bytes"] = ("the algorithmic figure counts 8 bytes per class PRESENT per (slab, SNP) (the sample's mean); the kernel "
                                                        "requests a tile's dictionary rows as far as the tile's RICHEST SNP has classes (tile_rows), ~1.7 x the mean")
    json.dump(res, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
    print(json.dumps(res, indent=1)[:2500])


if __name__ == "__main__":
    main()
