#!/usr/bin/env python3
"""Collect rocprofv3 outputs (gpurun_out/<prefix>_{kt,fetch,write,sq}) into profiles/<name>/.

    python tools/summarize_profile.py p2 r01_v2_interleaved "10000000x1000xK10 exact"

Writes kernel_stats.csv (copy of rocprofv3 --stats), pmc_summary.json (per-kernel means of every
counter collected, plus traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch: FETCH_SIZE
is in KiB and on gfx950 reports half of a wide coalesced read stream -- MI355X_MICROARCH.md, HBM).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# substrings of the kernel names whose counters are kept (every kernel of the hot path)
KERNELS = ("em_sweep", "em_coded", "class_encode", "local_encode", "slab_rows", "tokenise", "em_decide", "assign_", "score_", "parts_", "chain_", "ssq_reduce", "rmse", "fisher_", "block_prefix")


def main():
    prefix, name, config = sys.argv[1], sys.argv[2], sys.argv[3]
    out = os.path.join(ROOT, "profiles", name)
    os.makedirs(out, exist_ok=True)
    def newest(pattern):
        """gpurun merges every call's files into the same directories: only the most recent run counts"""
        found = sorted(glob.glob(pattern), key=os.path.getmtime)
        return found[-1:]

    kt = newest(os.path.join(ROOT, "gpurun_out", prefix + "_kt", "*", "*_kernel_stats.csv"))
    if kt:
        shutil.copy(kt[0], os.path.join(out, "kernel_stats.csv"))
        # effective clock per kernel needs its duration: keep the mean duration beside the counters
        durations = {r["Name"].strip('"'): float(r["AverageNs"]) for r in csv.DictReader(open(kt[0]))}
    else:
        durations = {}
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("fetch", "write", "sq", "sq2"):
        for f in newest(os.path.join(ROOT, "gpurun_out", "%s_%s" % (prefix, sub), "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if any(t in k for t in KERNELS):
                    counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary = {"config": config, "kernels": {}}
    # what was profiled: the ids the library on the GPU box reported (tools/gpu_prof.sh), and this tree's commit
    ids = os.path.join(ROOT, "gpurun_out", prefix + "_ids.json")
    if os.path.exists(ids):
        summary.update(json.load(open(ids)))
    import subprocess
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "wgsassign_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
    summary["git_head"] = head + ("+uncommitted changes under csrc/" if dirty else "")
    if len(sys.argv) > 4:   # m_per_gpu n K mode, so bench.py can match its workload to this measurement
        summary["bench_config"] = {"snps_per_gpu": int(sys.argv[4]), "n": int(sys.argv[5]), "K": int(sys.argv[6]), "mode": sys.argv[7]}
    for k, cs in counters.items():
        e = {c: {"dispatches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()}
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            e["traffic_bytes_per_launch"] = (2 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024
        if "SQ_ACTIVE_INST_VALU" in cs and "GRBM_GUI_ACTIVE" in cs:
            # SQ_ACTIVE_INST_VALU counts quad-cycles over all SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' cycles
            # (MI355X_MICROARCH.md: PMC units); 256 CUs x 4 SIMDs
            cycles = e["GRBM_GUI_ACTIVE"]["mean"] / 8.0
            e["valu_busy_frac"] = e["SQ_ACTIVE_INST_VALU"]["mean"] * 4.0 / (1024.0 * cycles)
            e["valu_cycles_per_instruction"] = e["SQ_ACTIVE_INST_VALU"]["mean"] * 4.0 / e["SQ_INSTS_VALU"]["mean"] if "SQ_INSTS_VALU" in cs else None
        if k in durations:
            e["avg_duration_ms_kernel_trace"] = durations[k] / 1e6
            if "GRBM_GUI_ACTIVE" in cs:
                e["effective_clock_ghz"] = e["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / durations[k]
        summary["kernels"][k] = e
    json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
