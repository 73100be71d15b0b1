#!/usr/bin/env python3
"""Turns the kernel trace of tools/prof_ingest.sh into profiles/<dir>/{kernel_stats.csv, ids.json, summary.md}: per-kernel calls and
times of the device-resident BGZF ingest, and the tokeniser's roofline line (text bytes in + slab bytes out over its time, against the
HBM peak).   python tools/summarize_ingest_prof.py gpurun_out profiles/r04_ingest"""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
ids = json.load(open(os.path.join(src, "ing_ids.json")))
stats = sorted(glob.glob(os.path.join(src, "ing_kt", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]      # (the most recent run only)
assert stats, "no kernel_stats.csv under %s/ing_kt" % src
shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
json.dump(ids, open(os.path.join(dst, "ids.json"), "w"), indent=1)
rows = list(csv.DictReader(open(stats[0])))
plain = None
for line in open(os.path.join(src, "ing_plain.log")):
    if line.lstrip().startswith("{"):
        plain = json.loads(line)
n, m = ids["individuals"], ids["sites"]
PASSES = 2          # bench_reader.py --only-device-inflate streams the file twice (the second run is the warm one)
text = PASSES * (plain.get("text_MB", 0) * 1e6 if plain else 0.0)
slab = PASSES * 8.0 * n * m
out = ["# Device-resident BGZF ingest: kernel trace (`tools/prof_ingest.sh %d %d`)" % (m, n), "",
       "Library ids: build `%s`, kernels `%s`, ingest kernels `%s` (sha256 over ingest.hip, inflate.hip, common.h)." %
       (ids["build_id"], ids["kernels_id"], ids["ingest_kernels_id"]), "",
       "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
tok_ms = None
for r in rows:
    name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    tot = float(r["TotalDurationNs"]) / 1e6
    if "tokenise_kernel" in name:
        tok_ms = tot
    if tot >= 0.05:
        out.append("| `%s` | %s | %.2f | %.3f | %s |" % (name[:70], r["Calls"], tot, float(r["AverageNs"]) / 1e6, r["Percentage"]))
out.append("")
if tok_ms:
    out.append("Tokeniser roofline (both passes of the file): %.2f GB of text in + %.2f GB of slab rows out in %.2f ms = **%.2f TB/s = %.0f %% of the 8 TB/s HBM peak** "
               "(its algorithmic bytes: every text byte read once, every slab byte written once)." %
               (text / 1e9, slab / 1e9, tok_ms, (text + slab) / tok_ms / 1e9, (text + slab) / tok_ms / 1e9 / 8.0 * 100))
if plain:
    out += ["", "Un-profiled run of the same command (`ing_plain.log`):", "", "```json", json.dumps(plain, indent=1), "```"]
open(os.path.join(dst, "summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
