#!/usr/bin/env python3
"""Per kernel: dispatches, summed duration and summed counters of rocprofv3 --pmc passes: python tools/sum_counters.py <dir> [<dir> ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def newest(pattern):
    """gpurun merges every call's files into the same directories: only the most recent run counts"""
    return sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1:]


def short(name):
    name = re.sub(r"^void |\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)


for d in sys.argv[1:]:
    sums = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    for f in newest(d + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
    dur = defaultdict(float)
    for f in newest(d + "/**/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    print("==", d)
    for k in sorted(sums, key=lambda x: -dur[x])[:8]:
        print("%-44s %5d dispatches %10.2f ms  " % (k, len(seen[k]), dur[k]) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(sums[k].items())))
