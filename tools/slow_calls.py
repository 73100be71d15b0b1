#!/usr/bin/env python3
"""HIP API calls of a `rocprofv3 --hip-trace` run that took longer than a threshold: python tools/slow_calls.py <dir> [ms]"""
import csv
import glob
import sys

d, thr = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows:
        dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        if dt >= thr:
            print("%10.1f ms  +%9.1f ms  tid %s  %s" % (dt, (int(r["Start_Timestamp"]) - t0) * 1e-6, r.get("Thread_Id"), r["Function"]))
