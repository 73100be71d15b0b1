#!/usr/bin/env python3
"""What the box lets this process use: CPUs by affinity, by cgroup quota, and what OpenMP makes of a few thread counts."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def read(p):
    try:
        return open(p).read().strip()
    except OSError:
        return None


res = {"os.cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
       "cgroup_v2_cpu.max": read("/sys/fs/cgroup/cpu.max"), "cgroup_v1_quota": read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
       "cgroup_v1_period": read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"), "proc_self_cgroup": read("/proc/self/cgroup"),
       "loadavg": read("/proc/loadavg")}
from wgsassign_amd import comm
res["usable_cpus"] = comm.usable_cpus()
from oracle import oracle as orc
orc.build()
L = (np.random.default_rng(1).random((100_000, 200)) * 0.5).astype(np.float32)
for t in (8, 16, 32, 64, 128, 256):
    if t > res["affinity"]:
        break
    f = np.full(L.shape[0], 0.25, dtype=np.float32)
    orc.emMAF_update(L, f, t)
    t0 = time.perf_counter()
    for _ in range(5):
        orc.emMAF_update(L, f, t)
    res["emMAF_update_%d_threads_ms" % t] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
print(json.dumps(res, indent=1))
