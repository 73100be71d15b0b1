#!/usr/bin/env python3
"""VGPRs / spills / occupancy of every kernel of a HIP source, as the compiler reports them for gfx950:
   python tools/kernel_resources.py wgsassign_amd/csrc/assign_kernels.hip [filter]"""
import re
import subprocess
import sys


def resources(path):
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c",
                        "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", path], capture_output=True, text=True)
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = out.setdefault(subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip(), {})
        elif cur is not None:
            cur[m.group(1)] = m.group(2)
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, v in resources(sys.argv[1]).items():
        if flt in name:
            short = re.sub(r"\(anonymous namespace\)::|\(.*\)$|void ", "", name)
            print("%-52s VGPR %3s AGPR %3s spill %3s occ %s LDS %s" % (short, v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill"),
                                                                        v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]")))
