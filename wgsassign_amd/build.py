"""Builds libwgsassign_hip.so in-tree with hipcc for gfx950 (MI355X).  No GPU needed to build."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwgsassign_hip.so")
SOURCES = ["api.hip", "em_api.hip", "score_api.hip", "codes.hip", "codes_kernels.hip", "em_kernels.hip", "assign_kernels.hip", "beagle_kernels.hip", "ingest.hip", "inflate.hip", "rccl_comm.hip", "reader.cpp"]
# -ffp-contract=off: the exact-mode kernels restate the reference's rounding sequence operation by
# operation; hipcc's default (fast) contraction would fuse a*b+c and change results.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"] + os.environ.get("WGSASSIGN_BUILD_DEFINES", "").split()


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP toolchain (ROCm) is required to build wgsassign_amd")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


KERNEL_SOURCES = ["em_kernels.hip", "assign_kernels.hip", "beagle_kernels.hip", "codes_kernels.hip", "common.h", "log_table.h"]   # what the profiled kernels are made of


INGEST_SOURCES = ["ingest.hip", "inflate.hip", "common.h"]   # the device side of the streamed reader: tokeniser, line listing, BGZF inflate


def source_ids():
    """(build id, kernels id, ingest kernels id): sha256[:16] over every source of the library / over the sources of the EM,
    scoring and encoder kernels / over the sources of the ingest kernels.  Compiled into the library (wgs_build_id, wgs_kernels_id) and written into every profile summary
    (tools/summarize_profile.py), so that bench.py quotes counters only from a profile of the kernels it is timing."""
    import hashlib

    def digest(paths):
        h = hashlib.sha256()
        for p in paths:
            h.update(os.path.basename(p).encode() + b"\0")
            h.update(open(p, "rb").read())
        return h.hexdigest()[:16]
    every = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")) and f != "build_id.h")
    every.append(os.path.join(HERE, "..", "include", "wgsassign_hip.h"))
    return digest(every), digest([os.path.join(CSRC, f) for f in KERNEL_SOURCES]), digest([os.path.join(CSRC, f) for f in INGEST_SOURCES])


def build(force=False, verbose=False):
    cc = hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "em_state.h"), os.path.join(CSRC, "log_table.h"), os.path.join(CSRC, "reader_text.h"),
               os.path.join(HERE, "..", "include", "wgsassign_hip.h")]
    id_header = os.path.join(CSRC, "build_id.h")
    text = '#define WGS_BUILD_ID "%s"\n#define WGS_KERNELS_ID "%s"\n#define WGS_INGEST_KERNELS_ID "%s"\n' % source_ids()
    if not os.path.exists(id_header) or open(id_header).read() != text:
        with open(id_header, "w") as fh:
            fh.write(text)
    # objects built with other flags (WGSASSIGN_BUILD_DEFINES) are stale whatever their time stamps say
    stamp = os.path.join(CSRC, ".build_flags")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(FLAGS):
        force = True
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers + ([id_header] if src == "api.hip" else [])):
            if src.endswith(".cpp"):      # host-only C++ (the streamed reader)
                jobs.append([cc, "-O3", "-std=c++17", "-fPIC", "-Wall", "-c", s, "-o", o])
            else:
                jobs.append([cc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    if force or jobs or _stale(LIB, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-lpthread", "-ldl"])
    with open(stamp, "w") as fh:
        fh.write(" ".join(FLAGS))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
