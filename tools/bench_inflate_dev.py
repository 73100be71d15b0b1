#!/usr/bin/env python3
"""BGZF inflate on the device (csrc/inflate.hip) on a generated Beagle file: every block against zlib, kernel rate.
   python tools/bench_inflate_dev.py --inds 2000 --sites 20000 [--block 60000]"""
import argparse
import ctypes
import json
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from wgsassign_amd import _lib, device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inds", type=int, default=2000)
    ap.add_argument("--sites", type=int, default=20000)
    ap.add_argument("--reblock", type=int, default=0, help="re-cut the text into BGZF blocks of this many bytes (0: the pool file's own blocks)")
    ap.add_argument("--lowdepth", action="store_true", help="text of a simulated 2x matrix (tests/synth.make_beagle: few distinct likelihoods, "
                                                            "long far matches, 10x compression) instead of the pool file's random digits (3x)")
    ap.add_argument("--benchfile", action="store_true", help="the file of bench.py's ingest leg (tools/beagle_files.write_lowdepth_bgzf: one line "
                                                             "per member, its name in a stored block in front of it)")
    a = ap.parse_args()
    d = tempfile.mkdtemp()
    path = os.path.join(d, "x.beagle.gz")
    if a.benchfile:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import beagle_files
        beagle_files.write_lowdepth_bgzf(path, a.inds, a.sites)
    elif a.lowdepth:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_cli
        L, IDs = synth.make_beagle(a.sites, a.inds, 5, seed=4242)
        bench_cli.write_beagle(path, L, os.path.join(d, "ids.txt"), IDs, "bgzf")
    else:
        synth.make_pool_file(path, a.inds, a.sites, pool=min(1024, a.sites))
    raw = open(path, "rb").read()
    if a.reblock:
        import gzip
        text = gzip.decompress(raw)
        raw = b"".join(synth.bgzf_block(text[i:i + a.reblock], level=6) for i in range(0, len(text), a.reblock)) + synth.bgzf_block(b"")
    in_off, in_len, isize = [], [], []
    off = 0
    while off < len(raw):
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        hdr = 12 + xlen
        in_off.append(off + hdr)
        in_len.append(bsize - hdr - 8)
        isize.append(struct.unpack_from("<I", raw, off + bsize - 4)[0])
        off += bsize
    n = len(in_off)
    in_off = np.array(in_off, dtype=np.uint64)
    in_len = np.array(in_len, dtype=np.uint32)
    isize = np.array(isize, dtype=np.uint32)
    out_off = np.concatenate([[0], np.cumsum(isize[:-1], dtype=np.uint64)]).astype(np.uint64)
    total = int(isize.sum())
    out = np.zeros(total, dtype=np.uint8)
    status = np.zeros(n, dtype=np.uint8)
    comp = np.frombuffer(raw, dtype=np.uint8).copy()
    ctx = device.get_context()
    ms = ctypes.c_float()
    best = None
    for _ in range(3):
        _lib.check(_lib.load().wgs_debug_inflate(ctx.handle, comp.ctypes.data, len(raw), in_off.ctypes.data, in_len.ctypes.data,
                                                 out_off.ctypes.data, isize.ctypes.data, n, out.ctypes.data, total, status.ctypes.data, ctypes.byref(ms)))
        best = ms.value if best is None else min(best, ms.value)
    t0 = time.perf_counter()
    ref = b"".join(zlib.decompress(raw[int(o):int(o) + int(l)], -15) for o, l in zip(in_off, in_len))
    t_zlib = time.perf_counter() - t0
    print(json.dumps({"blocks": n, "text_GB": round(total / 1e9, 3), "compressed_GB": round(len(raw) / 1e9, 3), "kernel_ms": round(best, 2),
                      "device_text_GB_per_s": round(total / 1e9 / (best * 1e-3), 2), "rejected_blocks": int(status.sum()),
                      "identical_to_zlib": bool(out.tobytes() == ref), "zlib_one_thread_GB_per_s": round(total / 1e9 / t_zlib, 2)}))


if __name__ == "__main__":
    main()
