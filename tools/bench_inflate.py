#!/usr/bin/env python3
"""Host side of the device ingest alone (no GPU): the text hand-over of csrc/reader.cpp -- inflate ahead into buffers, cut at
newlines, list the lines -- drained without parsing, with ordinary (not page-locked) buffers, per thread count.
   python tools/bench_inflate.py --inds 2000 --sites 100000 [--threads 4,8,16,32]"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth  # noqa: E402
from wgsassign_amd import _lib, reader_cy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inds", type=int, default=2000)
    ap.add_argument("--sites", type=int, default=100000)
    ap.add_argument("--threads", default="4,8,16,32")
    a = ap.parse_args()
    d = tempfile.mkdtemp()
    os.environ["WGSASSIGN_INDEX_DIR"] = d
    path = os.path.join(d, "x.beagle.gz")
    synth.make_pool_file(path, a.inds, a.sites, pool=min(1024, a.sites))
    text = a.sites * (27 * a.inds + 14)
    idx, _, _ = reader_cy.ensure_index(path)
    lib = _lib.load()
    res = {"individuals": a.inds, "sites": a.sites, "text_GB": round(text / 1e9, 2)}
    for t in (int(x) for x in a.threads.split(",")):
        best = None
        for _ in range(2):
            with reader_cy.BeagleStream(path, threads=t, index=idx, first_row=0) as st:
                got = ctypes.c_int64()
                t0 = time.perf_counter()
                _lib.check(lib.wgs_debug_reader_text_rows(st._h, 256 << 20, -1, None, 0, ctypes.byref(got)))
                dt = time.perf_counter() - t0
            assert got.value == a.sites
            best = dt if best is None else min(best, dt)
        res["%d_threads" % t] = {"seconds": round(best, 3), "text_GB_per_s": round(text / 1e9 / best, 2), "sites_per_s": round(a.sites / best)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
