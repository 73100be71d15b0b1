#!/usr/bin/env python3
"""Native Beagle reader rates on one generated file (no GPU needed unless --device):
   python tools/bench_reader.py --inds 2000 --sites 3000 [--device]
Reports the single inflate/index pass (MB/s of text), the threaded parse (sites/s, MB/s) from the start and
from the middle through the index, and with --device the streamed upload into population slabs."""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import reader_cy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inds", type=int, default=2000)
    ap.add_argument("--sites", type=int, default=3000)
    ap.add_argument("--device", action="store_true")
    ap.add_argument("--bgzf", action="store_true", help="write the file as BGZF (what ANGSD produces) instead of plain gzip")
    a = ap.parse_args()
    n, m = a.inds, a.sites
    rng = np.random.default_rng(1)
    d = tempfile.mkdtemp()
    os.environ["WGSASSIGN_INDEX_DIR"] = d
    path = os.path.join(d, "bench.beagle.gz")
    head = "marker\tallele1\tallele2\t" + "\t".join("I%d\tI%d\tI%d" % (i, i, i) for i in range(n))
    text_bytes = 0
    import struct
    import zlib

    class Bgzf:
        """minimal BGZF writer: 60 kB blocks, 'BC' size subfield, end marker"""
        def __init__(self, path):
            self.fh, self.buf = open(path, "wb"), b""

        def block(self, chunk):
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            payload = co.compress(chunk) + co.flush()
            self.fh.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" +
                          struct.pack("<HH", 2, 12 + 6 + len(payload) + 8 - 1) + payload +
                          struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))

        def write(self, text):
            self.buf += text.encode()
            while len(self.buf) >= 60000:
                self.block(self.buf[:60000])
                self.buf = self.buf[60000:]

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            if self.buf:
                self.block(self.buf)
            self.block(b"")
            self.fh.close()

    with (Bgzf(path) if a.bgzf else gzip.open(path, "wt", compresslevel=6)) as fh:
        fh.write(head + "\n")
        for s in range(m):
            g = rng.dirichlet((0.6, 0.6, 0.6), size=n)
            line = "chr1_%d\t0\t1\t" % (s + 1) + "\t".join("%.6f\t%.6f\t%.6f" % tuple(x) for x in g) + "\n"
            text_bytes += len(line)
            fh.write(line)
    res = {"individuals": n, "sites": m, "text_MB": round(text_bytes / 1e6, 1), "gz_MB": round(os.path.getsize(path) / 1e6, 1),
           "threads": min(len(os.sched_getaffinity(0)), 16), "format": "bgzf" if a.bgzf else "gzip"}
    t0 = time.perf_counter()
    idx, _, sites = reader_cy.ensure_index(path)
    t = time.perf_counter() - t0
    res["index_pass"] = {"seconds": round(t, 3), "text_MB_per_s": round(text_bytes / 1e6 / t, 1), "sites_per_s": round(m / t)}
    for label, first in (("parse_from_start", 0), ("parse_second_half_via_index", m // 2)):
        t0 = time.perf_counter()
        with reader_cy.BeagleStream(path, index=idx, first_row=first) as st:
            rows = sum(r.shape[0] for r, _ in st.chunks())
        t = time.perf_counter() - t0
        res[label] = {"seconds": round(t, 3), "sites": rows, "sites_per_s": round(rows / t),
                      "text_MB_per_s": round(text_bytes * rows / m / 1e6 / t, 1)}
    if a.device:
        from wgsassign_amd import device
        ctx = device.get_context()
        t0 = time.perf_counter()
        b, _, _, _ = reader_cy.stream_to_device(path, np.arange(n, dtype=np.int32) % 20, 20, ctx=ctx)
        ctx.sync()
        t = time.perf_counter() - t0
        res["stream_to_device"] = {"seconds": round(t, 3), "sites_per_s": round(m / t)}
        b.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
