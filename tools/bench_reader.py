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
    ap.add_argument("--only-device-inflate", action="store_true", help="with --device: only the device-resident path (inflate, listing and tokeniser "
                                                                     "on the device), twice -- what tools/prof_ingest.sh profiles")
    ap.add_argument("--bgzf", action="store_true", help="write the file as BGZF (what ANGSD produces) instead of plain gzip")
    ap.add_argument("--lowdepth", action="store_true", help="BGZF text of a simulated 2x matrix, whole lines per member (tools/beagle_files.py: few "
                                                            "distinct likelihoods per site, deflates 10-20x like the reference's bundled 2x files) "
                                                            "instead of the pool file (random digits, 3x, two members per line)")
    a = ap.parse_args()
    n, m = a.inds, a.sites
    d = tempfile.mkdtemp()
    os.environ["WGSASSIGN_INDEX_DIR"] = d
    path = os.path.join(d, "bench.beagle.gz")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import synth
    # lines come from a pool of pre-made GL sections (a third of the genotypes missing, like low-depth ANGSD output):
    # formatting m x 3n numbers in Python would take longer than everything measured here
    if a.lowdepth:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import beagle_files
        a.bgzf = True
        text_bytes, _, _ = beagle_files.write_lowdepth_bgzf(path, n, m)
    elif a.bgzf:
        synth.make_pool_file(path, n, m, pool=min(1024, m))
        text_bytes = m * (27 * n + 1 + len("chr7_1\tA\tC")) + sum(len(str(s + 1)) - 1 for s in range(m))
    else:
        tmp = os.path.join(d, "pool.bgzf")
        synth.make_pool_file(tmp, n, m, pool=min(1024, m))
        text_bytes = 0
        with gzip.open(tmp, "rb") as src, gzip.open(path, "wb", compresslevel=4) as dst:
            src.readline()
            dst.write(("marker\tallele1\tallele2\t" + "\t".join("I%d\tI%d\tI%d" % (i, i, i) for i in range(n)) + "\n").encode())
            while True:
                buf = src.read(64 << 20)
                if not buf:
                    break
                text_bytes += len(buf)
                dst.write(buf)
        os.remove(tmp)
    res = {"individuals": n, "sites": m, "text_MB": round(text_bytes / 1e6, 1), "gz_MB": round(os.path.getsize(path) / 1e6, 1),
           "threads": reader_cy.host_threads(), "format": "bgzf" if a.bgzf else "gzip"}
    t0 = time.perf_counter()
    idx, _, sites = reader_cy.ensure_index(path)
    t = time.perf_counter() - t0
    res["index_pass"] = {"seconds": round(t, 3), "text_MB_per_s": round(text_bytes / 1e6 / t, 1), "sites_per_s": round(m / t)}
    for label, first in (() if a.only_device_inflate else (("parse_from_start", 0), ("parse_second_half_via_index", m // 2))):
        t0 = time.perf_counter()
        with reader_cy.BeagleStream(path, index=idx, first_row=first) as st:
            rows = sum(r.shape[0] for r, _ in st.chunks())
        t = time.perf_counter() - t0
        res[label] = {"seconds": round(t, 3), "sites": rows, "sites_per_s": round(rows / t),
                      "text_MB_per_s": round(text_bytes * rows / m / 1e6 / t, 1)}
    if a.device:
        from wgsassign_amd import device
        ctx = device.get_context()
        for label, mode, inflate in (("into_slabs_device_inflate_and_tokeniser", "device", "device"),
                                     ("into_slabs_host_inflate_device_tokeniser", "device", "host"),
                                     ("into_slabs_host_parser", "host", "host")):
            if inflate == "device" and not a.bgzf:
                continue
            if a.only_device_inflate and inflate != "device":
                continue
            os.environ["WGSASSIGN_INGEST"] = mode
            os.environ["WGSASSIGN_INFLATE"] = inflate
            best = None
            for _ in range(2):                      # the second run has the pinned buffers' pages and the file cache warm
                t0 = time.perf_counter()
                b, _, _, _ = reader_cy.stream_to_device(path, np.arange(n, dtype=np.int32) % 20, 20, ctx=ctx, names="ends")
                ctx.sync()
                t = time.perf_counter() - t0
                st = b.ingest_stats
                b.close()
                if best is None or t < best[0]:
                    best = (t, st)
            t, st = best
            res[label] = {"seconds": round(t, 3), "sites_per_s": round(m / t), "text_GB_per_s": round(text_bytes / 1e9 / t, 2)}
            if st:
                res[label].update({"waited_for_inflate_s": round(st["wait_s"], 3), "producer_inflate_s": round(st["inflate_s"], 3),
                                   "producer_newline_scan_s": round(st["scan_s"], 3), "device_ms_h2d_plus_tokeniser": round(st["device_ms"], 1),
                                   "lines_parsed_on_host": int(st["host_lines"]), "chunks": int(st["chunks"]),
                                   "device_inflate_kernel_ms": round(st["device_inflate_kernel_ms"], 1),
                                   "members_inflated_on_device": int(st["blocks_inflated_on_device"]),
                                   "members_left_to_host": int(st["blocks_left_to_host_inflater"]),
                                   "producer_read_s": round(st["read_s"], 3), "ingest_create_s": round(st["create_s"], 3),
                                   "ingest_next_s": round(st["next_s"], 3)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
