#!/usr/bin/env python3
"""bench.py's ingest leg alone (extra.paths.config5_streamed_ingest_100kx2000), with every counter of the pipeline:
   python tools/bench_ingest_dev.py [--inds 2000 --sites 100000 --runs 3]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import beagle_files  # noqa: E402
from wgsassign_amd import device, reader_cy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inds", type=int, default=2000)
    ap.add_argument("--sites", type=int, default=100000)
    ap.add_argument("--pops", type=int, default=20)
    ap.add_argument("--runs", type=int, default=3)
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="wgs_ingest_")
    os.environ["WGSASSIGN_INDEX_DIR"] = d
    try:
        path = os.path.join(d, "shard.beagle.gz")
        text_bytes, vals, pick = beagle_files.write_lowdepth_bgzf(path, a.inds, a.sites)
        ctx = device.get_context()
        group_of = (np.arange(a.inds) % a.pops).astype(np.int32)
        t0 = time.perf_counter()
        reader_cy.ensure_index(path)
        res = {"text_MB": round(text_bytes / 1e6), "file_MB": round(os.path.getsize(path) / 1e6), "index_pass_seconds": round(time.perf_counter() - t0, 4), "runs": []}
        for _ in range(a.runs):
            t0 = time.perf_counter()
            b, _, _, _ = reader_cy.stream_to_device(path, group_of, a.pops, ctx=ctx, names="ends")
            ctx.sync()
            dt = time.perf_counter() - t0
            st = {k: round(float(v), 4) for k, v in b.ingest_stats.items()}
            same = all(b.download_rows(r, 1).tobytes() == vals[pick[r]].tobytes() for r in (0, a.sites // 3, a.sites - 1))
            b.close()
            res["runs"].append({"seconds": round(dt, 4), "rows_equal_source": bool(same), **st})
        res["cold_one_pass"] = []
        for _ in range(a.runs):
            for f in reader_cy.index_paths(path):
                if os.path.exists(f):
                    os.unlink(f)
            t0 = time.perf_counter()
            b, _, _, m_seen = reader_cy.stream_to_device(path, group_of, a.pops, ctx=ctx, names="ends")
            ctx.sync()
            dt = time.perf_counter() - t0
            same = m_seen == a.sites and all(b.download_rows(r, 1).tobytes() == vals[pick[r]].tobytes() for r in (0, a.sites // 3, a.sites - 1))
            b.close()
            res["cold_one_pass"].append({"seconds": round(dt, 4), "rows_equal_source": bool(same), "index_left": os.path.exists(reader_cy.index_paths(path)[0])})
        reader_cy.ensure_index(path)
        # the same steps one by one (reader_cy.stream_to_device), each with its own clock
        from wgsassign_amd.device import DeviceBeagle
        from wgsassign_amd import _lib
        lib = _lib.load()
        spent = {}
        for fn in ("wgs_ingest_destroy", "wgs_ingest_create", "wgs_ingest_next", "wgs_reader_close"):
            def wrap(name, real):
                def timed(*a):
                    t = time.perf_counter()
                    try:
                        return real(*a)
                    finally:
                        spent[name] = spent.get(name, 0.0) + time.perf_counter() - t
                return timed
            setattr(lib, fn, wrap(fn, getattr(lib, fn)))
        for _ in range(2):
            steps = {}
            spent.clear()
            t0 = time.perf_counter()

            def lap(name):
                nonlocal t0
                ctx.sync()
                t1 = time.perf_counter()
                steps[name] = round(t1 - t0, 4)
                t0 = t1
            index, _, m_file = reader_cy.ensure_index(path)
            lap("ensure_index")
            st = reader_cy.BeagleStream(path, None, index=index, first_row=0)
            lap("open_stream")
            b = DeviceBeagle(m_file, st.n, group_of, a.pops, site0=0, ctx=ctx)
            lap("device_matrix")
            rows = 0
            for nrows, names in st.ingest(b, 0, m_file, None):
                rows += nrows
            lap("ingest_loop")
            st.close()
            lap("close_stream")
            b.close()
            lap("close_matrix")
            steps.update({"in_" + k: round(v, 4) for k, v in spent.items()})
            steps["ingest_create_s"] = round(st.ingest_stats["create_s"], 4)
            steps["ingest_next_s"] = round(st.ingest_stats["next_s"], 4)
            res.setdefault("steps", []).append(steps)
        print(json.dumps(res, indent=1))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
