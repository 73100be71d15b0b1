#!/usr/bin/env python3
"""End-to-end CLI wall time on a generated Beagle file (parse + --get_reference_af + --loo + --get_pop_like).

    python tools/bench_cli.py --snps 100000 --inds 100 --pops 5 [--module WGSassign.WGSassign | wgsassign_amd.WGSassign]

The file is generated deterministically (tests/synth.py), so the reference CLI (in the build
container: PYTHONPATH=/tmp/wgs_oracle, --module WGSassign.WGSassign) and this build (on the GPU
box) time the same input.
"""
import argparse
import gzip
import json
import os
import struct
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402


def _bgzf_block(chunk):
    """One BGZF member (htslib layout: 'BC' extra subfield with the member size, raw deflate, CRC32 + ISIZE)."""
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    payload = co.compress(chunk) + co.flush()
    bsize = 12 + 6 + len(payload) + 8
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + payload +
            struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def text_rows(L, row0, row1, names=None):
    """Rows [row0, row1) of the Beagle text, as bytes: every value printed as %.6f (vectorised: the values are
    six-decimal numbers already, so their digits are those of round(v * 1e6))."""
    blk = L[row0:row1].astype(np.float64)
    m, n = blk.shape[0], blk.shape[1] // 2
    full = np.empty((m, 3 * n), dtype=np.float64)
    full[:, 0::3], full[:, 1::3] = blk[:, 0::2], blk[:, 1::2]
    full[:, 2::3] = np.maximum(0.0, 1.0 - blk[:, 0::2] - blk[:, 1::2])
    v = np.rint(full * 1e6).astype(np.int64)
    tok = np.empty((m, 3 * n, 9), dtype=np.uint8)
    tok[:, :, 0] = 48 + v // 1_000_000
    tok[:, :, 1] = ord(".")
    r = v % 1_000_000
    for d in range(6):
        tok[:, :, 7 - d] = 48 + r % 10
        r //= 10
    tok[:, :, 8] = ord("\t")
    tok[:, -1, 8] = ord("\n")
    flat = tok.reshape(m, -1)
    if names is not None:
        return b"".join(names[row0 + s].encode() + b"\t0\t1\t" + flat[s].tobytes() for s in range(m))
    return b"".join(b"chr1_%d\t0\t1\t" % (row0 + s + 1) + flat[s].tobytes() for s in range(m))


def write_beagle(path, L, ids_path, IDs, fmt="gzip", names=None):
    """fmt = gzip: one deflate stream (what `gzip` writes); bgzf: 60 kB members (what ANGSD / bgzip write)."""
    m, n = L.shape[0], L.shape[1] // 2
    head = ("marker\tallele1\tallele2\t" + "\t".join("Ind%d\tInd%d\tInd%d" % (i, i, i) for i in range(n)) + "\n").encode()
    step = max(1, (64 << 20) // (27 * n))
    text_bytes = 0
    if fmt == "bgzf":
        from concurrent.futures import ThreadPoolExecutor     # zlib releases the GIL
        with open(path, "wb") as fh, ThreadPoolExecutor(min(len(os.sched_getaffinity(0)), 16)) as pool:
            pending = head
            for r0 in range(0, m, step):
                pending += text_rows(L, r0, min(m, r0 + step), names)
                cut = len(pending) - len(pending) % 60000
                chunks = [pending[i:i + 60000] for i in range(0, cut, 60000)]
                for blk in pool.map(_bgzf_block, chunks):
                    fh.write(blk)
                text_bytes += cut
                pending = pending[cut:]
            if pending:
                fh.write(_bgzf_block(pending))
                text_bytes += len(pending)
            fh.write(_bgzf_block(b""))
    else:
        with gzip.open(path, "wb", compresslevel=1) as fh:
            fh.write(head)
            text_bytes += len(head)
            for r0 in range(0, m, step):
                t = text_rows(L, r0, min(m, r0 + step), names)
                fh.write(t)
                text_bytes += len(t)
    with open(ids_path, "w") as fh:
        for a, b in IDs:
            fh.write("%s\t%s\n" % (a, b))
    return text_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--inds", type=int, default=100)
    ap.add_argument("--pops", type=int, default=5)
    ap.add_argument("--module", default="wgsassign_amd.WGSassign")
    ap.add_argument("--threads", type=int, default=1, help="-t of the command line (1 = automatic)")
    ap.add_argument("--format", default="gzip", choices=["gzip", "bgzf"])
    ap.add_argument("--no-loo", action="store_true", help="skip the --loo run (hours on the reference at 1M x 200)")
    a = ap.parse_args()
    L, IDs = synth.make_beagle(a.snps, a.inds, a.pops, seed=4242)
    with tempfile.TemporaryDirectory() as td:
        bg, ids = os.path.join(td, "x.beagle.gz"), os.path.join(td, "x.IDs.txt")
        t0 = time.perf_counter()
        text_bytes = write_beagle(bg, L, ids, IDs, a.format)
        t_write = time.perf_counter() - t0
        env = dict(os.environ)
        if a.module.startswith("wgsassign_amd"):      # the reference run must not see this repo's WGSassign/ alias package
            env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        res = {"module": a.module, "snps": a.snps, "inds": a.inds, "pops": a.pops, "format": a.format, "file_mb": round(os.path.getsize(bg) / 1e6, 1),
               "text_mb": round(text_bytes / 1e6, 1), "write_s": round(t_write, 1), "digest": synth.digest(L)}
        runs = [("reference_af", ["--get_reference_af"])] + ([] if a.no_loo else [("reference_af_loo", ["--get_reference_af", "--loo"])])
        for name, extra in runs:
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", a.module, "--beagle", bg, "--pop_af_IDs", ids, "--out",
                                os.path.join(td, name), "--threads", str(a.threads)] + extra, cwd=td, env=env,
                               capture_output=True, text=True)
            res[name + "_s"] = round(time.perf_counter() - t0, 2)
            if r.returncode != 0:
                res[name + "_error"] = r.stderr[-500:]
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", a.module, "--beagle", bg, "--pop_af_file", os.path.join(td, "reference_af.pop_af.npy"),
                            "--get_pop_like", "--out", os.path.join(td, "like"), "--threads", str(a.threads)], cwd=td, env=env,
                           capture_output=True, text=True)
        res["pop_like_s"] = round(time.perf_counter() - t0, 2)
        res["af_digest"] = synth.digest(np.load(os.path.join(td, "reference_af.pop_af.npy")))
        res["pop_like_head"] = open(os.path.join(td, "like.pop_like.txt")).readline().strip()
        if not a.no_loo:
            res["loo_tsv_head"] = open(os.path.join(td, "reference_af_loo.pop_like_LOO.tsv")).read().splitlines()[1]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
