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
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402


def write_beagle(path, L, ids_path, IDs):
    m, n = L.shape[0], L.shape[1] // 2
    g2 = np.maximum(0.0, 1.0 - L[:, 0::2].astype(np.float64) - L[:, 1::2].astype(np.float64))
    full = np.empty((m, 3 * n), dtype=np.float64)
    full[:, 0::3], full[:, 1::3], full[:, 2::3] = L[:, 0::2], L[:, 1::2], g2
    with gzip.open(path, "wt", compresslevel=1) as fh:
        fh.write("marker\tallele1\tallele2\t" + "\t".join("Ind%d\tInd%d\tInd%d" % (i, i, i) for i in range(n)) + "\n")
        for s in range(m):
            fh.write("chr1_%d\t0\t1\t" % (s + 1) + "\t".join("%.6f" % v for v in full[s]) + "\n")
    with open(ids_path, "w") as fh:
        for a, b in IDs:
            fh.write("%s\t%s\n" % (a, b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--inds", type=int, default=100)
    ap.add_argument("--pops", type=int, default=5)
    ap.add_argument("--module", default="wgsassign_amd.WGSassign")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    L, IDs = synth.make_beagle(a.snps, a.inds, a.pops, seed=4242)
    with tempfile.TemporaryDirectory() as td:
        bg, ids = os.path.join(td, "x.beagle.gz"), os.path.join(td, "x.IDs.txt")
        t0 = time.perf_counter()
        write_beagle(bg, L, ids, IDs)
        t_write = time.perf_counter() - t0
        env = dict(os.environ)
        if a.module.startswith("wgsassign_amd"):      # the reference run must not see this repo's WGSassign/ alias package
            env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        res = {"module": a.module, "snps": a.snps, "inds": a.inds, "pops": a.pops, "file_mb": round(os.path.getsize(bg) / 1e6, 1),
               "write_s": round(t_write, 1), "digest": synth.digest(L)}
        for name, extra in (("reference_af", ["--get_reference_af"]), ("reference_af_loo", ["--get_reference_af", "--loo"])):
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
        res["loo_tsv_head"] = open(os.path.join(td, "reference_af_loo.pop_like_LOO.tsv")).read().splitlines()[1]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
