#!/usr/bin/env python3
"""One generated Beagle file through the command line with 1 process and with `--gpus N` (N ranks on GPU 0, TCP all-reduce):
every output file must be the same, byte for byte.

    python tools/compare_ranks.py --snps 1000000 --inds 200 --pops 5 --ranks 3 [--format bgzf] [--partitions 3]"""
import argparse
import filecmp
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_cli  # noqa: E402
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=1_000_000)
    ap.add_argument("--inds", type=int, default=200)
    ap.add_argument("--pops", type=int, default=5)
    ap.add_argument("--ranks", type=int, default=3)
    ap.add_argument("--partitions", type=int, default=3)
    ap.add_argument("--format", default="bgzf", choices=["gzip", "bgzf"])
    a = ap.parse_args()
    L, IDs = synth.make_beagle(a.snps, a.inds, a.pops, seed=4242)
    res = {"snps": a.snps, "inds": a.inds, "pops": a.pops, "ranks": a.ranks, "format": a.format}
    with tempfile.TemporaryDirectory() as td:
        bg, ids = os.path.join(td, "x.beagle.gz"), os.path.join(td, "x.IDs.txt")
        bench_cli.write_beagle(bg, L, ids, IDs, a.format)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env.update(WGSASSIGN_DEVICE="0", WGSASSIGN_COMM="socket", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""),
                   WGSASSIGN_INDEX_DIR=td)
        for tag, extra in (("one", []), ("many", ["--gpus", str(a.ranks)])):
            base = [sys.executable, "-m", "wgsassign_amd.WGSassign"] + extra + ["--beagle", bg]
            t0 = time.perf_counter()
            r = subprocess.run(base + ["--pop_af_IDs", ids, "--get_reference_af", "--ne_obs", "--loo", "--partition_sites", str(a.partitions),
                                       "--out", os.path.join(td, tag)], capture_output=True, text=True, env=env, cwd=td)
            assert r.returncode == 0, r.stderr[-3000:]
            r = subprocess.run(base + ["--pop_af_file", os.path.join(td, tag + ".pop_af.npy"), "--get_pop_like", "--out", os.path.join(td, tag)],
                               capture_output=True, text=True, env=env, cwd=td)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag + "_seconds"] = round(time.perf_counter() - t0, 2)
        same, differ = [], []
        for f in sorted(os.listdir(td)):
            if not f.startswith("one.") or f.endswith(".args"):
                continue
            g = "many." + f[4:]
            if f.endswith(".gz"):
                eq = gzip.open(os.path.join(td, f)).read() == gzip.open(os.path.join(td, g)).read()
            else:
                eq = filecmp.cmp(os.path.join(td, f), os.path.join(td, g), shallow=False)
            (same if eq else differ).append(f[4:])
        res["identical"], res["different"] = same, differ
    print(json.dumps(res))
    sys.exit(1 if differ else 0)


if __name__ == "__main__":
    main()
