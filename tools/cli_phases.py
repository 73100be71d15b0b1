"""Where an end-to-end command-line run spends its wall time: writes a simulated low-depth Beagle file (tools/bench_cli.py),
runs `--get_reference_af` on it under cProfile in a child process and prints the heaviest calls by cumulative time.
    python tools/cli_phases.py --snps 1000000 --inds 200 --format bgzf"""
import argparse
import json
import os
import pstats
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bench_cli  # noqa: E402
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=1_000_000)
    ap.add_argument("--inds", type=int, default=200)
    ap.add_argument("--pops", type=int, default=5)
    ap.add_argument("--format", default="bgzf", choices=["gzip", "bgzf"])
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    L, IDs = synth.make_beagle(a.snps, a.inds, a.pops, seed=4242)
    with tempfile.TemporaryDirectory() as td:
        bg, ids = os.path.join(td, "x.beagle.gz"), os.path.join(td, "x.IDs.txt")
        bench_cli.write_beagle(bg, L, ids, IDs, a.format)
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        walls = []
        for run in range(2):                       # the second run finds the index cache and a warm page cache
            prof = os.path.join(td, "run%d.prof" % run)
            t0 = time.perf_counter()
            subprocess.run([sys.executable, "-m", "cProfile", "-o", prof, "-m", "wgsassign_amd.WGSassign", "--beagle", bg, "--pop_af_IDs", ids,
                            "--out", os.path.join(td, "ref%d" % run), "--get_reference_af"], cwd=td, env=env, capture_output=True, text=True, check=True)
            walls.append(round(time.perf_counter() - t0, 3))
        print(json.dumps({"snps": a.snps, "inds": a.inds, "format": a.format, "wall_s_first_and_second_run": walls}))
        st = pstats.Stats(prof)
        st.sort_stats("cumulative").print_stats(a.top)


if __name__ == "__main__":
    main()
