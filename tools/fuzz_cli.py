#!/usr/bin/env python3
"""Random end-to-end cases through the command line, against the oracle pipeline (GPU needed):

    python tools/fuzz_cli.py [cases] [seed] [ranks]

ranks > 1: every command runs as `WGSassign --gpus ranks` (one process per rank, all on GPU 0, TCP all-reduce), i.e.
SNP-sharded -- the files must be the same.

Each case draws a shape (individuals, populations incl. populations of one, SNPs, partitions), writes a Beagle file
(plain gzip or BGZF, tabs or blanks, with or without a final newline) and an ID file, runs
    WGSassign --get_reference_af --loo --partition_sites P [--ne_obs]    and    WGSassign --get_pop_like
in-process and compares every output file with what the oracle (oracle/oracle.py: the CPU restatement pinned to the
real reference) computes from the same parsed matrix, written by the same writers the reference uses (np.save,
np.savetxt, the TSV of utils.py:49-123): .pop_af.npy and .fisher_obs.npy / .ne_obs.npy byte for byte, the text files
character for character -- which holds only if every float32 behind them is identical."""
import contextlib
import gzip
import io
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_cli  # noqa: E402
import synth  # noqa: E402
from oracle import oracle  # noqa: E402
from wgsassign_amd import WGSassign, utils  # noqa: E402


def write_case(rng, d, L, IDs):
    path = os.path.join(d, "x.beagle.gz")
    fmt = "bgzf" if rng.random() < 0.5 else "gzip"
    bench_cli.write_beagle(path, L, os.path.join(d, "ids.txt"), IDs, fmt)
    if fmt == "gzip" and rng.random() < 0.5:          # blanks instead of tabs, no final newline
        text = gzip.open(path, "rt").read().replace("\t", " ").rstrip("\n")
        with gzip.open(path, "wt", newline="") as fh:
            fh.write(text)
    return path, fmt


RANKS = 1


def run(argv):
    if RANKS > 1:
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env.update(WGSASSIGN_DEVICE="0", WGSASSIGN_COMM="socket", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-m", "wgsassign_amd.WGSassign", "--gpus", str(RANKS)] + argv, capture_output=True, text=True,
                           env=env, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        return r.stdout
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        WGSassign.main(argv)
    return buf.getvalue()


def expect_tsv(d, name, mat, samples, pops, IDs, P, part_col):
    f = os.path.join(d, name)
    with contextlib.redirect_stdout(io.StringIO()):
        utils.write_ass_mats(f, mat, samples, pops, partition_count=P, print_part_column=part_col, sample_locations=IDs[:, 1], doing_LOO=True)
    return (gzip.open(f, "rt") if f.endswith(".gz") else open(f)).read()


def ne_obs_txt(pops, ne_obs):
    """WGSassign.py:268-271: population names over the column means, as strings"""
    t = np.empty((2, len(pops)), dtype=np.dtype("U25"))
    t[0, :] = pops
    t[1, :] = np.mean(ne_obs, axis=0)
    sio = io.StringIO()
    np.savetxt(sio, t, fmt="%s")
    return sio.getvalue()


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    global RANKS
    RANKS = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        K = int(rng.integers(1, 7))
        sizes = rng.integers(1, 9, size=K)
        if rng.random() < 0.5:
            sizes[rng.integers(K)] = 1                                  # a population of one: NaN column under --loo
        if sizes.sum() < 2:
            sizes[0] = 2            # a one-line ID file makes np.loadtxt return a 1-D array: the reference fails there too
        labels = np.repeat(np.arange(K), sizes)
        rng.shuffle(labels)
        n = len(labels)
        m = int(rng.choice([1, 63, 64, 65, 449, 1000, 4097, 9000, 20_011] if RANKS == 1 else [63, 449, 4097, 9000, 20_011, 40_000, 100_003]))
        P = int(rng.choice([1, 2, 3, 7]))
        L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=int(rng.integers(1 << 30)), depth=float(rng.choice([0.5, 2.0, 6.0])))
        ne = bool(rng.random() < 0.5)
        ds = bool(rng.random() < 0.35) and m >= 63                 # --loo_downsampled_beagle: a subset of the sites, other GLs
        with tempfile.TemporaryDirectory() as d:
            os.environ["WGSASSIGN_INDEX_DIR"] = d
            path, fmt = write_case(rng, d, L, IDs)
            ids = os.path.join(d, "ids.txt")
            out = os.path.join(d, "run")
            argv = ["--beagle", path, "--pop_af_IDs", ids, "--get_reference_af", "--loo", "--out", out]
            L_ds = None
            if ds:
                keep = rng.random(m) < 0.8
                keep[int(rng.integers(m))] = True
                L2, _ = synth.make_beagle_for_labels(m, labels, K, seed=int(rng.integers(1 << 30)), depth=0.7)
                L_ds = np.ascontiguousarray(L2[keep])
                names = ["chr1_%d" % (s + 1) for s in np.flatnonzero(keep)]
                bench_cli.write_beagle(os.path.join(d, "ds.beagle.gz"), L_ds, os.path.join(d, "ids2.txt"), IDs,
                                       "bgzf" if rng.random() < 0.5 else "gzip", names)
                argv += ["--loo_downsampled_beagle", os.path.join(d, "ds.beagle.gz")]
                L = np.ascontiguousarray(L[keep])                   # WGSassign.py:172-198: the reference keeps the common sites only
            if P > 1:
                argv += ["--partition_sites", str(P)]
            if ne:
                argv += ["--ne_obs"]
            with np.errstate(all="ignore"):
                run(argv)
                if not ds:          # (the frequency file of a downsampled run covers the common sites only)
                    run(["--beagle", path, "--pop_af_file", out + ".pop_af.npy", "--get_pop_like", "--out", out])
                # the oracle pipeline on the same matrix
                pops, af, _, _ = oracle.fit_reference_af(L, IDs, t=4)
                samples = ["Ind%d" % i for i in range(n)]
                wrong = []

                def check(name, same):
                    if not same:
                        wrong.append(name)
                check("pop_af.npy", np.load(out + ".pop_af.npy").tobytes() == af.tobytes())
                check("pop_names.txt", open(out + ".pop_names.txt").read() == "".join(p + "\n" for p in pops))
                ll_o, parts_o = oracle.loo(L, af.copy(), IDs, 4, 200, 1e-4, L_ds, P)
                sfx = "_downsampled" if ds else ""
                check("pop_like_LOO.tsv", open(out + ".pop_like_LOO%s.tsv" % sfx).read() == expect_tsv(d, "e.tsv", ll_o, samples, pops, IDs, 1, False))
                if P > 1:
                    got = gzip.open(out + ".pop_like_LOO%s_partitions_%d.tsv.gz" % (sfx, P), "rt").read()
                    check("partitions.tsv.gz", got == expect_tsv(d, "e.tsv.gz", parts_o, samples, pops, IDs, P, True))
                if not ds:
                    sio = io.StringIO()
                    np.savetxt(sio, oracle.assignLL(L, af.copy(), 4), fmt="%.7f")
                    check("pop_like.txt", open(out + ".pop_like.txt").read() == sio.getvalue())
                if ne:
                    f_o, ne_o = oracle.fisher_obs(L, af.copy(), IDs, 4)
                    check("fisher_obs.npy", np.load(out + ".fisher_obs.npy").tobytes() == f_o.tobytes())
                    check("ne_obs.npy", np.load(out + ".ne_obs.npy").tobytes() == ne_o.tobytes())
                    check("ne_obs.txt", open(out + ".ne_obs.txt").read() == ne_obs_txt(pops, ne_o))
                    sio = io.StringIO()
                    np.savetxt(sio, oracle.fisher_obs_ind(L, af.copy(), IDs, 4).reshape(-1, 1), fmt="%.7f")
                    check("ne_ind.txt", open(out + ".ne_ind.txt").read() == sio.getvalue())
        ok = not wrong
        bad += not ok
        print("case %3d  m=%6d n=%3d K=%d sizes=%s P=%d %s ne=%d ds=%d  %s" % (c, m, n, K, list(map(int, sizes)), P, fmt, ne, ds, "ok" if ok else "MISMATCH " + " ".join(wrong)), flush=True)
    print("%d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
