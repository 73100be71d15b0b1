#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (the reference never travels to the GPU box):

    rm -rf /tmp/wgs_oracle && cp -r /root/reference /tmp/wgs_oracle && chmod -R u+w /tmp/wgs_oracle
    (cd /tmp/wgs_oracle && python3 setup.py build_ext --inplace)
    cd /tmp && PYTHONPATH=/tmp/wgs_oracle python3 /root/repo/tests/golden/make_golden.py

Everything written is data: inputs (or the seed that regenerates them via tests/synth.py,
with a digest of the generated input) and the outputs the reference computed for them.
Reference functions exercised: reader_cy.readBeagle (reader_cy.pyx:16), emMAF_cy.emMAF_update
/ rmse1d (emMAF_cy.pyx:10,26), emMAF.emMAF (emMAF.py:15), glassy_cy.loglike
(glassy_cy.pyx:12), glassy.assignLL / glassy.loo (glassy.py:18,47),
utils.partition_loglikes / filter_sites_to_common (utils.py:129,22) and the CLI
(WGSassign.py:109) for the text outputs.
"""
import contextlib
import io
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

from WGSassign import emMAF, emMAF_cy, glassy, glassy_cy, reader_cy, utils  # noqa: E402

DATA = os.path.join(HERE, "data")
BREED = os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz")
BREED80 = os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each_subset_80percent_sites.beagle.gz")
BREED_IDS = os.path.join(DATA, "amre.breeding.ind85.reference_k5.IDs.txt")
NONBREED = os.path.join(DATA, "amre.nonbreeding.ind34.ds_2x.sites-filter.top_50_each.beagle.gz")


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def captured(fn, *a, **kw):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **kw)
    return out, buf.getvalue()


def iters_from(text):
    return [int(x) for x in re.findall(r"converged at iteration: (\d+)", text)]


def gather(L, idx):
    cols = np.sort(np.concatenate((idx * 2, idx * 2 + 1)))
    return np.ascontiguousarray(L[:, cols])


def ref_fit(L, IDs, maf_iter=200, maf_tole=1e-4, t=4):
    """WGSassign.py:213-242 driven through the reference's own emMAF."""
    pops = np.unique(IDs[:, 1])
    m = L.shape[0]
    af = np.empty((m, len(pops)), dtype=np.float32)
    f_raw = np.empty((len(pops), m), dtype=np.float32)
    iters = []
    for i, p in enumerate(pops):
        idx = np.argwhere(IDs[:, 1] == p).reshape(-1)
        L_pop = gather(L, idx)
        f, txt = captured(emMAF.emMAF, L_pop, maf_iter, maf_tole, t)
        it = iters_from(txt)
        iters.append(it[0] if it else -1)
        f_raw[i] = f
        n_pop = L_pop.shape[1] // 2
        lo = 1 / (2 * (n_pop + 1))
        hi = 1 - lo
        f = f.copy()
        f[f < lo] = lo
        f[f > hi] = hi
        af[:, i] = f
    return pops, f_raw, af, np.array(iters, dtype=np.int32)


def g1_g2_amre_fit():
    L, samples, sites = reader_cy.readBeagle(BREED)
    IDs = np.loadtxt(BREED_IDS, delimiter="\t", dtype="str")
    pops, f_raw, af, iters = ref_fit(L, IDs)
    print("AMRE iters", iters, synth.digest(L), synth.digest(af))
    save("amre_fit.npz", L=L, samples=np.array(samples), sites=np.array(sites), IDs=IDs,
         pops=pops, f_raw=f_raw, pop_af=af, iters=iters)
    # G2: per-iteration trace for one population (South)
    k = list(pops).index("South")
    idx = np.argwhere(IDs[:, 1] == "South").reshape(-1)
    L_pop = gather(L, idx)
    f = np.full(L.shape[0], 0.25, dtype=np.float32)
    f_prev = f.copy()
    trace, diffs = [], []
    for it in range(200):
        emMAF_cy.emMAF_update(L_pop, f, 1)
        d = emMAF_cy.rmse1d(f, f_prev)
        trace.append(f.copy())
        diffs.append(d)
        if d < 1e-4:
            break
        f_prev = f.copy()
    assert len(trace) == iters[k]
    save("amre_trace.npz", L_pop=L_pop, trace=np.array(trace), diffs=np.array(diffs, dtype=np.float64))
    return L, samples, sites, IDs, af


def g3_amre_assign(af):
    L, samples, sites = reader_cy.readBeagle(NONBREED)
    logl, _ = captured(glassy.assignLL, L, af.copy(), 4)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "x.txt")
        np.savetxt(p, logl, fmt="%.7f")
        text = open(p).read()
    # one per-site vector, to pin the kernel below the sum
    vec = np.zeros(L.shape[0], dtype=np.float32)
    glassy_cy.loglike(L, af.copy(), vec, 1, 3, 2)
    print("AMRE assign", synth.digest(logl), logl[0])
    save("amre_assign.npz", L=L, samples=np.array(samples), sites=np.array(sites),
         logl=logl, text=np.array(text), vec_i3_k2=vec)


def g4_amre_loo(L, sites, IDs, af):
    out = {}
    for P in (1, 3):
        af_mut = af.copy()
        (ll, parts), _ = captured(glassy.loo, L, af_mut, IDs, 4, 200, 1e-4, None, P)
        out["loo_P%d" % P] = ll
        out["parts_P%d" % P] = parts
        out["af_after_P%d" % P] = af_mut
    assert np.array_equal(out["loo_P1"], out["loo_P3"])
    # downsampled variant: WGSassign.py:172-198 then glassy.loo with downsampled_L
    L_ds, samples_ds, sites_ds = reader_cy.readBeagle(BREED80)
    (L_f, sites_f), txt1 = captured(utils.filter_sites_to_common, L, sites, sites_ds)
    (L_ds_f, sites_ds_f), _ = captured(utils.filter_sites_to_common, L_ds, sites_ds, sites_f)
    assert sites_f == sites_ds_f
    mask = np.isin(np.array(sites), sites_ds)
    # the reference refits the population AFs on the filtered L before LOO (WGSassign.py:189 then :205)
    pops, f_raw_f, af_f, iters_f = ref_fit(L_f, IDs)
    af_mut = af_f.copy()
    (ll_ds, parts_ds), _ = captured(glassy.loo, L_f, af_mut, IDs, 4, 200, 1e-4, L_ds_f, 1)
    out.update(mask=mask, L_ds=L_ds, sites_ds=np.array(sites_ds), filter_text=np.array(txt1),
               pop_af_filtered=af_f, iters_filtered=iters_f, loo_ds=ll_ds, af_after_ds=af_mut)
    print("AMRE LOO row0", out["loo_P1"][0], "filtered", int((~mask).sum()))
    save("amre_loo.npz", **out)


def g4b_cli_text():
    """Text artefacts of the CLI (WGSassign.py:243-247,280-294,306; utils.py:113-121)."""
    res = {}
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ)
        base = [sys.executable, "-m", "WGSassign.WGSassign"]
        r = subprocess.run(base + ["--beagle", BREED, "--pop_af_IDs", BREED_IDS, "--get_reference_af", "--loo",
                                   "--partition_sites", "3", "--out", os.path.join(td, "ref"), "--threads", "2"],
                           cwd=td, env=env, capture_output=True, text=True, check=True)
        res["stdout_ref"] = r.stdout.replace(td, "<TMP>")
        res["pop_names"] = open(os.path.join(td, "ref.pop_names.txt")).read()
        res["loo_tsv"] = open(os.path.join(td, "ref.pop_like_LOO.tsv")).read()
        import gzip
        res["parts_tsv"] = gzip.open(os.path.join(td, "ref.pop_like_LOO_partitions_3.tsv.gz"), "rt").read()
        res["pop_af_npy"] = np.load(os.path.join(td, "ref.pop_af.npy"))
        r = subprocess.run(base + ["--beagle", NONBREED, "--pop_af_file", os.path.join(td, "ref.pop_af.npy"),
                                   "--get_pop_like", "--out", os.path.join(td, "nb"), "--threads", "2"],
                           cwd=td, env=env, capture_output=True, text=True, check=True)
        res["stdout_like"] = r.stdout.replace(td, "<TMP>")
        res["pop_like_txt"] = open(os.path.join(td, "nb.pop_like.txt")).read()
    save("amre_cli.npz", **{k: (v if isinstance(v, np.ndarray) else np.array(v)) for k, v in res.items()})


def g5_edge():
    """Tiny hand-built cases: exact GL corners, f -> 0, n_pop 1/2, NaN, iter exhausted."""
    rows = np.array([
        [1, 0], [0, 0], [0, 1], [0.333333, 0.333333], [0.6, 0.400001], [0.000001, 0.999999],
        [0.5, 0.5], [0.25, 0.5], [0.999999, 0.000001], [0, 0.5]], dtype=np.float32)
    out = {}
    # (a) each SNP = one corner repeated over n individuals, for several n
    for n in (1, 2, 5):
        L = np.ascontiguousarray(np.repeat(rows, n, axis=0).reshape(len(rows), 2 * n))
        f = np.full(len(rows), 0.25, dtype=np.float32)
        fs = []
        for _ in range(4):
            emMAF_cy.emMAF_update(L, f, 1)
            fs.append(f.copy())
        out["corner_n%d_L" % n] = L
        out["corner_n%d_f" % n] = np.array(fs)
    # (b) mixed SNPs incl. an all-hom-ref SNP (f -> 0.0) and a 0/0 row (NaN)
    rng = np.random.Generator(np.random.PCG64(5))
    n = 7
    L = np.round(rng.dirichlet([1, 1, 1], size=(12, n))[:, :, :2], 6).astype(np.float32).reshape(12, 2 * n)
    L[0] = np.tile([1, 0], n)
    L[1] = np.tile([0, 0], n)          # p0=p1=0, p2>0 -> f -> 1
    L[2] = np.tile([0, 1], n)
    L[3, :2] = [0, 0]
    L = np.ascontiguousarray(L)
    f, txt = captured(emMAF.emMAF, L, 200, 1e-4, 1)
    out.update(mixed_L=L, mixed_f=f, mixed_iters=np.array(iters_from(txt) or [-1]))
    # start from f = 0 and f = 1 exactly (0/0 -> NaN for some rows)
    for name, f0 in (("zero", 0.0), ("one", 1.0)):
        f = np.full(12, f0, dtype=np.float32)
        with np.errstate(all="ignore"):
            emMAF_cy.emMAF_update(L, f, 1)
        out["mixed_from_" + name] = f
    # (c) iter exhausted: 3 iterations only
    f, txt = captured(emMAF.emMAF, L, 3, 1e-12, 1)
    out.update(exhaust_f=f, exhaust_printed=np.array(len(iters_from(txt))))
    # (d) rmse1d corner values
    v1 = np.array([0.25, 0.5, np.nan], dtype=np.float32)
    v2 = np.array([0.5, 0.5, 0.1], dtype=np.float32)
    out["rmse_nan"] = np.array(emMAF_cy.rmse1d(v1, v2))
    out["rmse_small"] = np.array(emMAF_cy.rmse1d(v1[:2].copy(), v2[:2].copy()))
    # (e) loglike corners: A in {lo clamp, 0.5, hi clamp, 0, 1}
    A = np.ascontiguousarray(np.tile(np.array([[1 / 16, 0.5, 15 / 16, 0.0, 1.0]], dtype=np.float32), (12, 1)))
    vecs = np.zeros((n, 5, 12), dtype=np.float32)
    with np.errstate(all="ignore"):
        for i in range(n):
            for k in range(5):
                glassy_cy.loglike(L, A, vecs[i, k], 1, i, k)
        logl, _ = captured(glassy.assignLL, L, A, 1)
    out.update(ll_A=A, ll_vecs=vecs, ll_mat=logl)
    # accumulate-into semantics: second call adds onto the first (glassy_cy.pyx:21)
    v = np.zeros(12, dtype=np.float32)
    glassy_cy.loglike(L, A, v, 1, 2, 1)
    glassy_cy.loglike(L, A, v, 1, 4, 2)
    out["ll_accum"] = v
    # (f) LOO with a population of size 1 -> NaN column that sticks (glassy.py:69-89)
    Ls, IDs = synth.make_beagle(40, 7, 2, seed=11)
    IDs[:, 1] = np.array(["a", "a", "a", "b", "a", "a", "a"])
    pops, f_raw, af, iters = ref_fit(Ls, IDs)
    af_mut = af.copy()
    with np.errstate(all="ignore"):
        (ll, parts), _ = captured(glassy.loo, Ls, af_mut, IDs, 1, 20, 1e-4, None, 2)
    out.update(single_L=Ls, single_IDs=IDs, single_af=af, single_loo=ll, single_parts=parts,
               single_af_after=af_mut)
    save("edge.npz", **out)


def g6_accum_order():
    """One update + converged f for large n (documents accumulation order)."""
    out = {}
    for n in (85, 200, 1000, 2000):
        m = 20000 if n <= 200 else 4000
        L, _ = synth.make_beagle(m, n, 1, seed=600 + n)
        f1 = np.full(m, 0.25, dtype=np.float32)
        emMAF_cy.emMAF_update(L, f1, 4)
        f, txt = captured(emMAF.emMAF, L, 200, 1e-4, 4)
        out["n%d_digest" % n] = np.array(synth.digest(L))
        out["n%d_m" % n] = np.array(m)
        out["n%d_f1" % n] = f1
        out["n%d_f" % n] = f
        out["n%d_iters" % n] = np.array(iters_from(txt) or [-1])
    save("accum.npz", **out)


def g7_rmse():
    out = {}
    for m in (449, 100_000, 1_000_000, 10_000_000):
        rng = np.random.Generator(np.random.PCG64(700 + m))
        v1 = rng.random(m, dtype=np.float32)
        v2 = (v1 + rng.normal(0, 1.2e-4, m).astype(np.float32)).astype(np.float32)
        out["m%d" % m] = np.array(emMAF_cy.rmse1d(v1, v2))
    # magnitudes spanning many binades and exact ties
    rng = np.random.Generator(np.random.PCG64(77))
    m = 300_000
    v1 = rng.random(m, dtype=np.float32)
    v2 = (v1 + (rng.normal(0, 1, m) * 10.0 ** rng.uniform(-7, -1, m)).astype(np.float32)).astype(np.float32)
    out["wide"] = np.array(emMAF_cy.rmse1d(v1, v2))
    save("rmse.npz", **out)


def g8_synth_mid():
    m, n, K = 50_000, 100, 5
    L, IDs = synth.make_beagle(m, n, K)
    pops, f_raw, af, iters = ref_fit(L, IDs, t=8)
    logl, _ = captured(glassy.assignLL, L[:5000], af[:5000].copy(), 8)
    # LOO on a subsample of SNPs keeps the fixture generator quick (100 EM re-fits)
    ms = 8000
    Ls = np.ascontiguousarray(L[:ms])
    pops2, f_raw2, af2, iters2 = ref_fit(Ls, IDs, t=8)
    af_mut = af2.copy()
    (ll, parts), _ = captured(glassy.loo, Ls, af_mut, IDs, 8, 200, 1e-4, None, 4)
    print("synth mid iters", iters, iters2)
    save("synth_mid.npz", digest=np.array(synth.digest(L)), m=np.array(m), n=np.array(n), K=np.array(K),
         pop_af=af, iters=iters, logl_5000=logl, loo_ms=np.array(ms), loo_pop_af=af2, loo_iters=iters2,
         loo=ll, loo_parts=parts, loo_af_after=af_mut)
    # interleaved population labels (strided gathers)
    Li, IDi = synth.make_beagle(6000, 37, 4, seed=88, interleave=True)
    popsi, f_rawi, afi, itersi = ref_fit(Li, IDi)
    af_mut = afi.copy()
    (lli, partsi), _ = captured(glassy.loo, Li, af_mut, IDi, 8, 200, 1e-4, None, 1)
    save("synth_interleaved.npz", digest=np.array(synth.digest(Li)), pop_af=afi, iters=itersi, loo=lli,
         af_after=af_mut)


def g9_fisher(L, IDs, af):
    """--ne_obs: fisher.fisher_obs / fisher_obs_ind (fisher.py:11-60, fisher_cy.pyx:12-65)
    and the text artefacts of WGSassign.py:252-274."""
    from WGSassign import fisher
    f_obs, ne_obs = fisher.fisher_obs(L, af.copy(), IDs, 4)
    ne_ind = fisher.fisher_obs_ind(L, af.copy(), IDs, 4)
    res = dict(f_obs=f_obs, ne_obs=ne_obs, ne_ind=ne_ind)
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([sys.executable, "-m", "WGSassign.WGSassign", "--beagle", BREED, "--pop_af_IDs", BREED_IDS,
                            "--get_reference_af", "--ne_obs", "--out", os.path.join(td, "ne"), "--threads", "2"],
                           cwd=td, capture_output=True, text=True, check=True)
        res["stdout"] = np.array(r.stdout.replace(td, "<TMP>"))
        res["ne_obs_txt"] = np.array(open(os.path.join(td, "ne.ne_obs.txt")).read())
        res["ne_ind_txt"] = np.array(open(os.path.join(td, "ne.ne_ind.txt")).read())
        assert np.load(os.path.join(td, "ne.fisher_obs.npy")).tobytes() == f_obs.tobytes()
        assert np.load(os.path.join(td, "ne.ne_obs.npy")).tobytes() == ne_obs.tobytes()
    # synthetic, larger n per population, interleaved labels
    Ls, IDs_s = synth.make_beagle(5000, 61, 3, seed=31, interleave=True)
    pops, f_raw, af_s, iters = ref_fit(Ls, IDs_s)
    f2, ne2 = fisher.fisher_obs(Ls, af_s.copy(), IDs_s, 4)
    ni2 = fisher.fisher_obs_ind(Ls, af_s.copy(), IDs_s, 4)
    res.update(synth_digest=np.array(synth.digest(Ls)), synth_af=af_s, synth_f_obs=f2, synth_ne_obs=ne2, synth_ne_ind=ni2)
    save("fisher.npz", **res)



def g10_cli_downsampled():
    """CLI run with --loo_downsampled_beagle (WGSassign.py:172-198, 276-294): stdout and the TSV."""
    res = {}
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([sys.executable, "-m", "WGSassign.WGSassign", "--beagle", BREED, "--pop_af_IDs", BREED_IDS,
                            "--get_reference_af", "--loo", "--loo_downsampled_beagle", BREED80, "--out",
                            os.path.join(td, "ds"), "--threads", "2"], cwd=td, capture_output=True, text=True, check=True)
        res["stdout"] = np.array(r.stdout.replace(td, "<TMP>").replace(DATA, "<DATA>"))
        res["loo_tsv"] = np.array(open(os.path.join(td, "ds.pop_like_LOO_downsampled.tsv")).read())
        res["pop_af_npy"] = np.load(os.path.join(td, "ds.pop_af.npy"))
    save("amre_cli_downsampled.npz", **res)


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "downsampled":
    g10_cli_downsampled()
    sys.exit(0)

if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "fisher":
    _g = np.load(os.path.join(HERE, "amre_fit.npz"))
    g9_fisher(_g["L"], _g["IDs"], _g["pop_af"])
    sys.exit(0)

if __name__ == "__main__":
    L, samples, sites, IDs, af = g1_g2_amre_fit()
    g3_amre_assign(af)
    g4_amre_loo(L, sites, IDs, af)
    g4b_cli_text()
    g5_edge()
    g6_accum_order()
    g7_rmse()
    g8_synth_mid()
    g9_fisher(L, IDs, af)
    g10_cli_downsampled()


