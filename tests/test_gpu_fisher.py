"""--ne_obs: observed Fisher information / effective sample sizes on the device
against the golden vectors of the real reference (fisher.py / fisher_cy.pyx)."""
import contextlib
import io
import os

import numpy as np
import pytest

import synth
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def test_fisher_obs_bit_exact(golden):
    from wgsassign_amd import fisher
    g, fit = golden("fisher.npz"), golden("amre_fit.npz")
    f_obs, ne_obs = fisher.fisher_obs(fit["L"], fit["pop_af"].copy(), fit["IDs"], 1)
    assert same(f_obs, g["f_obs"]) and same(ne_obs, g["ne_obs"])
    ne_ind = fisher.fisher_obs_ind(fit["L"], fit["pop_af"].copy(), fit["IDs"], 1)
    assert same(ne_ind, g["ne_ind"])                      # np.mean formed on the device as NumPy forms it
    L, IDs = synth.make_beagle(5000, 61, 3, seed=31, interleave=True)
    f_obs, ne_obs = fisher.fisher_obs(L, g["synth_af"].copy(), IDs, 1)
    assert same(f_obs, g["synth_f_obs"]) and same(ne_obs, g["synth_ne_obs"])
    ne_ind = fisher.fisher_obs_ind(L, g["synth_af"].copy(), IDs, 1)
    assert same(ne_ind, g["synth_ne_ind"])


@pytest.mark.parametrize("m", [1, 5, 8, 9, 127, 128, 129, 257, 1000, 4097, 8192, 8193, 8197, 16384 + 3, 100_003, 2_000_000])
def test_device_mean_is_numpy_mean_at_every_length(m):
    """np.mean of a float32 row: chunks of 8192 elements added in order, each summed pairwise -- blocks of at most 128
    elements with eight interleaved accumulators, halves split at multiples of 8 -- then divided in float64.  The device
    forms the same tree (csrc/em_kernels.hip: pairwise_leaf_kernel, pairwise_combine_kernel), so its means equal np.mean
    of the downloaded rows bit for bit at every length, around every block and chunk edge and at 2M sites."""
    from wgsassign_amd import _lib, device, fisher
    n, K = 7, 2
    labels = np.arange(n) % K
    L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=m % 1000 + 1)
    rng = np.random.default_rng(m)
    af = np.clip(rng.random((m, K)), 0.02, 0.98).astype(np.float32)
    fast = fisher.fisher_obs_ind(L, af, IDs, 1)
    host = fisher.fisher_obs_ind(L, af, IDs, 1, host_mean=True)
    assert fast.dtype == np.float32 and same(fast, host), m
    # one individual per call and a budget that forces batches of 2 give the same values
    assert same(fisher.fisher_obs_ind(L, af, IDs, 1, exact_budget_bytes=8 * m), fast)


def test_cli_ne_obs(tmp_path, golden):
    from wgsassign_amd import WGSassign
    g = golden("fisher.npz")
    data = os.path.join(GOLDEN, "data")
    out = str(tmp_path / "ne")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        WGSassign.main(["--beagle", os.path.join(data, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"),
                        "--pop_af_IDs", os.path.join(data, "amre.breeding.ind85.reference_k5.IDs.txt"),
                        "--get_reference_af", "--ne_obs", "--out", out, "--threads", "2"])
    assert same(np.load(out + ".fisher_obs.npy"), g["f_obs"]) and same(np.load(out + ".ne_obs.npy"), g["ne_obs"])
    assert open(out + ".ne_obs.txt").read() == str(g["ne_obs_txt"])          # means of bit-identical columns
    assert open(out + ".ne_ind.txt").read() == str(g["ne_ind_txt"])
    ref_lines = str(g["stdout"]).replace("<TMP>/", "").splitlines()
    got_lines = [l.replace(str(tmp_path) + "/", "") for l in buf.getvalue().splitlines()]
    assert got_lines == ref_lines
