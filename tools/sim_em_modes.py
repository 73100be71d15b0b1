#!/usr/bin/env python3
"""Which operations of the EM term (emMAF_cy.pyx:19-22) need more than float32 for the converged allele
frequencies to stay within 1e-6 of the exact mode?  A NumPy testbed (CPU, no GPU): the exact mode's rounding
sequence (what em_kernels.hip: term_exact computes, itself bit-pinned to the reference) against candidate
evaluations of the same term, iterated to convergence on a synthetic population (tests/synth.py).

    python tools/sim_em_modes.py [m] [n_pop] [variants...]

Reports, per variant, the share of SNPs whose frequency differs after ONE update, and after the full fit the
worst and the 99.9999 % relative deviation of the clamped frequencies and whether the iteration count matches.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth  # noqa: E402

f32, f64 = np.float32, np.float64


def rcp32(x, rng):
    """v_rcp_f32: 1 ulp.  Modelled as the correctly rounded reciprocal moved by -1, 0 or +1 ulp at random."""
    r = (f32(1) / x).astype(f32)
    k = rng.integers(-1, 2, size=r.shape).astype(np.int32)
    return (r.view(np.int32) + k).view(f32)


def two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


class Exact:
    """term_exact: double products rounded to float32, float32 sum, double quotient, float32 accumulator."""
    name = "exact"

    def start(self, f):
        fd = f.astype(f64)
        self.omf, self.fd2, self.fd = 1.0 - fd, 2.0 * fd, fd

    def term(self, g0, g1, tmp):
        g0d, g1d = g0.astype(f64), g1.astype(f64)
        p0 = ((g0d * self.omf) * self.omf).astype(f32)
        p1 = ((g1d * self.fd2) * self.omf).astype(f32)
        p2 = ((((1.0 - g0d) - g1d) * self.fd) * self.fd).astype(f32)
        s = (p0 + p1) + p2
        num = p1.astype(f64) + 2.0 * p2.astype(f64)
        return (tmp.astype(f64) + 0.5 * (num / s.astype(f64))).astype(f32)


class Fast(Exact):
    """term_fast of round 2: everything in float32, one approximate reciprocal."""
    name = "fast_r2"

    def __init__(self, rng):
        self.rng = rng

    def start(self, f):
        self.omff, self.ff2, self.ff = f32(1) - f, f32(2) * f, f

    def term(self, g0, g1, tmp):
        p0 = g0 * self.omff * self.omff
        p1 = g1 * self.ff2 * self.omff
        p2 = ((f32(1) - g0) - g1) * self.ff * self.ff
        s = (p0 + p1) + p2
        num = p1 + f32(2) * p2
        return tmp + num * rcp32(f32(2) * s, self.rng)


class Mixed(Exact):
    """Configurable: where each group of operations is evaluated.
       prod: 'f32' (float32 factors hoisted per SNP), 'hoist64' (double factors A=(1-f)^2, B=2f(1-f), C=f^2 hoisted per
             SNP, one double product + rounding per p), 'exact'
       div:  'f32rcp' (approximate reciprocal), 'f32newton' (residual-corrected float32 quotient of the float32 numerator),
             'f32comp' (same with the numerator as an exact float32 pair), 'f64'
       acc:  'f32' (fma in float32), 'f64' (the reference's double add, rounded)"""

    def __init__(self, prod, div, acc, rng):
        self.prod, self.div, self.acc, self.rng = prod, div, acc, rng
        self.name = "%s/%s/%s" % (prod, div, acc)

    def start(self, f):
        Exact.start(self, f)
        self.A, self.B, self.C = self.omf * self.omf, self.fd2 * self.omf, self.fd * self.fd
        self.Af, self.Bf, self.Cf = self.A.astype(f32), self.B.astype(f32), self.C.astype(f32)

    def term(self, g0, g1, tmp):
        if self.prod == "f32":
            p0, p1, p2 = g0 * self.Af, g1 * self.Bf, ((f32(1) - g0) - g1) * self.Cf
        elif self.prod == "hoist64":
            g0d, g1d = g0.astype(f64), g1.astype(f64)
            p0, p1, p2 = (g0d * self.A).astype(f32), (g1d * self.B).astype(f32), (((1.0 - g0d) - g1d) * self.C).astype(f32)
        else:
            g0d, g1d = g0.astype(f64), g1.astype(f64)
            p0 = ((g0d * self.omf) * self.omf).astype(f32)
            p1 = ((g1d * self.fd2) * self.omf).astype(f32)
            p2 = ((((1.0 - g0d) - g1d) * self.fd) * self.fd).astype(f32)
        s = (p0 + p1) + p2
        if self.div == "f64":
            q = (p1.astype(f64) + 2.0 * p2.astype(f64)) / s.astype(f64)
        else:
            with np.errstate(all="ignore"):
                r = rcp32(s, self.rng)
                if self.div == "f32rcp":
                    q = (p1 + f32(2) * p2) * r
                elif self.div == "f32newton":
                    num = p1 + f32(2) * p2
                    q0 = num * r
                    rem = (num.astype(f64) - s.astype(f64) * q0.astype(f64)).astype(f32)       # fma(-s, q0, num): exact
                    q = (q0.astype(f64) + rem.astype(f64) * r.astype(f64)).astype(f32)         # fma(rem, r, q0)
                else:                                                                          # f32comp
                    nh, nl = two_sum(p1, f32(2) * p2)
                    q0 = nh * r
                    rem = (nh.astype(f64) - s.astype(f64) * q0.astype(f64)).astype(f32)
                    q = (q0.astype(f64) + (rem + nl).astype(f64) * r.astype(f64)).astype(f32)
                bad = ~np.isfinite(q) | (s <= 0)
                if bad.any():
                    q = np.where(bad, ((p1.astype(f64) + 2.0 * p2.astype(f64)) / s.astype(f64)).astype(q.dtype), q)
        if self.acc == "f64":
            return (tmp.astype(f64) + 0.5 * q.astype(f64)).astype(f32)
        return (tmp.astype(f64) + 0.5 * q.astype(f32).astype(f64)).astype(f32)                 # float32 fma: one rounding


def update(mode, L, f):
    mode.start(f)
    tmp = np.zeros_like(f)
    for i in range(L.shape[1] // 2):
        tmp = mode.term(L[:, 2 * i], L[:, 2 * i + 1], tmp)
    return tmp / f32(L.shape[1] // 2)


def rmse(a, b):
    d = (a - b).astype(f32)
    acc = np.cumsum((d * d).astype(f32), dtype=f32)[-1]
    return float(np.sqrt(f64(acc / f32(len(a)))))


def fit(mode, L, iters=200, tole=1e-4):
    f = np.full(L.shape[0], 0.25, dtype=f32)
    for it in range(iters):
        fn = update(mode, L, f)
        if rmse(fn, f) < tole:
            return fn, it + 1
        f = fn
    return f, iters


def clamp(f, n):
    lo = f32(1.0 / (2 * (n + 1)))
    return np.minimum(np.maximum(f, lo), f32(1) - lo)


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    want = sys.argv[3:]
    L, _ = synth.make_beagle(m, n, 1, seed=77)
    rng = np.random.default_rng(5)
    modes = [Fast(rng)] + [Mixed(p, d, a, rng) for p, d, a in [
        ("exact", "f64", "f64"), ("hoist64", "f64", "f64"), ("f32", "f64", "f64"), ("hoist64", "f32comp", "f64"),
        ("hoist64", "f32newton", "f64"), ("hoist64", "f32newton", "f32"), ("hoist64", "f32comp", "f32"), ("f32", "f32comp", "f32"),
        ("f32", "f32newton", "f32"), ("hoist64", "f32rcp", "f32"), ("f32", "f32rcp", "f64")]]
    if want:
        modes = [x for x in modes if x.name in want]
    ex = Exact()
    f1 = update(ex, L, np.full(m, 0.25, dtype=f32))
    fe, ie = fit(ex, L)
    ce = clamp(fe, n).astype(f64)
    print("m=%d n=%d exact: %d iterations" % (m, n, ie))
    for md in modes:
        g1 = update(md, L, np.full(m, 0.25, dtype=f32))
        fm, im = fit(md, L)
        d = np.abs(clamp(fm, n).astype(f64) - ce) / ce
        print("%-28s one update: %6.3f %% of SNPs differ | fit: iters %s  max rel %.2e  p99.9999 %.2e  share > 1e-6: %.2e" % (
            md.name, 100 * np.mean(g1 != f1), "same" if im == ie else "%d != %d" % (im, ie), d.max(), np.quantile(d, 0.999999), np.mean(d > 1e-6)))


if __name__ == "__main__":
    main()
