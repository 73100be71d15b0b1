"""A CPU stand-in for wgsassign_amd.device.EMBatch, backed by the oracle's C kernels.

TEST INFRASTRUCTURE: lets the `-m "not gpu"` suite drive the product's host-side EM driver loop
(wgsassign_amd.device.run_em: convergence decisions, per-fit freezing, serial-chain carry
hand-off between SNP shards) without a GPU.  It mirrors the primitives of the C ABI:
step() == wgs_em_step, rmse_chain() == wgs_em_rmse_chain, set_active() == wgs_em_set_active.
"""
import ctypes

import numpy as np


class _Shape:
    def __init__(self, m):
        self.m = m


class OracleEMBatch:
    GUARD = 0.0

    def __init__(self, oracle, L, groups_idx, guard=None):
        """L: this shard's rows (m, 2n); groups_idx: list of individual-index arrays, one per fit."""
        self.orc = oracle
        self.n_fits = len(groups_idx)
        self.slabs = [oracle.gather(L, idx, 2) if len(idx) else np.empty((L.shape[0], 0), np.float32) for idx in groups_idx]
        m = L.shape[0]
        self.b = _Shape(m)
        self.f = [np.full(m, 0.25, dtype=np.float32) for _ in groups_idx]
        self.f_prev = [x.copy() for x in self.f]
        self.active = np.ones(self.n_fits, dtype=bool)
        self.chain_calls = 0
        if guard is not None:
            self.GUARD = guard

    def step(self):
        ssq = np.zeros(self.n_fits)
        for j in range(self.n_fits):
            if not self.active[j]:
                continue
            self.f_prev[j] = self.f[j].copy()
            self.orc.emMAF_update(self.slabs[j], self.f[j], 2)
            d = self.f[j] - self.f_prev[j]
            ssq[j] = float(np.sum((d * d).astype(np.float64)))
        return ssq

    def rmse_chain(self, fit, carry_in):
        """emMAF_cy.pyx:30-31 continued from carry_in over this shard (serial float32)."""
        self.chain_calls += 1
        d = self.f[fit] - self.f_prev[fit]
        sq = d * d
        if len(sq) == 0:
            return np.float32(carry_in)
        with np.errstate(all="ignore"):
            acc = np.cumsum(np.concatenate(([np.float32(carry_in)], sq)).astype(np.float32), dtype=np.float32)
        return np.float32(acc[-1])

    def set_active(self, fit, active):
        self.active[fit] = bool(active)
