"""Beagle reader: drop-in for `reader_cy.readBeagle` (reader_cy.pyx:16-77).

Host-side text parsing (next-tier component): gzip -> tokens -> float32 (g0, g1)
pairs.  Header: every 3rd token after the first three is a sample name; per line token 0 is
the site name, two allele columns are skipped, GL0 and GL1 are kept and GL2 dropped; values
go through double (atof) and are then rounded to float32.
"""
import gzip

import numpy as np


def readBeagle(beagle):
    with gzip.open(beagle, "rb") as fh:
        header = fh.readline().split()
        sample_names = [t.decode() for t in header[3::3]]
        n = len(header[3:]) // 3
        site_names, chunks = [], []
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            site_names.append(tok[0].decode())
            gl = np.array(tok[3:3 + 3 * n], dtype=np.float64).reshape(n, 3)
            chunks.append(gl[:, :2].astype(np.float32).reshape(-1))
    L = np.ascontiguousarray(np.array(chunks, dtype=np.float32).reshape(len(chunks), 2 * n))
    return L, sample_names, site_names
