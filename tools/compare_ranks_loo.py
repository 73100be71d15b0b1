#!/usr/bin/env python3
"""--loo at BASELINE config-4 size (device-generated matrix), SNP-sharded over N ranks that share GPU 0, against the
same call on one shard: log-likelihoods, partition sums and iteration counts must be bit-identical.

    python tools/compare_ranks_loo.py [ranks] [snps] [inds] [pops] [partitions]      (launches its own ranks)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, m, n, K, P):
    import contextlib
    import io
    import numpy as np
    from wgsassign_amd import device, glassy
    from wgsassign_amd.comm import SocketComm, shard_range
    comm = SocketComm(rank, world, "127.0.0.1", port)
    group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
    ctx = device.Context(0)
    comm.attach(ctx)          # wgs_em_fit / wgs_loo run across the ranks (the host-backed communicator)

    def run(lo, hi, c):
        b = device.DeviceBeagle(hi - lo, n, group_of, K, site0=lo, ctx=ctx)
        b.synth(20260313, 2.0)
        em = device.EMBatch(b, np.arange(K, dtype=np.int32))
        em.run(200, 1e-4, comm=c, m_total=m)
        af = np.stack([em.get_f(k) for k in range(K)], axis=1)
        cnt = np.bincount(group_of, minlength=K)
        for k in range(K):
            lo_c = 1.0 / (2 * (cnt[k] + 1))
            col = af[:, k]
            col[col < lo_c] = lo_c
            col[col > 1 - lo_c] = 1 - lo_c
        em.close()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            ll, parts = glassy.loo_device(b, b, af, group_of, 200, 1e-4, P, comm=c, verbose=False)
        dt = time.perf_counter() - t0
        b.close()
        return ll, parts, af, dt
    lo, hi = shard_range(m, rank, world)
    ll, parts, af, dt = run(lo, hi, comm)
    ok = True
    if rank == 0:
        ll1, parts1, af1, dt1 = run(0, m, None)
        ok = ll.tobytes() == ll1.tobytes() and parts.tobytes() == parts1.tobytes() and af.tobytes() == af1[lo:hi].tobytes()
        print("ranks %d: --loo %d x %d, K=%d, P=%d: sharded %.2f s, one shard %.2f s, bit-identical: %s" % (world, m, n, K, P, dt, dt1, ok), flush=True)
    comm.barrier()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(*[int(x) for x in sys.argv[2:9]])
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    shape = [int(x) for x in sys.argv[2:6]] if len(sys.argv) >= 6 else [2_000_000, 500, 8]
    P = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    from wgsassign_amd.comm import free_port_pair
    port = free_port_pair()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), str(port)] +
                              [str(x) for x in shape] + [str(P)]) for r in range(world)]
    sys.exit(max(p.wait() for p in procs))
