"""CPU-only multi-process tests of the TCP-star communicator (wgsassign_amd/comm.py): the bootstrap
and gather channel of RcclComm and the all-reduce of its SocketComm fallback.  Three ranks issue
collectives back to back (the pattern that raced when every call accepted fresh connections), and the
SNP-sharded EM driver loop runs over it with the oracle standing in for the device."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_host_logic_cpu import free_port

_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from wgsassign_amd.comm import SocketComm, shard_range
rank, world = int(sys.argv[1]), int(sys.argv[2])
comm = SocketComm(rank, world, "127.0.0.1", {port})
ok = True
rng = np.random.default_rng(5)
base = rng.standard_normal((world, 7, 3))
for rep in range(200):                                   # back-to-back collectives of different kinds
    tot = comm.allreduce_sum(base[rank] * (rep + 1))
    ok &= tot.tobytes() == sum(base[r] * (rep + 1) for r in range(world)).tobytes()     # rank-order sum: same bits everywhere
    objs = comm.allgather_object([["site%d" % (rank * 10 + rep)], rep, rank])
    ok &= objs == [[["site%d" % (r * 10 + rep)], rep, r] for r in range(world)]
    rows = comm.gather_rows(np.full((rank + 1, 4), rank + rep, dtype=np.float32))
    if rank == 0:
        ok &= rows.shape == (world * (world + 1) // 2, 4) and rows.dtype == np.float32
        ok &= [int(x) for x in rows[:, 0]] == [r + rep for r in range(world) for _ in range(r + 1)]
    else:
        ok &= rows is None
comm.barrier()
# a big array through gather_rows (raw buffers, no pickling)
big = np.arange((rank + 1) * 300_000, dtype=np.float32).reshape(-1, 5)
rows = comm.gather_rows(big)
if rank == 0:
    ok &= rows.shape[0] == sum((r + 1) * 60_000 for r in range(world))

# the collective leave-one-out batch size: ranks see different free memory -> all use the minimum
from wgsassign_amd import glassy
class Ctx:
    def mem_info(self):
        return (int(1e9) * (3 - rank), int(4e9))
class B:
    ctx, m = Ctx(), 1_000_000
sizes = comm.allgather_object(glassy.loo_batch_size(B(), 5000, comm))
ok &= len(set(sizes)) == 1 and sizes[0] == int(0.8 * 1e9 * (3 - (world - 1))) // (int(1_000_000 * 8.2) + 4096)
os.environ["WGSASSIGN_LOO_BATCH"] = str(40 + rank)       # even a forced size is agreed on
sizes = comm.allgather_object(glassy.loo_batch_size(B(), 5000, comm))
ok &= sizes == [40] * world
del os.environ["WGSASSIGN_LOO_BATCH"]

# the SNP-sharded EM driver loop over this communicator (oracle stand-in for the device)
from oracle import oracle as orc
from cpu_standin import OracleEMBatch
from wgsassign_amd.device import run_em
g = np.load(os.path.join({root!r}, "tests", "golden", "amre_fit.npz"))
L, IDs = g["L"], g["IDs"]
lo, hi = shard_range(L.shape[0], rank, world)
pops = np.unique(IDs[:, 1])
groups = [np.flatnonzero(IDs[:, 1] == p) for p in pops]
em = OracleEMBatch(orc, np.ascontiguousarray(L[lo:hi]), groups, guard=float(sys.argv[3]))
iters = run_em(em, 200, 1e-4, comm=comm, m_total=L.shape[0])
ok &= list(iters) == list(g["iters"])
for k in range(len(pops)):
    ok &= em.f[k].tobytes() == g["f_raw"][k][lo:hi].tobytes()
print("RANK", rank, "OK" if ok else "FAIL", flush=True)
comm.barrier(); comm.close()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("guard", [0.0, 1e9])
def test_three_ranks_over_the_tcp_star(tmp_path, guard):
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3", str(guard)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for r in range(3)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "RANK %d OK" % r in o


def test_side_channel_rejects_strangers(tmp_path):
    """A connection that does not present magic | rank | world | token is dropped; the real peer still joins."""
    import socket
    import threading
    import time
    from wgsassign_amd.comm import SideChannel
    port = free_port()
    res = {}

    def hub():
        res["hub"] = SideChannel(0, 2, "127.0.0.1", port, token="secret", timeout=30)

    t = threading.Thread(target=hub)
    t.start()
    time.sleep(0.3)
    with socket.create_connection(("127.0.0.1", port)) as s:      # a stranger: wrong magic
        s.sendall(b"GET / HTTP/1.0\r\n\r\n" + b"x" * 64)
    with pytest.raises(RuntimeError):                              # right magic, wrong token
        SideChannel(1, 2, "127.0.0.1", port, token="guess", timeout=2)
    peer = SideChannel(1, 2, "127.0.0.1", port, token="secret", timeout=30)
    t.join(30)
    assert not t.is_alive()
    got = {}
    th = threading.Thread(target=lambda: got.setdefault("b", peer.bcast(None)))
    th.start()
    res["hub"].bcast(b"hello")
    th.join(10)
    assert got["b"] == b"hello"
    peer.close()
    res["hub"].close()


def test_unpack_array_refuses_object_dtype():
    from wgsassign_amd import comm
    blob = comm._pack_array(np.arange(6, dtype=np.float32).reshape(2, 3))
    assert comm._unpack_array(blob).tolist() == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
    import json
    import struct
    head = json.dumps({"dtype": "|O", "shape": [1]}).encode()
    with pytest.raises(RuntimeError):
        comm._unpack_array(struct.pack("<i", len(head)) + head + b"\0" * 8)


def test_launch_local_ranks_env_stdout_and_failure(tmp_path, capfd):
    """`WGSassign --gpus N` starts its ranks through comm.launch_local_ranks: every rank sees RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*, only rank 0 reaches stdout, a failing rank ends the others and its status is returned."""
    import sys
    import time
    from wgsassign_amd import comm
    code = ("import os, sys; r = os.environ['RANK']; "
            "open(os.path.join(%r, 'r' + r), 'w').write(' '.join(os.environ[k] for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY'))); print('out of rank', r)" % str(tmp_path))
    assert comm.launch_local_ranks(3, [sys.executable, "-c", code]) == 0
    seen = [open(tmp_path / ("r%d" % r)).read().split() for r in range(3)]
    assert [s[:4] for s in seen] == [[str(r), str(r), "3", "127.0.0.1"] for r in range(3)]
    assert len({s[4] for s in seen}) == 1 and all(s[5] == "0" for s in seen)
    out = capfd.readouterr().out
    assert out.strip().splitlines() == ["out of rank 0"]
    # rank 1 fails at once, the others would run for a minute: ended, status 7 returned promptly
    code = "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(60)"
    t0 = time.time()
    assert comm.launch_local_ranks(3, [sys.executable, "-c", code]) == 7
    assert time.time() - t0 < 20


def test_launch_local_ranks_retries_over_sockets_when_rccl_did_not_come_up(tmp_path, monkeypatch):
    """A rank that leaves with status 75 (RcclComm's watchdog: the RCCL communicator did not initialise) makes
    `WGSassign --gpus N` start all ranks again with WGSASSIGN_COMM=socket; any other failure is final."""
    import sys
    from wgsassign_amd import comm
    monkeypatch.delenv("WGSASSIGN_COMM", raising=False)
    monkeypatch.delenv("WGSASSIGN_BACKEND", raising=False)
    code = ("import os, sys; open(os.path.join(%r, os.environ.get('WGSASSIGN_COMM', 'rccl') + os.environ['RANK']), 'w').close(); "
            "sys.exit(0 if os.environ.get('WGSASSIGN_COMM') == 'socket' else 75)" % str(tmp_path))
    assert comm.launch_local_ranks(2, [sys.executable, "-c", code]) == 0
    assert sorted(os.listdir(tmp_path)) == ["rccl0", "rccl1", "socket0", "socket1"] or {"socket0", "socket1"} <= set(os.listdir(tmp_path))
    monkeypatch.setenv("WGSASSIGN_COMM", "socket")           # an explicit choice is not overridden
    assert comm.launch_local_ranks(2, [sys.executable, "-c", "import sys; sys.exit(75)"]) == 75


def test_launch_local_ranks_tries_rccl_twice_and_sees_every_exit_status(tmp_path, monkeypatch):
    """The attempts of comm.comm_attempts, each in fresh processes: RCCL with HSA_ENABLE_IPC_MODE_LEGACY=0, RCCL with
    the setting flipped, sockets.  A status 75 counts whichever rank reports it and even when a lower rank has just
    left with another status (a peer whose TCP star lost the rank that timed out)."""
    import sys
    from wgsassign_amd import comm
    monkeypatch.delenv("WGSASSIGN_COMM", raising=False)
    monkeypatch.delenv("WGSASSIGN_BACKEND", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    assert [a[1] for a in comm.comm_attempts({})] == [{"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, {"HSA_ENABLE_IPC_MODE_LEGACY": "1"},
                                                      {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "WGSASSIGN_COMM": "socket"}]
    # comes up only with legacy IPC: the second attempt, still RCCL
    code = ("import os, sys; e = os.environ; tag = e.get('WGSASSIGN_COMM', 'rccl') + e['HSA_ENABLE_IPC_MODE_LEGACY'] + '_' + e['RANK']; "
            "open(os.path.join(%r, tag), 'w').close(); sys.exit(0 if e['HSA_ENABLE_IPC_MODE_LEGACY'] == '1' else 75)" % str(tmp_path))
    assert comm.launch_local_ranks(2, [sys.executable, "-c", code]) == 0
    assert sorted(os.listdir(tmp_path)) == ["rccl0_0", "rccl0_1", "rccl1_0", "rccl1_1"]
    # rank 0 dies with status 1 at once, rank 1 reports 75 half a second later: still a communicator failure -> next attempt
    for f in os.listdir(tmp_path):
        os.remove(tmp_path / f)
    code = ("import os, sys, time; e = os.environ; first = e.get('WGSASSIGN_COMM', 'rccl') == 'rccl' and e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'; "
            "open(os.path.join(%r, ('a' if first else 'b') + e['RANK']), 'w').close(); "
            "(sys.exit(1) if e['RANK'] == '0' else (time.sleep(0.5), sys.exit(75))) if first else sys.exit(0)" % str(tmp_path))
    assert comm.launch_local_ranks(2, [sys.executable, "-c", code]) == 0
    assert sorted(os.listdir(tmp_path)) == ["a0", "a1", "b0", "b1"]


_INDEX_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
rank = int(sys.argv[1])
os.environ.update(LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE="3", WGSASSIGN_INDEX_DIR={cache!r}, WGSASSIGN_THREADS="6")
from wgsassign_amd import reader_cy
from wgsassign_amd.comm import SocketComm
comm = SocketComm(rank, 3, "127.0.0.1", {port})
idx, _, sites = reader_cy.ensure_index({path!r}, comm)
assert sites == {sites}, sites
with reader_cy.BeagleStream({path!r}, threads=1, index=idx, first_row=1000 * rank + 7) as st:
    rows, names = next(st.chunks(max_rows=5))
assert names[0] == "ctg%d_%d" % ((1000 * rank + 7) % 7, 1000 * rank + 8), names
comm.barrier()
comm.close()
print("RANK %d OK" % rank)
'''


def test_three_ranks_split_the_bgzf_index_pass(tmp_path):
    """`ensure_index` with several ranks on a node: every rank inflates and summarises the blocks of its third of a BGZF
    file, the first rank chains the parts -- the index is byte for byte the one a single process builds; a plain-gzip file
    takes the single-rank pass."""
    import ctypes
    import subprocess
    import sys
    import numpy as np
    import synth
    from conftest import ROOT
    from test_reader_cpu import _bgzf_write, _text_of
    from wgsassign_amd import _lib, reader_cy
    from wgsassign_amd.comm import free_port_pair
    m, n = 4000, 15
    L, _ = synth.make_beagle(m, n, 2, seed=21)
    text = _text_of(L)
    for kind in ("bgzf", "gzip"):
        p = str(tmp_path / ("x_%s.beagle.gz" % kind))
        if kind == "bgzf":
            _bgzf_write(p, text, block=5000)
        else:
            import gzip
            with gzip.open(p, "wt", newline="") as fh:
                fh.write(text)
        cache = str(tmp_path / ("cache_" + kind))
        os.makedirs(cache, mode=0o700)
        port = free_port_pair()
        script = tmp_path / ("w_%s.py" % kind)
        script.write_text(_INDEX_WORKER.format(root=ROOT, cache=cache, port=port, path=p, sites=m))
        procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for r in range(3)]
        outs = [q.communicate(timeout=300)[0] for q in procs]
        for r, (q, o) in enumerate(zip(procs, outs)):
            assert q.returncode == 0 and "RANK %d OK" % r in o, "rank %d failed:\n%s" % (r, o[-3000:])
        files = sorted(os.listdir(cache))
        assert len(files) == 1 and files[0].endswith(".idx"), files            # no part files left behind
        alone = str(tmp_path / ("alone_%s.idx" % kind))
        sites = ctypes.c_int64()
        _lib.check(_lib.load().wgs_reader_build_index(p.encode(), alone.encode(), None, reader_cy.INDEX_SPAN_BYTES,
                                                      reader_cy.INDEX_MAX_POINTS, ctypes.byref(sites)))
        assert open(os.path.join(cache, files[0]), "rb").read() == open(alone, "rb").read()


_OUT_OF_STEP_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from wgsassign_amd.comm import SocketComm, CollectiveMismatch, COMM_DIVERGED
rank, world, variant = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
comm = SocketComm(rank, world, "127.0.0.1", {port})
x = np.arange(5, dtype=np.float64) + rank
def step_a():
    return comm.allreduce_sum(x)                        # one call site ...
def step_b():
    return comm.allreduce_sum(x)                        # ... and another: a different collective although the payloads look alike
try:
    for it in range(6):
        assert step_a().tobytes() == sum(np.arange(5, dtype=np.float64) + r for r in range(world)).tobytes()
        if variant == "other_call_site" and rank == 1 and it == 3:
            step_b()                                    # rank 1 takes another branch of the host code: same size, other meaning
        elif variant == "other_iteration" and it == 3:
            comm.allreduce_sum(x, tag=(7, it + (rank == 2), 0, 0))      # rank 2 believes it is one iteration further
        elif variant == "other_kind" and rank == 1 and it == 3:
            comm.allgather_object(["names"])            # a gather where the others all-reduce
        elif variant == "other_size" and rank == 2 and it == 3:
            comm.allreduce_sum(np.zeros(9))
        else:
            step_a()
    print("RANK", rank, "went through", flush=True)
    sys.exit(0)
except CollectiveMismatch as e:
    print("RANK", rank, "MISMATCH:", e, flush=True)
    os._exit(COMM_DIVERGED)
'''


@pytest.mark.parametrize("variant", ["other_call_site", "other_iteration", "other_kind", "other_size"])
def test_ranks_out_of_step_stop_with_both_tuples(tmp_path, variant):
    """Three ranks over the TCP star, one of which issues a DIFFERENT collective at its fourth step (another call site with the
    same payload size, another iteration in its tag, another kind of collective, another size): no rank may go on -- every rank
    ends with status 76 and a message that names what two ranks issued.  Before round 5 the first two variants summed the
    payloads of different collectives without a word."""
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(_OUT_OF_STEP_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3", variant], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(3)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert "went through" not in o, "rank %d did not notice:\n%s" % (r, o[-2000:])
        assert p.returncode == 76, "rank %d: status %s\n%s" % (r, p.returncode, o[-2000:])
        assert "collective mismatch" in o
    text = "\n".join(outs)
    if variant == "other_call_site":       # both call sites are named, file and line
        import re
        sites = set(re.findall(r"host all-reduce at (worker\.py|<other file>):(\d+)", text))
        assert len({line for _, line in sites}) == 2, text[-2000:]
    if variant == "other_iteration":
        assert "generation 7, iteration 3" in text and "generation 7, iteration 4" in text
    if variant == "other_size":
        assert "5 payload elements" in text or "float64" in text


def test_tag_rows_carry_a_free_word_per_rank(tmp_path):
    """aux is not compared: ranks report different values in it and every rank sees all of them (how wgs_em_fit learns that
    every rank can run two iterations per sweep)."""
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from wgsassign_amd.comm import SocketComm
rank = int(sys.argv[1])
comm = SocketComm(rank, 3, "127.0.0.1", %d)
out = comm.allreduce_sum(np.ones(2), tag=(1, 0, 0, 10 + rank))
assert out.tolist() == [3.0, 3.0] and comm.last_rows[:, 7].tolist() == [10.0, 11.0, 12.0], comm.last_rows
comm.barrier(); comm.close()
''' % (ROOT, port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(3)]
    for r, p in enumerate(procs):
        o = p.communicate(timeout=60)[0]
        assert p.returncode == 0, "rank %d:\n%s" % (r, o[-2000:])


def test_two_gloo_ranks_out_of_step_stop_with_both_tuples(tmp_path):
    """The same over torch.distributed (gloo, world_size 2): rank 1 issues its fourth all-reduce from another call site; both
    ranks see both rows in the all-gathered table and stop with the two call sites named."""
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=int(sys.argv[1]), world_size=2)
from wgsassign_amd.comm import TorchComm, CollectiveMismatch, COMM_DIVERGED
comm = TorchComm()
x = np.arange(4.0) + comm.rank
try:
    for it in range(6):
        if comm.rank == 1 and it == 3:
            got = comm.allreduce_sum(x)          # another line of the host code
        else:
            got = comm.allreduce_sum(x)
        assert got.tolist() == (2 * np.arange(4.0) + 1).tolist()
    print("went through", flush=True)
except CollectiveMismatch as e:
    print("MISMATCH:", e, flush=True)
    os._exit(COMM_DIVERGED)
''' % (ROOT, port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 76 and "went through" not in o, "rank %d: status %s\n%s" % (r, p.returncode, o[-2000:])
        assert "rank 0 issued collective #4" in o and "rank 1 issued collective #4" in o, o[-2000:]
