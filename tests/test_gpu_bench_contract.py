"""bench.py prints exactly one JSON line with the contract's keys (small workload)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--snps", "200000", "--inds", "100", "--pops", "5",
                        "--steps", "3", "--warmup", "1", "--cpu-snps", "20000", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f64" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > cb["value"]
    assert abs(d["value"] - 5 * 200000 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_bench_line_says_what_rccl_saw():
    """WGS_FORCE_DIST=1 sends the one-GPU run down the N > 1 path over a real RCCL communicator of one rank: the line carries
    RCCL's own count of the communicator's ranks (ncclCommCount -- what tells an N-rank RCCL run from a socket fall-back), every
    rank's sweep time and the device time of the collectives; the headline stays the float32 sweep with frac < 1, and the
    class-coded leg reports the fit cold, warm and over the float32 slabs."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--snps", "200000", "--inds", "100", "--pops", "5",
                        "--steps", "3", "--warmup", "1", "--no-cpu", "--no-paths"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, WGS_FORCE_DIST="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["config"]["rccl_ranks_seen"] == 1 and d["config"]["comm_native_rccl"] is True
    assert len(d["extra"]["per_rank_sweep_kernel_ms"]) == 1 and d["extra"]["per_rank_sweep_kernel_ms"][0] > 0
    assert d["extra"]["collectives"]["allreduce_us"] > 0
    assert d["roofline"]["kernel"] == "em_sweep_kernel<exact>" and 0 < d["roofline"]["frac"] < 1
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--snps", "200000", "--inds", "100", "--pops", "5",
                          "--steps", "3", "--warmup", "1", "--no-cpu", "--no-paths"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d1["config"]["rccl_ranks_seen"] is None and "collectives" not in d1["extra"]
    c = d1["extra"]["coded"]
    assert c["identical_frequencies"] and c["fit_cold"]["seconds"] > 0 and c["fit_warm"]["seconds"] > 0 and c["fit_direct"]["seconds"] > 0
    assert d1["extra"]["assign"]["kernel"].startswith("score_sweep_kernel") and d1["extra"]["assign"]["coded"]["identical_sums"]


@pytest.mark.parametrize("comm_env", [{"WGSASSIGN_COMM": "socket"}, {"WGSASSIGN_BACKEND": "gloo"}])
def test_bench_self_launches_two_ranks(comm_env):
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (both on the one
    GPU of the test box: WGSASSIGN_DEVICE=0; the all-reduce over the TCP star, or over torch's gloo), rank 0
    prints the one JSON line; the sharded sums of squares equal the single-process ones."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--snps", "200000", "--inds", "100", "--pops", "5",
            "--steps", "3", "--warmup", "1", "--no-cpu", "--no-assign"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = subprocess.run(base, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.strip()][0])
    two = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(env, WGSASSIGN_DEVICE="0", **comm_env))
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.strip() and "[Gloo]" not in l]
    assert len(lines) == 1, two.stdout[-2000:]
    d2 = json.loads(lines[0])
    # rank 0's shard: the cut at 100000 moved down to a multiple of 8192 sites (comm.shard_range)
    assert d2["n_gpus"] == 2 and d2["steps"] == 3 and d2["config"]["snps_per_gpu"] == 98304 and d2["scaling"] == "strong"
    assert abs(d2["value"] - 5 * 200000 * 3 / (d2["ms_per_step"] * 3e-3)) / d2["value"] < 1e-6
    a, b = np.array(d1["extra"]["ssq_last"]), np.array(d2["extra"]["ssq_last"])
    assert np.all(np.abs(a - b) <= 1e-12 * np.abs(a)), (a, b)      # same sums, shard partials added in rank order


def test_bench_failure_of_a_rank_propagates(tmp_path):
    """A rank that dies takes the launcher's exit status with it (here: an impossible shape)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--snps", "1", "--inds", "10", "--pops", "2",
                        "--steps", "1", "--warmup", "0", "--no-cpu", "--no-assign"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, WGSASSIGN_DEVICE="0", WGSASSIGN_COMM="socket"))
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_under_torchrun(tmp_path):
    """The driver's documented multi-GPU launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` (here N = 2 on the one GPU, all-reduce over the TCP
    star): every rank process supervises one worker, rank 0's JSON line is the only stdout line."""
    from wgsassign_amd.comm import free_port_pair
    port = free_port_pair()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WGSASSIGN_DEVICE="0", WGSASSIGN_COMM="socket")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--snps", "200000", "--inds", "100",
                        "--pops", "5", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-assign"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["snps_per_gpu"] == 98304


def test_rccl_init_failure_falls_back_to_the_tcp_all_reduce():
    """Default communicator (the library's RCCL one) with two ranks on ONE GPU: RCCL refuses the duplicate device, the
    ranks agree over the TCP star to keep the socket all-reduce, the run completes and says so -- the failure path a
    mis-configured multi-GPU node would take (librccl's stdout banner must not reach the JSON consumer)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                 "WGSASSIGN_COMM", "WGSASSIGN_BACKEND")}
    env.update(WGSASSIGN_DEVICE="0", WGS_BENCH_INIT_TIMEOUT="90")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--snps", "200000", "--inds", "100", "--pops", "5",
                        "--steps", "3", "--warmup", "1", "--no-cpu", "--no-assign"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "RCCL init failed" in d["config"]["comm"] and d["config"]["comm"].startswith("socket all-reduce")
