#!/usr/bin/env python3
"""bench.py -- WGSassign hot path on MI355X: EM allele-frequency sweep + assignment log-lik sweep.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (same result)

For N > 1 without a launcher, bench.py starts the N ranks itself (one child process per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches the GPU; rank 0 prints the JSON line.

A "step" is ONE EM update (emMAF_cy.pyx:10-23) of EVERY population over the whole synthetic
Beagle matrix, fused with the convergence sums, plus -- for N > 1 -- the one collective the path
needs (RCCL all-reduce of the K per-population sums).  Workload: BASELINE.json configs[2],
synthetic 10M SNPs x 1000 individuals, K=10 (the configuration the metric's roofline target is
quoted on; 80 GB of genotype likelihoods, fits one 288 GB MI355X).  For N > 1 the SAME 10M SNPs
are sharded by contiguous SNP range over the ranks (scaling = "strong").

metric  = per-population SNP-updates/s (1 SNP-update = one SNP's EM update over the n_call=n/K
          individuals of one population); whole-job value over all ranks.
roofline= the EM sweep kernel: algorithmic bytes (8n + 8K per SNP) / HIP-event kernel time.
cpu_baseline = the oracle's C/OpenMP restatement of the reference path (per-population column
          gather + emMAF_update per population) on a bounded SNP sample, host cores of this box.
The collective runs over RCCL through the library's own communicator (no PyTorch; WGSASSIGN_COMM=torch
uses torch.distributed instead, WGSASSIGN_COMM=socket a TCP all-reduce).  A rank whose RCCL communicator
cannot initialise exits with status 75 and the launcher starts the ranks again over the socket all-reduce.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
SEED = 20260313


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--snps", dest="m", type=int, default=10_000_000, help="total SNPs (sharded over ranks)")
    ap.add_argument("--inds", dest="n", type=int, default=1000, help="individuals")
    ap.add_argument("--pops", dest="K", type=int, default=10, help="populations")
    ap.add_argument("--arith", dest="mode", default=os.environ.get("WGSASSIGN_MODE", "exact"), choices=["exact", "fast"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-assign", action="store_true", help="skip the assignment sweep leg")
    ap.add_argument("--no-paths", action="store_true", help="skip extra.paths (whole-path timings of the other configurations)")
    ap.add_argument("--no-projection", action="store_true", help="skip extra.shard_projection (one-GPU timings of the per-rank shards of N = 2, 4, 8)")
    ap.add_argument("--no-coded", action="store_true", help="skip extra.coded and the coded scoring sweep (profiles of the float32 kernels alone: every launch of a kernel then does the same work)")
    ap.add_argument("--cpu-snps", type=int, default=200_000, help="SNP sample for the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="minimum CPU-baseline EM timing window")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)     # one rank, started by the launcher below
    return ap.parse_args()


COMM_INIT_FAILED = 75          # wgsassign_amd.comm.COMM_INIT_FAILED (the launcher must not import the package)
INIT_WATCHDOG_S = float(os.environ.get("WGS_BENCH_INIT_TIMEOUT", "120"))


def free_port():
    """MASTER_PORT for the ranks: a port p with p + 1 free as well (the communicators' TCP star listens on p + 1; + 7 is the
    retry's)."""
    import socket
    for _ in range(200):
        with socket.socket() as a:
            a.bind(("127.0.0.1", 0))
            p = a.getsockname()[1]
            if p >= 65520:
                continue
            try:
                for q in (p + 1, p + 7, p + 8, p + 14, p + 15):
                    with socket.socket() as b:
                        b.bind(("127.0.0.1", q))
            except OSError:
                continue
            return p
    raise RuntimeError("no free port range")


def launch_ranks(n_ranks, fixed_env):
    """Start worker processes for ranks (rank, local_rank) in `fixed_env` ... and wait: returns the exit
    status.  Only rank 0's stdout (the JSON line) is forwarded.  Children are separate processes started
    with subprocess -- this process never initialises the GPU and never execs."""
    argv = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--worker"] + ["--worker"]
    # Attempts, each with fresh child processes (wgsassign_amd.comm.comm_attempts; restated here because the launcher
    # must not import the package): RCCL with dmabuf IPC between the ranks (HSA_ENABLE_IPC_MODE_LEGACY=0, what this
    # host driver supports), RCCL with that setting flipped, the socket all-reduce.
    first = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("WGSASSIGN_COMM", "rccl") == "rccl" and os.environ.get("WGSASSIGN_BACKEND") != "gloo":
        attempts = [("rccl, HSA_ENABLE_IPC_MODE_LEGACY=" + first, {"HSA_ENABLE_IPC_MODE_LEGACY": first}),
                    ("rccl, HSA_ENABLE_IPC_MODE_LEGACY=" + ("1" if first == "0" else "0"), {"HSA_ENABLE_IPC_MODE_LEGACY": "1" if first == "0" else "0"}),
                    ("socket all-reduce", {"HSA_ENABLE_IPC_MODE_LEGACY": first, "WGSASSIGN_COMM": "socket"})]
    else:
        attempts = [("as configured", {"HSA_ENABLE_IPC_MODE_LEGACY": first})]
    for attempt, (label, over) in enumerate(attempts):
        procs = []
        for env_r in fixed_env:
            env = dict(os.environ, **env_r)
            env.update(over)
            env["WGS_BENCH_COMM_ATTEMPT"] = label
            env["MASTER_PORT"] = str(int(env["MASTER_PORT"]) + 7 * attempt)      # a fresh side-channel port per attempt
            # rank 0's stdout is filtered: only the JSON line goes to this process's stdout (librccl prints a version
            # banner on stdout when its communicator initialises); everything else is passed on to stderr
            out = subprocess.PIPE if env["RANK"] == "0" else subprocess.DEVNULL
            procs.append(subprocess.Popen(argv, env=env, stdout=out, text=True))
        pumps = []
        for p in procs:
            if p.stdout is not None:
                def pump(stream=p.stdout):
                    for line in stream:
                        dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
                        dst.write(line)
                        dst.flush()
                th = threading.Thread(target=pump, daemon=True)
                th.start()
                pumps.append(th)
        # wait for all ranks; once one has failed the others (possibly stuck in a collective with it) get 15 s, then are ended
        failed_at = None
        while any(p.poll() is None for p in procs):
            if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
                failed_at = time.time()
            if failed_at is not None and time.time() - failed_at > 15.0:
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                failed_at = float("inf")
            time.sleep(0.05)
        codes = [p.wait() for p in procs]
        for th in pumps:
            th.join(10)
        if all(c == 0 for c in codes):
            return 0
        if any(c == COMM_INIT_FAILED for c in codes) and attempt + 1 < len(attempts):
            print("bench.py: the communicator did not initialise (exit 75; %s); starting the ranks again: %s" % (label, attempts[attempt + 1][0]),
                  file=sys.stderr, flush=True)
            continue
        return next((c for c in codes if c > 0), 1)       # a rank's own status rather than the -SIGTERM of the ones ended here
    return 1


def launcher(args):
    """Decide how this process takes part: returns None to run the benchmark in-process (one GPU), else
    the exit status after having run worker processes."""
    world_env = os.environ.get("WORLD_SIZE")
    if args.worker:
        return None
    if world_env is None or (int(world_env) == 1 and args.gpus > 1 and "RANK" not in os.environ):
        if args.gpus <= 1:
            return None                          # N = 1: everything in this process, exactly as before
        port = free_port()
        envs = [{"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(port)} for r in range(args.gpus)]
        return launch_ranks(args.gpus, envs)
    if int(world_env) == 1:
        return None
    # one rank of a torchrun job: supervise ONE worker so that a failed RCCL bootstrap can be retried
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
    return launch_ranks(1, [{k: os.environ[k] for k in keys if k in os.environ}])


def main():
    args = parse()
    rc = launcher(args)
    if rc is not None:
        sys.exit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    use_dist = world > 1 or os.environ.get("WGS_FORCE_DIST") == "1"   # WGS_FORCE_DIST: rehearse the collective path on 1 GPU
    kind = os.environ.get("WGSASSIGN_COMM", "torch" if os.environ.get("WGSASSIGN_BACKEND") == "gloo" else "rccl")
    device_index = int(os.environ.get("WGSASSIGN_DEVICE", local_rank))     # (narrowed below to the devices this process can see)
    dist = torch = None
    # a rank that cannot build its communicator in time leaves with status 75: the launcher retries over TCP
    watchdog = threading.Timer(INIT_WATCHDOG_S, lambda: os._exit(COMM_INIT_FAILED))
    watchdog.daemon = True
    if use_dist:
        watchdog.start()
    if use_dist and kind == "torch":
        # torch ships its own HIP runtime: it must initialise BEFORE libwgsassign_hip.so touches the
        # device (the other order leaves torch with "No HIP GPUs are available")
        import torch
        import torch.distributed as dist
        backend = os.environ.get("WGSASSIGN_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(device_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from wgsassign_amd import comm as wcomm
    from wgsassign_amd import device
    from wgsassign_amd._lib import MODE_EXACT, MODE_FAST
    if not (use_dist and kind == "torch"):
        device_index = device.default_device_index()       # LOCAL_RANK modulo the visible devices (WGSASSIGN_DEVICE wins)
    mode = MODE_EXACT if args.mode == "exact" else MODE_FAST
    ctx = device.Context(device_index)
    comm_note = None
    if use_dist and kind == "torch":
        comm = wcomm.TorchComm(device=torch.device("cuda", device_index) if dist.get_backend() == "nccl" else None)
        comm_note = "torch.distributed/" + dist.get_backend()
    elif use_dist:
        addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29400"))
        try:
            comm = wcomm.SocketComm(rank, world, addr, port).attach(ctx) if kind == "socket" else wcomm.RcclComm(ctx, rank, world, addr, port)
        except Exception as e:
            print("bench.py rank %d: communicator failed: %s" % (rank, e), file=sys.stderr, flush=True)
            os._exit(COMM_INIT_FAILED)
        if kind == "rccl" and not comm.native:
            comm_note = "socket all-reduce (RCCL init failed: %s)" % comm.native_error
        else:
            comm_note = "rccl (library communicator, no torch)" if kind == "rccl" else "socket all-reduce"
        if os.environ.get("WGS_BENCH_COMM_ATTEMPT"):
            comm_note += " [launcher attempt: %s]" % os.environ["WGS_BENCH_COMM_ATTEMPT"]
    else:
        comm = wcomm.LocalComm()
    comm.force_device = True
    watchdog.cancel()
    # what the communicator is by RCCL's own account (ncclCommCount / ncclCommUserRank / ncclCommCuDevice through wgs_comm_info;
    # wgs_comm_init refuses a communicator that differs from what was asked for)
    comm_info = comm.info() if use_dist and hasattr(comm, "info") and getattr(comm, "handle", None) is not None else None

    m_total, n, K = args.m, args.n, args.K
    lo, hi = wcomm.shard_range(m_total, rank, world)
    m = hi - lo
    per = n // K
    group_of = np.minimum(np.arange(n) // per, K - 1).astype(np.int32)
    n_call = float(n) / K

    t0 = time.time()
    beagle = device.DeviceBeagle(m, n, group_of, K, site0=lo, ctx=ctx)
    beagle.synth(SEED, 2.0)
    ctx.sync()
    t_gen = time.time() - t0
    gl_bytes = beagle.nbytes()
    # The headline is the sweep over the float32 matrix (8 bytes per (SNP, individual) read once): the class codes
    # (csrc/codes.hip) are switched off for it and measured beside it in extra.coded, their one-time build included.
    user_codes = os.environ.get("WGSASSIGN_CODES")
    os.environ["WGSASSIGN_CODES"] = "0"
    em = device.EMBatch(beagle, np.arange(K, dtype=np.int32), mode=mode)

    def barrier():
        ctx.sync()
        if use_dist:
            comm.barrier()
        ctx.sync()

    def max_over_ranks(x):
        if not use_dist:
            return x
        slots = np.zeros(world)          # max via sums: one slot per rank
        slots[rank] = x
        return float(np.max(comm.allreduce_sum(slots)))

    # A step = one EM iteration of the production loop.  With one shard or RCCL shards that loop is wgs_em_fit
    # (device.EMBatch.fit): per iteration the sweep, the sum reduction, [the RCCL all-reduce of the K sums, enqueued
    # behind it], the device-side convergence decision and the state readback, enqueued one iteration ahead of the
    # host.  tole = 0 never converges, so fit(K, 0.0) runs EXACTLY K iterations.  Other communicators (socket, gloo)
    # take the step-by-step protocol: sweep -> host all-reduce -> one readback per iteration.
    pipelined = not use_dist or getattr(comm, "handle", None) is not None

    def run_steps(e, k):
        if pipelined:
            e.fit(k, 0.0, comm if use_dist else None, m_total)
            return e.fit_stats()[3] / max(1, k)
        ms = []
        for _ in range(k):
            e.step_reduced(comm)
            ms.append(e.last_sweep_ms())
        return float(np.mean(ms)) if ms else 0.0

    run_steps(em, args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = [run_steps(em, args.steps)]
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    ssq = em.step_reduced(comm if use_dist else None)       # untimed: the sums after warmup + steps + 1 updates
    ms_per_step = elapsed / args.steps * 1e3
    value = K * m_total * args.steps / elapsed          # per-population SNP-updates/s, whole job

    # roofline of the dominant kernel (this rank's shard): algorithmic bytes per launch / avg duration
    alg_bytes = (8.0 * n + 8.0 * K) * m
    k_avg = float(np.mean(kernel_ms)) * 1e-3
    achieved = alg_bytes / k_avg
    roofline = {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK, 4), "traffic": None,
                "kernel": "em_sweep_kernel<%s>" % args.mode, "kernel_ms_avg": round(k_avg * 1e3, 4),
                "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_snp": 8 * n + 8 * K,
                "achievable_copy_rate_GBps": 6300.0, "frac_of_achievable": round(achieved / 6.3e12, 4)}
    pmc = committed_pmc(m, n, K, args.mode, False, False)
    roofline["traffic"], roofline["traffic_source"] = pmc.get("em_traffic"), pmc.get("source")
    if pmc.get("reason"):
        roofline["traffic_note"] = pmc["reason"]
    if pmc.get("em_valu_busy_frac") is not None:
        # the exact-mode sweep sits on the FP64 issue roof as well: share of cycles the vector units were busy
        roofline["valu_busy_frac"] = round(pmc["em_valu_busy_frac"], 4)
        roofline["effective_clock_ghz"] = round(pmc["em_clock_ghz"], 3)

    extra = {"gl_pair_terms_per_s": value * n_call, "synth_seconds": round(t_gen, 2),
             "ssq_last": [float(x) for x in np.asarray(ssq)[:3]]}
    if use_dist:
        # every rank's own sweep time (the slowest decides a step), and the device time of the collectives the sharded path uses
        slots = np.zeros(world)
        slots[rank] = k_avg * 1e3
        extra["per_rank_sweep_kernel_ms"] = [round(float(x), 4) for x in comm.allreduce_sum(slots)]
        if comm_info is not None:
            extra["collectives"] = comm.time_collectives(reps=20, n=max(16, K))
    if user_codes is None:
        os.environ.pop("WGSASSIGN_CODES")
    else:
        os.environ["WGSASSIGN_CODES"] = user_codes
    codes_on = args.mode == "exact" and os.environ.get("WGSASSIGN_CODES", "1") != "0" and not args.no_coded
    codes = {"available": False}
    if codes_on and not use_dist:
        extra["coded"], codes = coded_em_leg(ctx, device, beagle, em, K, per, n, m, mode, args)
        if "steady_state_sweep" in extra["coded"]:
            extra["coded"]["steady_state_sweep"]["speedup_over_float32_sweep"] = round(k_avg * 1e3 / extra["coded"]["steady_state_sweep"]["kernel_ms_avg"], 3)

    # the same sweep in WGS_MODE_FAST (float32 term evaluation, ~1e-6 of the reference), for comparison
    if args.mode == "exact":
        em_fast = device.EMBatch(beagle, np.arange(K, dtype=np.int32), mode=MODE_FAST)
        fast_ms = []
        for i in range(4):
            em_fast.step()
            if i:
                fast_ms.append(em_fast.last_sweep_ms())
        em_fast.close()
        fk = float(np.mean(fast_ms)) * 1e-3
        extra["fast_mode_sweep"] = {"kernel_ms_avg": round(fk * 1e3, 4), "hbm_frac": round(alg_bytes / fk / HBM_PEAK, 4),
                                    "note": "float32 EM update (WGSASSIGN_EM_MODE=fast): same iteration counts, frequencies up to "
                                            "7e-6 relative off at this size -- OUTSIDE the 1e-6 bar, opt-in only; the headline "
                                            "value above is exact mode"}

    # assignment log-likelihood sweep (one pass producing all n x K sums), same matrix
    if not args.no_assign:
        afs = device.AFSet(m, K, ctx=ctx)
        for k in range(K):
            em.clamp(k, per)
            afs.set_column_from_em(k, em, k)
        ctx.sync()
        # the float32 matrix first (WGSASSIGN_CODES=0): the sweep SURVEY 8d's algorithmic bytes describe
        os.environ["WGSASSIGN_CODES"] = "0"
        if args.warmup > 0:                       # like the EM leg: one untimed pass first (code objects, workspace, log table)
            device.assign(beagle, afs, mode=mode, comm=comm if use_dist else None)
        barrier()
        t0 = time.perf_counter()
        out, _ = device.assign(beagle, afs, mode=mode, comm=comm if use_dist else None)
        barrier()
        t_as = max_over_ranks(time.perf_counter() - t0)
        as_ms = device.assign.last_ms
        if user_codes is None:
            os.environ.pop("WGSASSIGN_CODES")
        else:
            os.environ["WGSASSIGN_CODES"] = user_codes
        extra["assign"] = {"metric": "assignment log-lik SNPs/s (all n x K terms of a SNP = 1)",
                           "value": m_total / t_as, "unit": "SNPs/s", "seconds": round(t_as, 4),
                           "terms_per_s": m_total * float(n) * K / t_as,
                           "kernel_ms": round(as_ms, 3),
                           "hbm_frac": round((8.0 * n + 4.0 * K) * m / (as_ms * 1e-3) / HBM_PEAK, 4) if as_ms > 0 else None,
                           "kernel": "score_sweep_kernel<%s> + block_prefix_kernel (one launch over all population slabs; float32 matrix)" % args.mode,
                           "bound": "valu_fp64_issue", "checksum": float(np.sum(out))}
        if pmc.get("assign_valu_busy_frac") is not None and pmc.get("assign_insts_valu"):
            # bound: FP64 vector issue (one double log per term), not HBM.  valu_frac = SQ_ACTIVE_INST_VALU * 4 /
            # (1024 SIMDs * GRBM_GUI_ACTIVE / 8) from the committed rocprofv3 PMC pass of this workload
            extra["assign"].update({"valu_frac": round(pmc["assign_valu_busy_frac"], 4),
                                    "valu_insts_per_term": round(pmc["assign_insts_valu"] * 64.0 / (float(m) * n * K), 2),
                                    "traffic": pmc.get("assign_traffic"), "pmc_source": pmc.get("source")})
        if codes_on:
            # through the class codes: the second call is the steady state; a cold call (codes built inside it) is
            # extra.coded.pop_like_cold on a matrix whose codes were dropped
            o1, _ = device.assign(beagle, afs, mode=mode, comm=comm if use_dist else None)
            ctx.sync()
            t0 = time.perf_counter()
            o1, _ = device.assign(beagle, afs, mode=mode, comm=comm if use_dist else None)
            t_warm = time.perf_counter() - t0
            info = beagle.codes_info()
            extra["assign"]["coded"] = {"available": info["available"], "kernel": "score_coded_kernel<%s> (per-class value table in LDS)" % args.mode,
                                        "seconds_warm": round(t_warm, 4), "kernel_ms": round(device.assign.last_ms, 3), "snps_per_s_steady_state": m_total / (device.assign.last_ms * 1e-3) / 1.0 if device.assign.last_ms > 0 else None,
                                        "identical_sums": bool(o1.tobytes() == out.tobytes()), "codes_build_ms_once_per_matrix": round(info["build_ms"], 1)}
        if args.mode == "exact":
            # the float32 scoring sweep (WGSASSIGN_MODE=fast), validated against exact on this very matrix
            os.environ["WGSASSIGN_CODES"] = "0"
            out_f, _ = device.assign(beagle, afs, mode=MODE_FAST, comm=comm if use_dist else None)
            if user_codes is None:
                os.environ.pop("WGSASSIGN_CODES")
            else:
                os.environ["WGSASSIGN_CODES"] = user_codes
            dev_rel = float(np.max(np.abs(out_f - out) / np.abs(out)))
            extra["assign"]["fast_mode"] = {"kernel_ms": round(device.assign.last_ms, 3), "snps_per_s": m_total / (device.assign.last_ms * 1e-3),
                                            "max_rel_dev_of_sums_vs_exact": dev_rel, "within_1e-6": bool(dev_rel < 1e-6)}
        if codes_on and not use_dist and "coded" in extra:
            # --get_pop_like alone on a fresh matrix: the codes are built inside the call
            beagle.synth(SEED, 2.0)
            ctx.sync()
            t0 = time.perf_counter()
            o2, _ = device.assign(beagle, afs, mode=mode)
            dt = time.perf_counter() - t0
            info = beagle.codes_info()
            extra["coded"]["pop_like_cold"] = {"seconds": round(dt, 4), "of_which_codes_build_ms": round(info["build_ms"], 1), "alloc_ms": round(info["alloc_ms"], 1),
                                               "seconds_direct_sweep": round(t_as, 4), "identical_sums": bool(o2.tobytes() == out.tobytes()),
                                               "note": "one call on a matrix without codes: sample pass + allocation + encode pass + coded sweep"}
        afs.close()

    if "assign" in extra:
        # the other half of BASELINE's metric, where the driver's parser keeps it: the scoring sweep over the float32 matrix is bound
        # by FP64 issue (one double log per term), so its roof is stated as the share of the issue slots its instructions fill
        a = extra["assign"]
        issue = m * float(n) * K * INSTS_PER_TERM["score_sweep_kernel<exact>"] / 64.0 / WAVE_ISSUE_PER_S / (a["kernel_ms"] * 1e-3) if a["kernel_ms"] > 0 and args.mode == "exact" else None
        roofline["assign"] = {"bound": "valu_fp64_issue", "kernel": "score_sweep_kernel<%s>" % args.mode, "kernel_ms": a["kernel_ms"],
                              "snps_per_s": a["value"], "hbm_frac": a["hbm_frac"], "valu_frac": a.get("valu_frac"),
                              "fp64_issue_frac": round(issue, 4) if issue else None,
                              "coded_kernel_ms": a.get("coded", {}).get("kernel_ms"), "coded_identical_sums": a.get("coded", {}).get("identical_sums")}

    cpu = None
    if rank == 0 and not args.no_cpu:
        # rank 0's host cores; at N > 1 a shorter window (the other ranks wait at the closing barrier)
        cpu = cpu_baseline(beagle, group_of, K, min(args.cpu_snps, m), args.cpu_seconds if world == 1 else min(args.cpu_seconds, 6.0))
    paths = None
    if rank == 0 and world == 1 and not args.no_paths:
        # the other BASELINE.json configurations and the reference README's LOO shape, whole paths, after the 80 GB
        # of the headline workload have been released
        em.close()
        beagle.close()
        em = beagle = None
        paths = whole_paths(ctx, device, args.mode)
        extra["paths"] = paths
    if rank == 0 and world == 1 and not use_dist and args.mode == "exact" and not args.no_projection:
        if em is not None:
            em.close()
            beagle.close()
            em = beagle = None
        extra["shard_projection"] = shard_projection(ctx, device, wcomm, args, ms_per_step * 1e-3, extra)

    if rank == 0:
        line = {"metric": "EM SNP-updates/s (per-population update, n_call=%g)" % n_call, "value": value,
                "unit": "SNP-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64" if args.mode == "exact" else "f32", "data": "synthetic",
                "config": {"workload": "%d GPU(s), SNP-sharded: synthetic Beagle %d SNPs x %d ind, K=%d, EM sweep (+ pop_like sweep)" % (world, m_total, n, K),
                           "mode": args.mode, "snps_per_gpu": m,
                           "gl_bytes_per_gpu": gl_bytes, "comm": comm_note,
                           "rccl_ranks_seen": comm_info["rccl_ranks_seen"] if comm_info and comm_info["native"] else None,
                           "comm_native_rccl": bool(comm_info and comm_info["native"]),
                           "step": ("wgs_em_fit iteration (enqueued ahead of the host)" if not use_dist or getattr(comm, "native", False)
                                    else "wgs_em_fit iteration, all-reduce staged through the host (TCP)") if pipelined
                           else "sweep + host all-reduce + readback"},
                "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
        print(json.dumps(line), flush=True)
    if em is not None:
        em.close()
        beagle.close()
    if use_dist:
        comm.barrier()
        comm.close()
        if dist is not None:
            dist.destroy_process_group()


def coded_em_leg(ctx, device, beagle, em_direct, K, per, n, m, mode, args):
    """extra.coded: the EM fit through the class codes (csrc/codes.hip: one byte per (SNP, individual) + per-slab dictionaries, built
    by one pass over the matrix; same frequencies bit for bit) with everything it costs: the COLD fit -- --get_reference_af as the
    command line runs it: one fit per matrix, the codes built inside wgs_em_fit when its cost model expects them to pay -- beside
    the same fit over the float32 slabs and the warm fit; the steady-state sweep with ITS algorithmic bytes."""
    def fit(label):
        e = device.EMBatch(beagle, np.arange(K, dtype=np.int32), mode=mode)
        mal0 = device.malloc_seconds()
        t0 = time.perf_counter()
        iters = e.run(200, 1e-4)
        ctx.sync()
        dt = time.perf_counter() - t0
        mal = device.malloc_seconds() - mal0
        late_ms = beagle.codes_wait()                        # (kernel times are not readable while the codes' hipMalloc is in flight)
        st = e.fit_stats()
        r = {"seconds": round(dt, 4), "iterations": [int(x) for x in iters], "sweep_kernels_ms": round(st[3], 2)}
        if mal > 1e-3:                                       # (buffers of the fused sweep, allocated at first use: what the driver took for them)
            r["of_which_hipMalloc_seconds"] = round(mal, 4)
        if late_ms > 0:
            r["codes_memory_arrived_after_the_fit_hipMalloc_ms"] = round(late_ms, 1)
        return e, r

    res = {}
    e_cold, res["fit_cold"] = fit("cold")                    # nothing built yet: the cost model decides inside wgs_em_fit
    built_by_fit = beagle.codes_state() == 1
    res["fit_cold"]["codes_built_inside_the_fit"] = built_by_fit
    if not built_by_fit and "codes_memory_arrived_after_the_fit_hipMalloc_ms" in res["fit_cold"]:
        # the model wanted the codes, their memory was not there within 3 ms -- the hipMalloc (on the library's helper thread) of VRAM
        # an earlier process used takes seconds on this driver -- and the fit ran over the float32 slabs instead of waiting
        res["fit_cold"]["note"] = "the codes' memory was not there in time: the fit did not wait and ran over the float32 slabs; the next fit builds them"
        e2, res["fit_second_builds_the_codes"] = fit("second")
        e2.close()
    # one key for readers whatever the box did: the fit that finds nothing built and the codes' memory at hand -- the cold fit itself, or
    # (where the driver took seconds to hand that memory out) the fit after it
    res["fit_building_the_codes_seconds"] = res.get("fit_second_builds_the_codes", res["fit_cold"])["seconds"] if (built_by_fit or "fit_second_builds_the_codes" in res) else None
    e_warm, res["fit_warm"] = fit("warm")
    info = beagle.codes_info() if beagle.codes_state() == 1 else None
    os.environ["WGSASSIGN_CODES"] = "0"
    e_dir, res["fit_direct"] = fit("direct")
    os.environ.pop("WGSASSIGN_CODES")
    same = all(e_cold.get_f(k).tobytes() == e_dir.get_f(k).tobytes() for k in (0, K - 1))
    res["identical_frequencies"] = bool(same and res["fit_cold"]["iterations"] == res["fit_direct"]["iterations"])
    for e in (e_cold, e_warm, e_dir):
        e.close()
    md = beagle.codes_model(0)                   # the cost model's own numbers beside the measurements (as in extra.paths)
    its = float(max(res["fit_direct"]["iterations"])) or 14.0
    f32_ms = res["fit_direct"]["seconds"] * 1e3
    pred_warm = its * md["em_share_saved_by_a_coded_sweep"] * md["em_float32_sweep_ms"] if md["builds_for_a_fit"] else 0.0
    pred_cold = pred_warm - md["encode_ms"] if md["builds_for_a_fit"] else 0.0
    meas_cold = f32_ms - (res["fit_building_the_codes_seconds"] or res["fit_cold"]["seconds"]) * 1e3
    meas_warm = f32_ms - res["fit_warm"]["seconds"] * 1e3
    res["cost_model"] = {"builds_the_codes": md["builds_for_a_fit"], "classes_per_slab_and_snp": round(md["sample_classes_per_slab_and_snp"], 2),
                         "float32_fit_ms": {"predicted": round(its * md["em_float32_sweep_ms"], 3), "measured": round(f32_ms, 3)},
                         "encode_ms": {"predicted": round(md["encode_ms"], 3), "measured": round(info["build_ms"], 2) if info else None},
                         "saving_cold_ms": {"predicted": round(pred_cold, 3), "measured": round(meas_cold, 3)},
                         "saving_warm_ms": {"predicted": round(pred_warm, 3), "measured": round(meas_warm, 3)},
                         "abs_error_share_of_float32_fit": {"cold": round(abs(pred_cold - meas_cold) / f32_ms, 4), "warm": round(abs(pred_warm - meas_warm) / f32_ms, 4)}}
    if info is None:
        res["note"] = "the cost model (csrc/api.hip: em_codes_pay) kept the float32 slabs for this fit"
        return res, {"available": False}
    if built_by_fit:
        res["fit_cold"]["of_which_codes_build_ms"] = round(info["build_ms"], 1)
        # (the hipMalloc of the codes' memory runs on a helper thread -- 0.3 ms ... seconds for these 42 GB, by what earlier processes
        # left in VRAM -- and the fit waits 3 ms for it at most)
        res["fit_cold"]["waited_for_the_codes_memory_ms"] = round(info["alloc_wait_ms"], 2)
    # steady state: exactly `steps` coded sweeps
    e = device.EMBatch(beagle, np.arange(K, dtype=np.int32), mode=mode)
    e.fit(max(1, args.warmup), 0.0)
    e.fit(args.steps, 0.0)
    ck = e.fit_stats()[3] / args.steps * 1e-3
    e.close()
    its = float(np.mean(res["fit_cold"]["iterations"])) or 15.0
    # what the coded sweep must read and write per SNP: a code byte per individual, the (g0, g1) of every class present in every
    # slab (8 bytes each; the sample pass's mean), f in and out -- its own algorithmic bytes, not the float32 matrix's
    alg = (float(n) + 8.0 * K * info["sample_mean_classes_per_slab"] + 8.0 * K) * m
    res["steady_state_sweep"] = {"kernel": "em_coded_kernel", "kernel_ms_avg": round(ck * 1e3, 4), "snp_updates_per_s": K * float(m) / ck,
                                 "algorithmic_bytes_per_launch": alg, "bytes_per_snp": alg / m, "hbm_frac_of_its_own_bytes": round(alg / ck / HBM_PEAK, 4),
                                 "speedup_over_float32_sweep": None}
    res["amortised_ms_per_iteration"] = {"iterations": its, "coded": round((info["build_ms"] + its * ck * 1e3) / its, 3),
                                         "note": "(codes build + iterations x coded sweep) / iterations of the one fit --get_reference_af performs"}
    res["class_codes"] = {"bytes": info["bytes"], "build_ms": round(info["build_ms"], 1), "encode_kernel_ms": round(info["encode_kernel_ms"], 1),
                          "sample_ms": round(info["sample_ms"], 2), "alloc_ms": round(info["alloc_ms"], 2),
                          "encode_hbm_frac": round(8.0 * n * m / (info["encode_kernel_ms"] * 1e-3) / HBM_PEAK, 4) if info["encode_kernel_ms"] > 0 else None,
                          "mean_classes_per_snp": round(info["mean_classes"], 2), "max_classes_per_snp": info["max_classes"],
                          "mean_classes_per_slab_and_snp": round(info["sample_mean_classes_per_slab"], 2), "hash_slots_per_snp": info["hash_slots"],
                          "uncoded_snp_share": info["rich_snp_share"], "em_table_rows": info["em_table_rows"],
                          "em_direct_tile_share": round(info["em_direct_tile_share"], 5), "slab_numbering_bytes": info["slab_numbering_bytes"],
                          "probe_rounds_per_16_lookups": round(info["probe_rounds_per_buffer"], 2)}
    return res, info


def shard_projection(ctx, device, wcomm, args, full_step_s, extra):
    """extra.shard_projection (N = 1 runs only): no 8-GPU node was available to any round of this build, so the per-rank work of the
    SNP-sharded path is timed here, on ONE MI355X, at the shard sizes of N = 2, 4, 8 -- the LAST rank's SNP range of the same 10M-SNP
    matrix, so `site0 != 0` -- for every leg the path has grown since the float32 sweep: the float32 EM step, the --get_reference_af
    fit cold (class codes built inside it) and warm, --get_pop_like cold and warm, and the --loo batch of BASELINE configs[3].  A
    rank's wall time at N GPUs is predicted as its shard's time + the collectives the leg issues (counted by the same rules
    tests/test_gpu_multirank.py asserts through wgs_comm_stats) x the measured device time of one tagged collective on a ONE-RANK RCCL
    communicator -- launch, RCCL kernel, tag-row kernels; the xGMI hop itself is NOT in that figure, so `headroom_us_per_collective`
    says how slow a collective may be before the leg falls below 6x at 8 GPUs.  speedup = seconds on the whole matrix (this same run) /
    predicted seconds at N."""
    from wgsassign_amd import glassy
    m_total, n, K = args.m, args.n, args.K
    out = {"note": "one MI355X at the per-rank shard sizes of N = 2, 4, 8 (last rank's SNP range); predicted = shard seconds + collectives x "
                   "one-rank RCCL collective time; no N > 1 run over RCCL/xGMI has executed anywhere yet"}
    coll = {"allreduce_us": None, "bcast_us": None}
    try:
        # (librccl prints a version banner on STDOUT when its first communicator initialises: this process's stdout carries the one JSON
        # line and nothing else, so the descriptor points at stderr meanwhile)
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            c1 = wcomm.RcclComm(ctx, 0, 1)
            coll = c1.time_collectives(reps=50, n=2 * K)
            c1.close()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
        coll["what"] = "wgs_comm_time_collectives on a one-rank RCCL communicator: tag rows + ncclAllReduce / ncclBroadcast + check kernel, on the stream"
    except Exception as e:                       # no librccl: the projection still shows the shards' own times
        coll["error"] = str(e)
    ar, bc = (coll.get("allreduce_us") or 0.0) * 1e-6, (coll.get("bcast_us") or 0.0) * 1e-6
    if (coll.get("bcast_us") or 0.0) < 1.0:      # (a one-rank broadcast returns at once: take the all-reduce's time for a hop)
        bc = ar
        coll["bcast_us_used"] = coll.get("allreduce_us")
    out["collectives"] = coll

    def codes_env(v):
        old = os.environ.get("WGSASSIGN_CODES")
        if v is None:
            os.environ.pop("WGSASSIGN_CODES", None)
        else:
            os.environ["WGSASSIGN_CODES"] = v
        return old

    def leg(full_s, shard_s, n_ar, n_bc, N):
        if not full_s or not shard_s:
            return None
        pred = shard_s + n_ar * ar + n_bc * bc
        r = {"shard_seconds": round(shard_s, 5), "allreduces": int(n_ar), "broadcasts": int(n_bc), "predicted_seconds": round(pred, 5),
             "speedup": round(full_s / pred, 2), "efficiency": round(full_s / pred / N, 3)}
        if N == 8:
            n_coll = n_ar + n_bc
            r["holds_6x"] = bool(full_s / pred >= 6.0)
            r["headroom_us_per_collective"] = round((full_s / 6.0 - shard_s) / n_coll * 1e6, 1) if n_coll else None
        return r

    coded = extra.get("coded", {})
    full = {"float32_em_step": full_step_s,
            "fit_cold": coded.get("fit_building_the_codes_seconds") or coded.get("fit_cold", {}).get("seconds"),
            "fit_warm": coded.get("fit_warm", {}).get("seconds"),
            "pop_like_cold": coded.get("pop_like_cold", {}).get("seconds"),
            "pop_like_warm": extra.get("assign", {}).get("coded", {}).get("seconds_warm")}
    out["seconds_on_the_whole_matrix"] = {k: (round(v, 5) if v else None) for k, v in full.items()}
    per = n // K
    group_of = np.minimum(np.arange(n) // per, K - 1).astype(np.int32)
    for N in (2, 4, 8):
        lo, hi = wcomm.shard_range(m_total, N - 1, N)
        ms = hi - lo
        b = device.DeviceBeagle(ms, n, group_of, K, site0=lo, ctx=ctx)
        b.synth(SEED, 2.0)
        ctx.sync()
        res = {"snps": ms, "site0": lo}
        # the float32 EM step (the headline's kernel)
        old = codes_env("0")
        e = device.EMBatch(b, np.arange(K, dtype=np.int32))
        e.fit(max(1, args.warmup), 0.0)
        ctx.sync()
        t0 = time.perf_counter()
        e.fit(args.steps, 0.0)
        ctx.sync()
        step_s = (time.perf_counter() - t0) / args.steps
        res["float32_em_step"] = leg(full["float32_em_step"], step_s, 1, 0, N)
        res["float32_em_step"]["sweep_kernel_ms"] = round(e.fit_stats()[3] / args.steps, 4)
        e.close()
        codes_env(old)
        # --get_reference_af: the cold fit (nothing built), then the warm one
        for label in ("fit_cold", "fit_warm"):
            e = device.EMBatch(b, np.arange(K, dtype=np.int32))
            t0 = time.perf_counter()
            iters = e.fit(200, 1e-4)         # (the metric over the shard's own SNPs: the iteration count of the whole matrix, 14)
            ctx.sync()
            dt = time.perf_counter() - t0
            b.codes_wait()
            st = e.fit_stats()
            res[label] = leg(full[label], dt, st[0] + 1, N * st[1], N)
            if res[label]:
                res[label].update({"iterations": int(max(iters)), "sweeps_enqueued": int(st[0]), "codes": b.codes_state() == 1})
            if label == "fit_warm":
                afs = device.AFSet(ms, K, ctx=ctx)
                for k in range(K):
                    e.clamp(k, per)
                    afs.set_column_from_em(k, e, k)
            e.close()
        # --get_pop_like: warm (codes there), then cold on the same matrix generated again
        device.assign(b, afs)
        ctx.sync()
        t0 = time.perf_counter()
        device.assign(b, afs)
        res["pop_like_warm"] = leg(full["pop_like_warm"], time.perf_counter() - t0, 0, N, N)
        b.synth(SEED, 2.0)
        ctx.sync()
        t0 = time.perf_counter()
        device.assign(b, afs)
        res["pop_like_cold"] = leg(full["pop_like_cold"], time.perf_counter() - t0, 0, N, N)
        b.codes_wait()
        afs.close()
        b.close()
        out["N=%d" % N] = res
    # BASELINE configs[3]: --loo --partition_sites 3 at 2M x 500, K=8; whole matrix from extra.paths when it ran
    loo_full = extra.get("paths", {}).get("config4_2Mx500_K8", {}).get("loo_partition_sites_3", {}).get("seconds")
    if loo_full:
        m4, n4, K4 = 2_000_000, 500, 8
        g4 = np.minimum(np.arange(n4) // (n4 // K4), K4 - 1).astype(np.int32)
        out["seconds_on_the_whole_matrix"]["config4_loo"] = loo_full
        for N in (2, 4, 8):
            lo, hi = wcomm.shard_range(m4, N - 1, N)
            b = device.DeviceBeagle(hi - lo, n4, g4, K4, site0=lo, ctx=ctx)
            b.synth(SEED, 2.0)
            e = device.EMBatch(b, np.arange(K4, dtype=np.int32))
            e.fit(200, 1e-4)
            af = np.empty((hi - lo, K4), dtype=np.float32)
            cnt = np.bincount(g4, minlength=K4)
            for k in range(K4):
                e.clamp(k, int(cnt[k]))
                af[:, k] = e.get_f(k)
            e.close()
            tm = {}
            t0 = time.perf_counter()
            glassy.loo_device(b, b, af, g4, 200, 1e-4, 3, verbose=False, timings=tm, need_parts=True)
            dt = time.perf_counter() - t0
            # per batch: the batch-size agreement, one all-reduce per sweep enqueued, the one that closes the fit; `world` broadcasts per
            # chain resolution, for the totals and for the partition chains
            r = leg(loo_full, dt, 2 * tm.get("em_batches", 1) + tm.get("em_iterations_enqueued", 0), N * (2 * tm.get("em_batches", 1) + tm.get("em_chain_resolutions", 0)), N)
            r["snps"] = hi - lo
            out["N=%d" % N]["config4_loo"] = r
            b.close()
    verdict = {}
    for k, v in out.get("N=8", {}).items():
        if isinstance(v, dict) and "holds_6x" in v:
            verdict[k] = {"speedup_at_8": v["speedup"], "holds_6x": v["holds_6x"]}
    out["at_8_gpus"] = verdict
    return out


FP64_ISSUE_CLOCK_GHZ = 2.4      # MI355X peak engine clock: issue fractions below are lower bounds of the busy share
WAVE_ISSUE_PER_S = 1024 * FP64_ISSUE_CLOCK_GHZ * 1e9 / 4.0       # 1024 SIMDs, 4 cycles per wave-wide FP64-rate instruction
# VALU wave-instructions per (SNP, individual[, population]) term of the FP64-issue-bound kernels, from the committed PMC
# passes (profiles/r02_e_loo_final: em_sweep_group_kernel 2.49e10 per sweep of 6.15e10 terms; r02_g_final: score sweep;
# r03_b_final: coded score sweep 4.26e9 per 1e11 terms; r04_loo: em_coded_group_kernel 1.32e10 per sweep of 6.15e10 terms at 62
# individuals and 12.7 classes per slab -- a figure of that shape, not a constant of the kernel)
INSTS_PER_TERM = {"em_sweep_group_kernel<exact>": 25.9, "em_coded_group_kernel": 13.7, "score_sweep_kernel<exact>": 41.6, "score_coded_kernel<exact>": 2.7}


def whole_paths(ctx, device, mode_name):
    """extra.paths: every other BASELINE.json configuration and the reference README's --loo shape (README.md:129-131:
    "30 min" at ~5M SNPs x 180 individuals) as WHOLE paths on device-generated data -- all kernel launches, readbacks,
    exact chains and collectives-of-one included, file parsing excluded.  Every fit and every --get_pop_like is timed COLD
    (`seconds_cold`: a matrix nothing has been built for, as the command line meets it -- the class codes, when the cost
    model of csrc/api.hip wants them, are built inside the timed call), WARM (`seconds_warm`: codes present) and over the
    float32 slabs (`seconds_float32`: WGSASSIGN_CODES=0, the round-2 path).  Roofs of the dominant kernel: `hbm` (fraction of
    8 TB/s the sweeps' algorithmic bytes / kernel time reach) or `valu_fp64_issue` (wave-instructions of the committed PMC pass
    x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time): a lower bound of the busy share, the chip clocks lower under FP64 load)."""
    from wgsassign_amd import glassy
    out = {"mode": mode_name, "note": "seconds = wall clock of the whole call(s) on one MI355X, matrix resident in HBM; cold = nothing built for the "
                                      "matrix before the call, warm = class codes present, float32 = WGSASSIGN_CODES=0"}

    class codes_off:
        def __enter__(self):
            self.old = os.environ.get("WGSASSIGN_CODES")
            os.environ["WGSASSIGN_CODES"] = "0"

        def __exit__(self, *exc):
            if self.old is None:
                os.environ.pop("WGSASSIGN_CODES")
            else:
                os.environ["WGSASSIGN_CODES"] = self.old

    def matrix(m, n, K):
        group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
        b = device.DeviceBeagle(m, n, group_of, K, ctx=ctx)
        b.synth(SEED, 2.0)
        ctx.sync()
        return b, group_of, np.bincount(group_of, minlength=K)

    def codes_note(b):
        if b.codes_state() != 1:
            return {"available": False}
        info = b.codes_info()
        return {"available": True, "build_ms": round(info["build_ms"], 2), "encode_kernel_ms": round(info["encode_kernel_ms"], 2), "alloc_ms": round(info["alloc_ms"], 2),
                "bytes": info["bytes"], "mean_classes_per_snp": round(info["mean_classes"], 2), "em_table_rows": info["em_table_rows"],
                "em_direct_tile_share": round(info["em_direct_tile_share"], 5), "uncoded_snp_share": info["rich_snp_share"], "hash_slots_per_snp": info["hash_slots"]}

    def one_fit(b, K):
        mal0 = device.malloc_seconds()
        t0 = time.perf_counter()
        em = device.EMBatch(b, np.arange(K, dtype=np.int32))
        iters = em.run(200, 1e-4)
        ctx.sync()
        dt = time.perf_counter() - t0
        one_fit.malloc_s = device.malloc_seconds() - mal0    # (hipMalloc of VRAM an earlier process used: up to ~100 ms per GB on this driver)
        late_ms = b.codes_wait()                 # (kernel times are not readable while the codes' hipMalloc is in flight on the helper thread)
        return em, dt, iters, em.fit_stats() + (late_ms,)

    def fit(b, K, counts):
        """--get_reference_af: cold, warm, float32"""
        em, dt, iters, st = one_fit(b, K)
        built = b.codes_state() == 1
        alg = float(np.sum([(8.0 * counts[k] + 8.0) * b.m * iters[k] for k in range(K)]))
        mal_cold = one_fit.malloc_s
        res = {"seconds_cold": round(dt, 4), "iterations": [int(x) for x in iters], "exact_chain_batches": int(st[1]),
               "codes_built_inside_the_cold_fit": built, "cold_sweep_kernels_ms": round(st[3], 3), "class_codes": codes_note(b)}
        if st[-1] > 0 and not built:             # wanted, but the memory came too late for this fit (it did not wait): the next fit builds them
            res["codes_memory_arrived_after_the_cold_fit_hipMalloc_ms"] = round(st[-1], 1)
            em_b, dt_b, it_b, st_b = one_fit(b, K)
            res["seconds_second_fit_building_the_codes"] = round(dt_b, 4)
            em_b.close()
        if built or "seconds_second_fit_building_the_codes" in res:      # (one key whatever the box did: nothing built, the codes' memory at hand)
            res["seconds_fit_building_the_codes"] = res.get("seconds_second_fit_building_the_codes", res["seconds_cold"])
        em2, dt2, it2, st2 = one_fit(b, K)
        res["seconds_warm"] = round(dt2, 4)
        slow = {"cold": mal_cold, "warm": one_fit.malloc_s}
        res["warm_sweep_kernel"] = "em_coded_kernel" if b.codes_state() == 1 else "em_sweep_kernel<exact>"
        res["warm_sweep_kernels_ms"] = round(st2[3], 3)
        em2.close()
        with codes_off():
            em3, dt3, it3, st3 = one_fit(b, K)
        res["seconds_float32"] = round(dt3, 4)
        res["float32_sweep_kernels_ms"] = round(st3[3], 3)
        slow["float32"] = one_fit.malloc_s
        if max(slow.values()) > 2e-3:            # (each figure above includes the allocation of its fit's buffers)
            res["of_which_hipMalloc_seconds"] = {k: round(v, 4) for k, v in slow.items()}
        res["bound"] = "hbm"
        res["hbm_frac_of_float32_sweeps"] = round(alg / (st3[3] * 1e-3) / HBM_PEAK, 4) if st3[3] > 0 else None
        res["identical_frequencies"] = bool(list(it3) == list(iters) and em3.get_f(0).tobytes() == em.get_f(0).tobytes())
        res["snp_updates_per_s_cold"] = float(b.m) * float(np.sum(iters)) / dt
        em3.close()
        forced = None
        if b.codes_state() != 1:
            # the model kept the float32 slabs: what the codes WOULD have given, so that a "no" is accountable too (the model switched
            # to "always" for two fits on the same matrix generated again, which is then generated once more: nothing built)
            b.synth(SEED, 2.0)
            ctx.sync()
            old_sw = os.environ.get("WGSASSIGN_EM_CODES_SWEEPS")
            os.environ["WGSASSIGN_EM_CODES_SWEEPS"] = "0"
            try:
                emf, dtf, itf, stf = one_fit(b, K)
                emf.close()
                if b.codes_state() == 1 and b.codes_info()["em_table_rows"] > 0:
                    emw, dtw, itw, stw = one_fit(b, K)
                    forced = {"seconds_cold": round(dtf, 4), "seconds_warm": round(dtw, 4), "build_ms": round(b.codes_info()["build_ms"], 2),
                              "identical_frequencies": bool(list(itw) == list(iters) and emw.get_f(0).tobytes() == em.get_f(0).tobytes())}
                    emw.close()
            finally:
                if old_sw is None:
                    os.environ.pop("WGSASSIGN_EM_CODES_SWEEPS")
                else:
                    os.environ["WGSASSIGN_EM_CODES_SWEEPS"] = old_sw
            b.synth(SEED, 2.0)
            ctx.sync()
        # the cost model's own numbers (csrc/em_api.hip: em_codes_model -- what em_codes_pay decided with) beside what was measured:
        # the fit over the float32 slabs, and what the class codes saved of it cold (their build inside the fit) and warm
        md = b.codes_model(0)
        its = float(max(iters)) if max(iters) > 0 else 14.0
        f32_ms = dt3 * 1e3
        pred_warm = its * md["em_share_saved_by_a_coded_sweep"] * md["em_float32_sweep_ms"] if md["builds_for_a_fit"] else 0.0
        pred_cold = pred_warm - md["encode_ms"] if md["builds_for_a_fit"] else 0.0
        meas_cold, meas_warm = (dt3 - res.get("seconds_fit_building_the_codes", dt)) * 1e3, (dt3 - dt2) * 1e3
        res["cost_model"] = {"builds_the_codes": md["builds_for_a_fit"], "from_the_sample_pass": md["from_the_sample_pass"],
                             "classes_per_slab_and_snp": round(md["sample_classes_per_slab_and_snp"], 2),
                             "float32_fit_ms": {"predicted": round(its * md["em_float32_sweep_ms"], 3), "measured": round(f32_ms, 3)},
                             "encode_ms": {"predicted": round(md["encode_ms"], 3), "measured": res["class_codes"].get("build_ms")},
                             "saving_cold_ms": {"predicted": round(pred_cold, 3), "measured": round(meas_cold, 3)},
                             "saving_warm_ms": {"predicted": round(pred_warm, 3), "measured": round(meas_warm, 3)},
                             "abs_error_share_of_float32_fit": {"cold": round(abs(pred_cold - meas_cold) / f32_ms, 4), "warm": round(abs(pred_warm - meas_warm) / f32_ms, 4)}}
        if forced is not None:
            # the decision "no" beside what "yes" would have been: predicted (the same model, its share and encode estimate) and measured
            p_warm = its * md["em_share_saved_by_a_coded_sweep"] * md["em_float32_sweep_ms"]
            m_cold, m_warm = (dt3 - forced["seconds_cold"]) * 1e3, (dt3 - forced["seconds_warm"]) * 1e3
            res["cost_model"]["had_it_built"] = {"measured": forced, "saving_cold_ms": {"predicted": round(p_warm - md["encode_ms"], 3), "measured": round(m_cold, 3)},
                                                 "saving_warm_ms": {"predicted": round(p_warm, 3), "measured": round(m_warm, 3)},
                                                 "the_no_was_right": bool(m_cold <= 0.02 * f32_ms),
                                                 "abs_error_share_of_float32_fit": {"cold": round(abs(p_warm - md["encode_ms"] - m_cold) / f32_ms, 4), "warm": round(abs(p_warm - m_warm) / f32_ms, 4)}}
        return em, res

    def pop_like(b, em, K, counts):
        """--get_pop_like after the fit (the codes are there if the fit built them), over the float32 slabs, and alone on a
        fresh matrix (cold)"""
        afs = device.AFSet(b.m, K, ctx=ctx)
        for k in range(K):
            em.clamp(k, int(counts[k]))
            afs.set_column_from_em(k, em, k)
        ctx.sync()
        had = b.codes_state() == 1
        t0 = time.perf_counter()
        o, _ = device.assign(b, afs)
        dt = time.perf_counter() - t0
        late_ms = b.codes_wait()
        terms = float(b.m) * b.n * K
        if late_ms > 0 and b.codes_state() != 1:
            device.assign(b, afs)                # (the memory came too late for that call, which did not wait: this one builds the codes)
        coded = b.codes_state() == 1
        kern = "score_coded_kernel<exact>" if coded else "score_sweep_kernel<exact>"
        res = {"seconds_after_the_fit": round(dt, 4), "codes_present_before_the_call": had, "kernel": kern}
        if late_ms > 0:
            res["codes_memory_arrived_after_the_call_hipMalloc_ms"] = round(late_ms, 1)
        o, _ = device.assign(b, afs)
        ms = device.assign.last_ms
        res.update({"kernel_ms_warm": round(ms, 3), "snps_per_s_warm": b.m / (ms * 1e-3), "bound": "valu_issue (+ LDS table reads)" if coded else "valu_fp64_issue",
                    "fp64_issue_frac": round(terms * INSTS_PER_TERM[kern] / 64.0 / WAVE_ISSUE_PER_S / (ms * 1e-3), 4), "checksum": float(np.sum(o))})
        with codes_off():
            t0 = time.perf_counter()
            od, _ = device.assign(b, afs)
            res["seconds_float32"] = round(time.perf_counter() - t0, 4)
            res["float32_kernel_ms"] = round(device.assign.last_ms, 3)
            res["float32_hbm_frac"] = round((8.0 * b.n + 4.0 * K) * b.m / (device.assign.last_ms * 1e-3) / HBM_PEAK, 4)
        b.synth(SEED, 2.0)                       # the same matrix again, nothing built for it
        ctx.sync()
        t0 = time.perf_counter()
        oc, _ = device.assign(b, afs)
        res["seconds_cold"] = round(time.perf_counter() - t0, 4)
        b.codes_wait()
        res["class_codes_cold"] = codes_note(b)
        res["identical_sums"] = bool(od.tobytes() == o.tobytes() == oc.tobytes())
        md = b.codes_model(K)                    # csrc/codes.hip: wgs_codes_scoring_model, what wgs_codes_pay_for_scoring decided with
        f32_ms = res["seconds_float32"] * 1e3
        pred_warm = md["score_float32_sweep_ms"] * (1.0 - md["score_share_of_the_coded_sweep"]) if md["builds_for_scoring"] else 0.0
        pred_cold = pred_warm - md["encode_for_scoring_ms"] if md["builds_for_scoring"] else 0.0
        meas_cold, meas_warm = f32_ms - res["seconds_cold"] * 1e3, res["float32_kernel_ms"] - res["kernel_ms_warm"]
        res["cost_model"] = {"builds_the_codes": md["builds_for_scoring"], "classes_per_snp": round(md["sample_classes_per_snp"], 2),
                             "float32_sweep_ms": {"predicted": round(md["score_float32_sweep_ms"], 3), "measured": res["float32_kernel_ms"]},
                             "encode_ms": {"predicted": round(md["encode_for_scoring_ms"], 3), "measured": res["class_codes_cold"].get("build_ms")},
                             "saving_cold_ms": {"predicted": round(pred_cold, 3), "measured": round(meas_cold, 3)},
                             "saving_warm_ms": {"predicted": round(pred_warm, 3), "measured": round(meas_warm, 3)},
                             "abs_error_share_of_float32_sweep": {"cold": round(abs(pred_cold - meas_cold) / f32_ms, 4), "warm": round(abs(pred_warm - meas_warm) / f32_ms, 4)}}
        af = afs.to_host()
        afs.close()
        return af, res

    def loo(b, group_of, counts, af, P):
        tm = {}
        mal0 = device.malloc_seconds()
        t0 = time.perf_counter()
        ll, parts = glassy.loo_device(b, b, af, group_of, 200, 1e-4, P, verbose=False, timings=tm, need_parts=P > 1)
        dt = time.perf_counter() - t0
        mal = device.malloc_seconds() - mal0
        # (the shared columns of the scoring step may have built codes without the slabs' own numbering: only codes WITH it serve the re-fits)
        coded_refits = b.codes_state() == 1 and b.codes_info()["em_table_rows"] > 0
        it = tm["iters"]
        terms = float(np.sum([float(it[i]) * (counts[group_of[i]] - 1) for i in range(b.n)])) * b.m
        kms = tm.get("em_sweep_kernel_ms", 0.0)
        return {"seconds": round(dt, 4), "of_which_hipMalloc_seconds": round(mal, 4), "re_fits": int(b.n), "partitions": P, "one_call_wgs_loo": bool(tm.get("one_call")),
                "em_seconds": round(tm.get("em_seconds", 0.0), 4), "scoring_seconds": round(tm.get("score_seconds", 0.0), 4),
                "partition_chain_seconds": round(tm.get("chain_seconds", 0.0), 4), "em_sweep_kernels_ms": round(kms, 2),
                "em_batches": tm.get("em_batches"), "iterations_min_max": [int(it.min()), int(it.max())], "bound": "valu_fp64_issue",
                "fp64_issue_frac_of_em_sweeps": round(terms * INSTS_PER_TERM["em_coded_group_kernel" if coded_refits else "em_sweep_group_kernel<exact>"]
                                                      / 64.0 / WAVE_ISSUE_PER_S / (kms * 1e-3), 4) if kms > 0 else None,
                "em_sweep_kernel": "em_coded_group_kernel (the slab's class table per fit)" if coded_refits else "em_sweep_group_kernel<exact> (float32 slabs)",
                "note": "the re-fits' buffers (8 bytes per SNP and fit) are allocated inside the call: of_which_hipMalloc_seconds is what the driver took for that "
                        "(VRAM an earlier process used is cleared when handed out again); scoring goes through per-individual columns over the float32 slabs",
                "self_assignment_accuracy": float(np.mean(np.argmax(ll, axis=1) == group_of)),
                "checksum": float(np.sum(ll.astype(np.float64))), "partitions_checksum": float(np.sum(parts.astype(np.float64))) if P > 1 else None}

    t_all = time.perf_counter()
    # BASELINE configs[1]: 1M x 200, K=5, --get_reference_af
    b, g, c = matrix(1_000_000, 200, 5)
    em, r = fit(b, 5, c)
    em.close()
    out["config2_1Mx200_K5_get_reference_af"] = r
    b.close()
    # BASELINE configs[3]: 2M x 500, K=8, --get_reference_af --loo --partition_sites 3 (+ --get_pop_like)
    b, g, c = matrix(2_000_000, 500, 8)
    em, r = fit(b, 8, c)
    af, rp = pop_like(b, em, 8, c)
    em.close()
    out["config4_2Mx500_K8"] = {"get_reference_af": r, "get_pop_like": rp, "loo_partition_sites_3": loo(b, g, c, af, 3)}
    b.close()
    # the reference README's timing claim (README.md:129-131): --loo at ~5M SNPs x 180 individuals, "30 min"
    b, g, c = matrix(5_000_000, 180, 5)
    em, r = fit(b, 5, c)
    af, rp = pop_like(b, em, 5, c)
    em.close()
    out["readme_5Mx180_K5"] = {"get_reference_af": r, "loo": loo(b, g, c, af, 1), "reference_readme_claim": "30 min (hardware and threads not stated)"}
    b.close()
    # BASELINE configs[4]: one GPU's shard of 50M x 2000, K=20 on 8 GPUs = 6.25M SNPs (100 GB of genotype likelihoods)
    b, g, c = matrix(6_250_000, 2000, 20)
    em, r = fit(b, 20, c)
    af, rp = pop_like(b, em, 20, c)
    em.close()
    out["config5_shard_6.25Mx2000_K20"] = {"get_reference_af": r, "get_pop_like": rp, "gl_bytes": b.nbytes()}
    b.close()
    # quality-dependent likelihoods (every read its own error rate, four base-quality bins as current instruments write them):
    # 80-100 classes per SNP among 1000 individuals instead of 27 -- what real ANGSD files look like to the class codes
    out["realistic_gl_2Mx1000_K10"] = realistic_gl(ctx, device, fit, pop_like)
    # BASELINE configs[4], the streamed reader: a BGZF Beagle file of n = 2000 individuals (tools/beagle_files.py: simulated
    # 2x data, whole lines per member) from the page cache into a resident matrix -- compressed members H2D, inflate, line
    # listing and tokeniser on the device (csrc/inflate.hip, csrc/ingest.hip); and the same with the host inflating
    out["config5_streamed_ingest_100kx2000"] = ingest_path(ctx, device, 2000, 100_000, 20)
    out["seconds_total"] = round(time.perf_counter() - t_all, 2)
    return out


def realistic_gl(ctx, device, fit, pop_like, m=2_000_000, n=1000, K=10):
    """Class codes on likelihoods computed from per-read base qualities (wgs_beagle_synth_quality; Q in {12, 23, 37} with
    probabilities 3 / 12 / 85 %): the encoder's larger hash tables, fewer SNPs per table of the coded scoring sweep, SNPs and
    tiles beyond the tables taken from the float32 slabs.  --get_reference_af and --get_pop_like cold / warm / float32."""
    group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    b = device.DeviceBeagle(m, n, group_of, K, ctx=ctx)
    b.synth_quality(SEED, 2.0)
    ctx.sync()
    synth_orig = b.synth
    b.synth = lambda seed, depth=2.0: b.synth_quality(seed, depth)      # (pop_like re-generates the matrix for its cold call)
    em, r = fit(b, K, counts)
    af, rp = pop_like(b, em, K, counts)
    em.close()
    b.synth = synth_orig
    info = b.codes_info()
    res = {"generator": "per-read qualities 12/23/37 (3/12/85 %), depth Poisson(2), likelihoods rounded to 6 decimals",
           "get_reference_af": r, "get_pop_like": rp,
           "classes_per_snp_mean_max": [round(info["mean_classes"], 1), info["max_classes"]] if info["available"] else None,
           "classes_per_slab_and_snp_mean": round(info["sample_mean_classes_per_slab"], 1) if info["available"] else None,
           "hash_slots_per_snp": info.get("hash_slots"), "snps_per_scoring_table": info.get("score_batch_snps"),
           "uncoded_snp_share": info.get("rich_snp_share"), "em_table_rows": info.get("em_table_rows"), "em_direct_tile_share": info.get("em_direct_tile_share")}
    b.close()
    # A class-RICH matrix (every read's quality drawn uniformly from eight values between Q20 and Q40: hundreds of classes per SNP among 1000 individuals): the
    # sample pass turns the class codes away and --get_pop_like runs over the float32 slabs.  Both arithmetic modes, and the rule that
    # chooses between them: EXACT unless the user sets WGSASSIGN_MODE=fast.  The float32 mode is inside north_star's 1e-6 on every matrix
    # measured (deviation below), but "inside 1e-6" is a measurement, not a bound the library can prove for the matrix at hand -- a sum of
    # m float32-rounded terms has no a-priori bound tighter than m x 2^-24 -- and bit-identity with the reference is what every parity
    # test of this build pins.  So the default pays the 3x and says so (README); the switch is one environment variable.
    from wgsassign_amd._lib import MODE_EXACT, MODE_FAST
    b = device.DeviceBeagle(m, n, group_of, K, ctx=ctx)
    b.synth_quality(SEED, 2.0, quals=(20, 23, 26, 29, 32, 35, 38, 40), probs=tuple([0.125] * 8))
    ctx.sync()
    A = np.random.default_rng(5).uniform(0.02, 0.98, size=(m, K)).astype(np.float32)
    afs = device.AFSet.from_host(A, ctx=ctx)
    device.assign(b, afs)                        # (sample pass, code objects)
    t0 = time.perf_counter()
    oe, _ = device.assign(b, afs, mode=MODE_EXACT)
    te, ke, state = time.perf_counter() - t0, device.assign.last_ms, b.codes_state()
    device.assign(b, afs, mode=MODE_FAST)
    t0 = time.perf_counter()
    of, _ = device.assign(b, afs, mode=MODE_FAST)
    tf, kf = time.perf_counter() - t0, device.assign.last_ms
    md = b.codes_model(K)
    res["class_rich_uniform_Q20_40"] = {"classes_per_snp_in_the_sample": round(md["sample_classes_per_snp"], 1), "codes_state": state,
                                        "get_pop_like_exact": {"seconds": round(te, 4), "kernel_ms": round(ke, 3),
                                                               "kernel": "score_coded_kernel<exact>" if state == 1 else "score_sweep_kernel<exact>"},
                                        "get_pop_like_float32_mode": {"seconds": round(tf, 4), "kernel_ms": round(kf, 3), "kernel": "score_sweep_kernel<fast>",
                                                                      "max_rel_dev_of_sums_vs_exact": float(np.max(np.abs(of - oe) / np.abs(oe)))},
                                        "price_of_exact": round(ke / kf, 2) if kf > 0 else None,
                                        "rule": "exact (bit-identical per-site values) unless WGSASSIGN_MODE=fast; the library does not switch by itself: "
                                                "the float32 mode's deviation is measured per matrix, not bounded a priori",
                                        "chosen": "exact"}
    afs.close()
    b.close()
    return res


def ingest_path(ctx, device, n, m, K):
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import beagle_files
    from wgsassign_amd import reader_cy
    d = tempfile.mkdtemp(prefix="wgs_bench_")
    old = {k: os.environ.get(k) for k in ("WGSASSIGN_INDEX_DIR", "WGSASSIGN_INFLATE")}
    try:
        os.environ["WGSASSIGN_INDEX_DIR"] = d
        path = os.path.join(d, "shard.beagle.gz")
        t0 = time.perf_counter()
        text_bytes, vals, pick = beagle_files.write_lowdepth_bgzf(path, n, m)
        res = {"file": "BGZF, %d sites x %d individuals, %.0f MB of text in %.0f MB" % (m, n, text_bytes / 1e6, os.path.getsize(path) / 1e6),
               "file_written_in_s": round(time.perf_counter() - t0, 2), "host_threads": reader_cy.host_threads()}
        group_of = (np.arange(n) % K).astype(np.int32)
        t0 = time.perf_counter()
        reader_cy.ensure_index(path)
        res["index_pass_seconds"] = round(time.perf_counter() - t0, 3)
        for label, inflate in (("device_inflate", "device"), ("host_inflate", "host")):
            os.environ["WGSASSIGN_INFLATE"] = inflate
            best = None
            for _ in range(2):                       # the second run has the page-locked staging and the code objects warm
                mal0 = device.malloc_seconds()
                t0 = time.perf_counter()
                b, _, _, _ = reader_cy.stream_to_device(path, group_of, K, ctx=ctx, names="ends")
                ctx.sync()
                dt = time.perf_counter() - t0
                st = dict(b.ingest_stats)
                st["malloc_s"] = device.malloc_seconds() - mal0
                probe = [0, m // 3, m - 1]
                same = all(b.download_rows(r, 1).tobytes() == vals[pick[r]].tobytes() for r in probe)
                b.close()
                if best is None or dt < best[0]:
                    best = (dt, st, same)
            dt, st, same = best
            res[label] = {"seconds": round(dt, 4), "sites_per_s": round(m / dt), "text_GB_per_s": round(text_bytes / 1e9 / dt, 2),
                          "device_ms": round(st["device_ms"], 1), "inflate_kernel_ms": round(st["device_inflate_kernel_ms"], 1),
                          # (hipMalloc of VRAM an earlier process used is cleared by the driver: 0.3 ms or 0.1 s for the same 1.6 GB matrix,
                          # by the box -- profiles/r04_alloc_ubench.txt)
                          "of_which_hipMalloc_ms": round(st["malloc_s"] * 1e3, 1),
                          "waited_for_producer_s": round(st["wait_s"], 3), "lines_parsed_on_host": int(st["host_lines"]),
                          "members_left_to_host_inflater": int(st["blocks_left_to_host_inflater"]), "probed_rows_equal_source": bool(same)}
        # the file as a first run meets it -- no index in the cache: index pass and device ingest at the same time
        # (reader_cy._stream_cold_file), against index_pass_seconds + device_inflate.seconds above for one after the other
        os.environ["WGSASSIGN_INFLATE"] = "device"
        best = None
        for _ in range(2):
            for f in reader_cy.index_paths(path):
                if os.path.exists(f):
                    os.unlink(f)
            mal0 = device.malloc_seconds()
            t0 = time.perf_counter()
            b, _, _, m_seen = reader_cy.stream_to_device(path, group_of, K, ctx=ctx, names="ends")
            ctx.sync()
            dt = time.perf_counter() - t0
            mal = device.malloc_seconds() - mal0
            same = m_seen == m and b.m == m and all(b.download_rows(r, 1).tobytes() == vals[pick[r]].tobytes() for r in (0, m // 3, m - 1))
            indexed = os.path.exists(reader_cy.index_paths(path)[0])
            b.close()
            if best is None or dt < best[0]:
                best = (dt, mal, same, indexed)
        res["cold_file_one_pass"] = {"seconds": round(best[0], 4), "of_which_hipMalloc_ms": round(best[1] * 1e3, 1),
                                     "index_pass_plus_ingest_seconds": round(res["index_pass_seconds"] + res["device_inflate"]["seconds"], 4),
                                     "probed_rows_equal_source": bool(best[2]), "index_left_in_cache": bool(best[3])}
        return res
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        shutil.rmtree(d, ignore_errors=True)


def committed_pmc(m, n, K, mode, coded_em=False, coded_score=False):
    """Counters of the EM sweep and the scoring sweep from the committed rocprofv3 PMC passes of THIS workload
    (profiles/*/pmc_summary.json, written by tools/summarize_profile.py: traffic = (2*FETCH_SIZE + WRITE_SIZE)
    * 1024 with the gfx950 correction, valu_busy_frac = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE
    / 8)); the newest matching profile wins, {} when none matches the workload being run."""
    import glob
    import re
    from wgsassign_amd import _lib
    loaded = _lib.load().wgs_kernels_id().decode()
    best = {"reason": "no committed profile of this workload (profiles/*/pmc_summary.json)"}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("bench_config") != {"snps_per_gpu": m, "n": n, "K": K, "mode": mode}:
            continue
        if d.get("kernels_id") != loaded:
            # counters of other code say nothing about the kernels being timed now
            if "em_traffic" not in best:
                best = {"reason": "the committed profile of this workload (%s) was taken from kernels %s, the loaded library has %s: "
                                  "re-profile (tools/gpu_prof.sh + tools/summarize_profile.py)" % (os.path.relpath(f, ROOT), d.get("kernels_id"), loaded)}
            continue
        cur = {"source": os.path.relpath(f, ROOT), "kernels_id": loaded}
        for k, e in d.get("kernels", {}).items():
            if ("em_coded_kernel" if coded_em else "em_sweep_kernel<%d" % (0 if mode == "exact" else 1)) in k:
                cur["em_traffic"] = e.get("traffic_bytes_per_launch")
                cur["em_valu_busy_frac"] = e.get("valu_busy_frac")
                cur["em_clock_ghz"] = e.get("effective_clock_ghz")
            ck = re.search(r"score_coded_kernel<(\d+), (\d+)(?:, \w+)?>", k)
            if coded_score and ck and int(ck.group(2)) == (0 if mode == "exact" else 1):
                cur["assign_traffic"] = e.get("traffic_bytes_per_launch")
                cur["assign_valu_busy_frac"] = e.get("valu_busy_frac")
                cur["assign_insts_valu"] = e.get("SQ_INSTS_VALU", {}).get("mean")
            sk = re.search(r"score_sweep_kernel<(\d+), (\d+), (\d+), (true|false)>", k)
            if not coded_score and sk and int(sk.group(3)) == (0 if mode == "exact" else 1) and sk.group(4) == "false":
                # ONE launch scores the whole matrix (template arguments: KB, NP, MODE, PER_IND)
                cur["assign_traffic"] = e.get("traffic_bytes_per_launch")
                cur["assign_valu_busy_frac"] = e.get("valu_busy_frac")
                cur["assign_insts_valu"] = e.get("SQ_INSTS_VALU", {}).get("mean")
        if cur.get("em_traffic") is not None:
            best = cur
    return best


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(beagle, group_of, K, ms, seconds=12.0):
    """Reference-shaped CPU path on this box's host cores: for each population gather its
    columns (WGSassign.py:227-233) and run emMAF_update (emMAF_cy.pyx:10-23) -- the oracle's
    bit-exact C/OpenMP restatement -- on the first `ms` SNPs of the same synthetic matrix."""
    from oracle import oracle as orc
    orc.build()
    from wgsassign_amd.comm import usable_cpus
    affinity = len(os.sched_getaffinity(0))
    # every CPU this process can keep busy (OMP_NUM_THREADS = nproc): the affinity set cut down to the control
    # group's CPU quota -- the one-GPU box shows 256 CPUs with a quota of 16, where 256 OpenMP threads run 50x slower than 16
    threads = int(os.environ.get("WGS_CPU_THREADS", usable_cpus()))
    rows = beagle.download_rows(0, ms)
    slabs = [orc.gather(rows, np.flatnonzero(group_of == k), threads) for k in range(K)]
    fs = [np.full(ms, 0.25, dtype=np.float32) for _ in range(K)]
    for k in range(K):                           # warm-up sweep
        orc.emMAF_update(slabs[k], fs[k], threads)
    sweeps, t0 = 0, time.perf_counter()
    while True:
        for k in range(K):
            orc.emMAF_update(slabs[k], fs[k], threads)
        sweeps += 1
        el = time.perf_counter() - t0
        if el > seconds:
            break
    t_g0 = time.perf_counter()
    for k in range(K):
        orc.gather(rows, np.flatnonzero(group_of == k), threads)
    t_gather = time.perf_counter() - t_g0
    # assignment leg of the reference shape: one scan of L per (individual, population) pair
    # (glassy.py:31-38); every pair costs the same (one pass over the sample), so a timed run of pairs spread over all
    # individuals and populations scales to the n*K pairs of the full output
    A = np.ascontiguousarray(np.stack(fs, axis=1))
    pairs, t_a0 = 0, time.perf_counter()
    while pairs < 6 or time.perf_counter() - t_a0 < min(6.0, seconds / 2):
        vec = np.zeros(ms, dtype=np.float32)
        orc.loglike(rows, A, vec, threads, (pairs * 37) % beagle.n, pairs % K)
        float(np.sum(vec, dtype=float))
        pairs += 1
    t_pair = (time.perf_counter() - t_a0) / pairs
    assign_snps_per_s = ms / (t_pair * beagle.n * K)
    return {"value": K * ms * sweeps / el, "unit": "SNP-updates/s", "cores": threads, "nproc": os.cpu_count(), "affinity": affinity, "cgroup_cpu_quota": usable_cpus(),
            "cpu_model": cpu_model(), "kind": "port",
            "assign_value": assign_snps_per_s, "assign_unit": "SNPs/s (all n x K terms of a SNP = 1)",
            "assign_extrapolated": True, "assign_pairs_timed": pairs, "assign_pairs_of_full_output": int(beagle.n * K),
            "sample": "first %d SNPs x %d ind of the same synthetic matrix, K=%d populations, %d sweeps in %.1f s "
                      "(OpenMP threads=%d; per-population gather %.2f s not included); assignment leg: %d (individual, population) "
                      "scans over the sample timed, %.2f ms each, of the %d a full matrix needs" %
                      (ms, beagle.n, K, sweeps, el, threads, t_gather, pairs, t_pair * 1e3, beagle.n * K)}


if __name__ == "__main__":
    main()
