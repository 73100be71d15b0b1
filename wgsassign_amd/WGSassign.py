"""`WGSassign` command line: drop-in for the hot-path options of the reference's
WGSassign.py (--get_reference_af, --loo, --get_pop_like), running on an AMD MI355X.

Same flags, defaults, stdout lines and output files as the reference (WGSassign.py:24-104,
109-308), including --ne_obs (Fisher information / effective sample sizes).  Options of the
reference that are outside this build's scope (z-scores, mixture proportions) are recognised and
refused with a message.  Installed as the console script `WGSassign` (pyproject.toml; reference
setup.py:48-50).
"""
import argparse
import os
import sys
from datetime import datetime

parser = argparse.ArgumentParser(prog="WGSassign")
parser.add_argument("-b", "--beagle", metavar="FILE",
                    help="Filepath to genotype likelihoods in gzipped Beagle format from ANGSD")
parser.add_argument("-t", "--threads", metavar="INT", type=int, default=1,
                    help="Number of threads (1: chosen automatically). The arithmetic runs on the GPU; the host threads of "
                         "this node's job inflate the Beagle file -- with several ranks the budget is divided between them")
parser.add_argument("-o", "--out", metavar="OUTPUT", default="wgsassign", help="Prefix for output files")
parser.add_argument("--maf_iter", metavar="INT", type=int, default=200,
                    help="Maximum iterations for minor allele frequencies estimation - EM (200)")
parser.add_argument("--maf_tole", metavar="FLOAT", type=float, default=1e-4,
                    help="Tolerance for minor allele frequencies estimation update - EM (1e-4)")
parser.add_argument("--pop_af_IDs", metavar="FILE", help="Filepath to individual IDs and populations for beagle")
parser.add_argument("--get_reference_af", action="store_true",
                    help="Estimate allele frequencies for reference populations")
parser.add_argument("--pop_names", metavar="FILE", help="Filepath to population names of allele frequency file")
parser.add_argument("--loo", action="store_true", help="Perform leave-one-out cross validation")
parser.add_argument("--loo_downsampled_beagle", metavar="FILE",
                    help="Optional Beagle file of downsampled genotype likelihoods to use for LOO assignment.")
parser.add_argument("--pop_af_file", metavar="FILE", help="Filepath to reference population allele frequencies")
parser.add_argument("--get_pop_like", action="store_true",
                    help="Estimate log likelihood of individual assignment to each reference population")
parser.add_argument("--partition_sites", type=int, metavar="INT", default=1,
                    help="Optional: partition sites into INT subsets (by modulo) and report assignment "
                         "log-likelihoods for each subset.")
parser.add_argument("--ne_obs", action="store_true",
                    help="Estimate population and individuals effective sample sizes")
parser.add_argument("--gpus", metavar="INT", type=int, default=1,
                    help="MI355X build: shard the SNPs over INT GPUs of this node, one process per GPU "
                         "(not needed under torchrun or any launcher that sets RANK / WORLD_SIZE)")
# recognised but not provided by this build (out of the hot-path scope)
for _flag in ("--get_assignment_z_score", "--get_reference_z_score", "--single_read_threshold",
              "--get_em_mix", "--get_mcmc_mix"):
    parser.add_argument(_flag, action="store_true", help=argparse.SUPPRESS)
for _flag in ("--ind_ad_file", "--allele_count_threshold", "--ind_start", "--ind_end", "--pop_like", "--pop_like_IDs",
              "--mixture_iter"):
    parser.add_argument(_flag, help=argparse.SUPPRESS)


def _run(args, comm):
    """The hot-path options on device-resident data.  Under torchrun (one process per GPU) the SNPs are
    sharded over the ranks: every rank parses and holds only its contiguous SNP range; the EM convergence sums,
    the serial-chain carry and the n x K log-likelihood sums cross ranks through one sum
    all-reduce (RCCL); rank 0 writes the reference's output files."""
    import numpy as np

    from . import emMAF, fisher, glassy, reader_cy, utils
    from .device import AFSet, assign, get_context

    root = comm.rank == 0

    def say(*a):
        if root:
            print(*a)

    ctx = get_context()
    reader_cy.set_threads(args.threads)
    say("Parsing Beagle file.")
    assert os.path.isfile(args.beagle), "Beagle file doesn't exist!"
    IDs = pops = None
    group_of, n_groups = None, 1
    if args.get_reference_af:
        assert os.path.isfile(args.pop_af_IDs), "Reference population ID file does not exist!!"
        IDs = np.loadtxt(args.pop_af_IDs, delimiter="\t", dtype="str")
        pops = np.unique(IDs[:, 1])
        group_of, n_groups = np.searchsorted(pops, IDs[:, 1]).astype(np.int32), len(pops)

    def summary(samples, m_sites, ends):
        heads = [x for e in ends for x in e[0]]
        tails = [x for e in ends for x in e[1]]
        shown = heads[:m_sites] if m_sites <= 4 else heads[:2] + tails[-2:]
        print(f"sample_names: {len(samples)} samples total: {utils.preview(samples)}")
        print(f"site_names: {m_sites} sites total: " + (", ".join(shown) if m_sites <= 4 else
                                                          ", ".join(shown[:2]) + ", ..., " + ", ".join(shown[2:])))

    scored = None
    if args.loo_downsampled_beagle:
        # WGSassign.py:172-198 with names-only passes: every rank derives the same two site masks, then
        # parses only its range of the KEPT sites of each file
        assert os.path.isfile(args.loo_downsampled_beagle), "Downsampled beagle file doesn't exist!"
        sample_names, names_ref = reader_cy.read_site_names(args.beagle, comm)
        sample_names_ds, names_ds = reader_cy.read_site_names(args.loo_downsampled_beagle, comm)
        n = len(sample_names)
        say("Loaded " + str(len(names_ref)) + " sites and " + str(n) + " individuals.")
        if root:
            utils.print_sample_and_site_summary(sample_names, names_ref)
        say("Parsing the optional downsampled Beagle file.")
        say("Loaded optional downsampled data set with " + str(len(names_ref)) + " sites and " + str(n) + " individuals.")
        if root:
            utils.print_sample_and_site_summary(sample_names_ds, names_ds)
        if sample_names != sample_names_ds:
            raise ValueError("Sample names in downsampled Beagle file do not match original.")
        say("Retaining only sites from the reference that are in the downsampled beagle file:")
        keep_ref = utils.site_mask(names_ref, names_ds)
        if root and int(np.sum(~keep_ref)) > 0:
            print(f"\tFiltered out {int(np.sum(~keep_ref))} sites not present in the target site list.")
        kept_ref = [x for x, k in zip(names_ref, keep_ref) if k]
        say("Removing sites from downsampled set that were not in the reference (should not occur...):")
        keep_ds = utils.site_mask(names_ds, kept_ref)
        if root and int(np.sum(~keep_ds)) > 0:
            print(f"\tFiltered out {int(np.sum(~keep_ds))} sites not present in the target site list.")
        if kept_ref != [x for x, k in zip(names_ds, keep_ds) if k]:
            raise ValueError("Site names in full and downsampled Beagle do not match after filtering.")
        beagle, _, site_names, m = reader_cy.stream_to_device(args.beagle, group_of, n_groups, ctx=ctx, rank=comm.rank,
                                                              world=comm.world, keep=keep_ref, comm=comm)
        scored, _, _, _ = reader_cy.stream_to_device(args.loo_downsampled_beagle, group_of, n_groups, ctx=ctx,
                                                     rank=comm.rank, world=comm.world, keep=keep_ds, comm=comm)
    else:
        beagle, sample_names, site_names, m = reader_cy.stream_to_device(
            args.beagle, group_of, n_groups, ctx=ctx, rank=comm.rank, world=comm.world, comm=comm, names="ends")
        n = beagle.n
        say("Loaded " + str(m) + " sites and " + str(n) + " individuals.")
        ends = comm.allgather_object((site_names[:4], site_names[-4:]))
        if root:
            summary(sample_names, m, ends)

    if args.get_reference_af:
        say("Parsing reference population ID file.")
        assert (n == IDs.shape[0]), "Number of individuals in beagle and reference ID file do not match!"
        import contextlib
        import io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            pops, af, _ = emMAF.emMAF_populations(None, IDs, args.maf_iter, args.maf_tole, beagle=beagle, comm=comm)
        if root:
            sys.stdout.write(buf.getvalue())
        af_full = comm.gather_rows(af)
        if root:
            np.save(args.out + ".pop_af", af_full)
        say("Saved reference population allele frequencies as " + str(args.out) + ".pop_af.npy (Binary - np.float32)\n")
        say("Column order of populations is: " + str(pops))
        if root:
            np.savetxt(args.out + ".pop_names.txt", pops, fmt="%s")
        say("Saved reference population names as " + str(args.out) +
            ".pop_names.txt (String: Order of pops for .pop_af.npy, .ne_obs.npy, and fisher_obs.npy files)\n")
        if args.ne_obs:                      # per-SNP quantities shard trivially; rank 0 assembles and writes
            say("Estimating Fisher information.")
            f_loc, ne_loc = fisher.fisher_obs(None, af, IDs, args.threads, beagle=beagle)
            f_obs, ne_obs = comm.gather_rows(f_loc), comm.gather_rows(ne_loc)
            if root:
                np.save(args.out + ".fisher_obs", f_obs)
            say("Saved reference population observed Fisher information per locus as " + str(args.out) +
                ".fisher_obs.npy (Binary - np.float32)\n")
            if root:
                np.save(args.out + ".ne_obs", ne_obs)
            say("Saved reference population effective sample size estimates per locus as " + str(args.out) +
                ".ne_obs.npy (Binary - np.float32)\n")
            if root:
                ne_obs_mean_out = np.empty((2, len(pops)), dtype=np.dtype('U25'))
                ne_obs_mean_out[0, :] = pops
                ne_obs_mean_out[1, :] = np.mean(ne_obs, axis=0)
                np.savetxt(args.out + ".ne_obs.txt", ne_obs_mean_out, fmt="%s")
            say("Saved reference population effective sample size estimates as " + str(args.out) +
                ".ne_obs.txt (String - np.U25)\n")
            say("Estimating individual effective sample sizes.")
            # np.mean's running float32 total is handed from SNP shard to SNP shard (fisher.fisher_obs_ind)
            ne_ind_full = fisher.fisher_obs_ind(None, af, IDs, args.threads, beagle=beagle, comm=comm, m_total=m)
            if root:
                np.savetxt(args.out + ".ne_ind.txt", ne_ind_full.reshape(-1, 1), fmt="%.7f")
            say("Save individual effective sample sizes as " + str(args.out) + ".ne_ind.txt")

        if args.loo:
            say("Performing leave-one-out cross validation.")
            say(str(n) + " individuals to assign to " + str(len(pops)) + " populations")
            if scored is not None:
                say("Using downsampled GLs for likelihood evaluation in LOO assignment.")
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ll, parts = glassy.loo_device(beagle, scored if scored is not None else beagle, af, group_of,
                                              args.maf_iter, args.maf_tole,
                                              args.partition_sites, comm=comm, need_parts=args.partition_sites > 1)
            if root:
                sys.stdout.write(buf.getvalue())
                suffix = "_downsampled" if scored is not None else ""
                outfile = f"{args.out}.pop_like_LOO{suffix}.tsv"
                partfile = f"{args.out}.pop_like_LOO{suffix}_partitions_{args.partition_sites}.tsv.gz"
                utils.write_ass_mats(outfile, ll, sample_names, pops, print_part_column=False,
                                     sample_locations=IDs[:, 1], doing_LOO=True)
                print(f"Saved leave-one-out cross validation log likelihoods as {outfile}")
                if args.partition_sites > 1:
                    utils.write_ass_mats(partfile, parts, sample_names, pops, partition_count=args.partition_sites,
                                         print_part_column=True, sample_locations=IDs[:, 1], doing_LOO=True)
                    print(f"Saved leave-one-out cross validation log likelihoods from partitioned sites as {partfile}")
                print(f"Column order of populations is: {pops}")

    if args.get_pop_like:
        say("Parsing population allele frequency file.")
        assert os.path.isfile(args.pop_af_file), "Population allele frequency file does not exist!!"
        from .comm import shard_range
        lo, hi = shard_range(m, comm.rank, comm.world)
        A = np.ascontiguousarray(np.load(args.pop_af_file, mmap_mode="r")[lo:hi], dtype=np.float32)
        say("Calculating likelihood of population assignment")
        say(str(n) + " individuals to assign to " + str(A.shape[1]) + " populations")
        afs = AFSet.from_host(A, ctx=ctx)
        out, _ = assign(beagle, afs, comm=comm)
        afs.close()
        if root:
            np.savetxt(args.out + ".pop_like.txt", out.astype(np.float32), fmt="%.7f")
        say("Saved population assignment log likelihoods as " + str(args.out) + ".pop_like.txt (text)")
    if scored is not None:
        scored.close()
    beagle.close()
    comm.barrier()


def main(argv=None):
    args = parser.parse_args(argv)
    if len(sys.argv) < 2 and argv is None:
        parser.print_help()
        sys.exit()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU, started here before anything touches the device; they come back through this function
        # with RANK / WORLD_SIZE set
        from .comm import launch_local_ranks
        sys.exit(launch_local_ranks(args.gpus, [sys.executable, "-m", "wgsassign_amd.WGSassign"] +
                                    list(sys.argv[1:] if argv is None else argv)))
    from .comm import init_from_env
    comm = init_from_env()          # LocalComm for one process; one rank per GPU under --gpus N / torchrun
    root = comm.rank == 0
    if root:
        print("WGSassign")
        print("Matt DeSaix.")
        print("Using " + str(args.threads) + " thread(s).\n")

    if args.loo_downsampled_beagle and not args.loo:
        raise ValueError("The --loo_downsampled_beagle option requires that --loo is also specified.")
    for unsupported in ("get_assignment_z_score", "get_reference_z_score", "get_em_mix", "get_mcmc_mix"):
        if getattr(args, unsupported):
            raise SystemExit("--%s is outside the scope of the MI355X build (EM allele frequencies, Fisher "
                             "information, leave-one-out and assignment likelihoods only)" % unsupported)

    if root:        # log-file of non-default arguments (WGSassign.py:127-141)
        full, deaf = vars(args), vars(parser.parse_args([]))
        with open(args.out + ".args", "w") as fh:
            fh.write("WGSassign\n")
            fh.write("Time: " + datetime.now().strftime("%d/%m/%Y %H:%M:%S") + "\n")
            fh.write("Directory: " + str(os.getcwd()) + "\n")
            fh.write("Options:\n")
            for key in full:
                if full[key] != deaf[key]:
                    if type(full[key]) is bool:
                        fh.write("\t-" + str(key) + "\n")
                    else:
                        fh.write("\t-" + str(key) + " " + str(full[key]) + "\n")

    if args.beagle is None:
        return
    # One code path for one or many GPUs: the Beagle file is streamed chunk by chunk into the device
    # slabs (host memory stays at one chunk; the reference holds two full copies of the matrix).
    from .comm import COMM_DIVERGED, CollectiveMismatch
    try:
        return _run(args, comm)
    except CollectiveMismatch as e:
        # the ranks issued different collectives: nothing computed from here on could be trusted, and nothing is retried --
        # every rank that sees it leaves with its own message; the launcher ends the others and reports this status
        print("wgsassign_amd: rank %d: %s" % (comm.rank, e), file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(COMM_DIVERGED)         # (not sys.exit: a peer blocked inside a collective would keep atexit handlers waiting)


if __name__ == "__main__":
    main()
