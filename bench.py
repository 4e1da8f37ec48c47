#!/usr/bin/env python3
"""bench.py -- pair-distances/s of the gen_dist() hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5|emboot] [--kernel ...]

One "step" = one job of the hot path over the resident data set: accumulation
kernel(s) -> deterministic slab reduction -> copy to host -> /cnt and
evolutionary-model transform with the host's libm (the tail of gen_dist,
ngsDist.cpp:372-401), with a reduce-scatter and an all-gather per job when N > 1.
Inputs are synthetic (counter-based generator), generated ON the GPU
before the timed region.  The timed K steps -- `value`, `ms_per_step` -- run one job
at a time: a job's tail ends before the next job's kernels start, so ms_per_step
is ONE job's latency, what a real run (one job) sees.  At N = 1 a second region
of K steps then runs the jobs two deep (job k's tail on a worker thread beside
job k+1's kernels) and is reported apart, as `pipelined` (throughput of a stream
of jobs; never `value`).  `--gpus N` without a RANK in the environment starts the
N ranks itself (torch.distributed.run).

N > 1 (--shard): the SAME job at every N unless --shard replicates is asked for.
  sites (default; strong scaling): each GPU holds 1/N of the sites of all individuals and computes every pair over
      its range; one RCCL reduce-scatter adds the partial sums and leaves each rank 1/N of the cells, whose tail
      (/cnt, evolutionary model: host libm) it runs on its own host cores; one RCCL all-gather of the finished cells;
  pairs (strong scaling): pair tiles dealt over GPUs, input replicated, disjoint results; same two collectives;
  replicates (weak scaling, single-matrix workloads only: the job GROWS with N): every GPU holds the data set and
      computes ONE matrix -- rank 0 the full-data matrix, rank r the r-th bootstrap replicate at the reference's
      default --boot_block_size 1, drawn from its taus stream -- and one RCCL all-gather of the finished matrices.

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit,
`roofline` for the dominant kernel from HIP-event timings taken inside this run,
and `cpu_baseline` = the CPU oracle (a port of the reference algorithm, NOT the
product path) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic cost per pair-site
BYTES_PER_PAIR_SITE = 48.0  # stream model: two 3-double GL vectors per pair-site
FLOPS_PER_PAIR_SITE = 6.0   # tiled model: 3 FP64 FMA per pair-site (P . Q^T, K = 3*n_sites)
PEAK_FP64_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (SURVEY 8d hardware constants)
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
# EM path: the per-pair algorithm's work (round 1's kernel), for comparison with what the table kernel executes
MEAN_EM_STEPS = 11.7  # EM steps per pair-site until emOptim2.cpp:127 stops, mean on the synthetic data (3 ... 50)
EMFAST_OPS_PER_PAIR_SITE = 272.0  # k_accum_em<fast>: SQ_THREAD_CYCLES_VALU per pair-site, profiles/r01_cfg4_em_pmc.md

WORKLOADS = {
    # BASELINE.json configs[1..4]
    "cfg2": dict(n_ind=200, n_sites=100_000, indep=True, evol_model=0, seed=2, n_boot=0, block=1),
    "cfg3": dict(n_ind=1000, n_sites=1_000_000, indep=True, evol_model=1, seed=3, n_boot=0, block=1),
    "cfg4": dict(n_ind=1000, n_sites=1_000_000, indep=False, evol_model=2, seed=3, n_boot=0, block=1),
    "cfg5": dict(n_ind=500, n_sites=500_000, indep=True, evol_model=1, seed=5, n_boot=64, block=1000),
    # The reference's own kind of bootstrap run (examples/test.sh:24-25: the EM path, --n_boot_rep, --boot_block_size 10;
    # parse_args.cpp:29-31) at cfg 4's shape on 1e5 sites: 101 matrices from ONE pass of the per-site EM -- the terms of
    # every (pair, unit of 10 sites) spilled once, one FP64 MFMA contraction with every matrix's weights (em_spill_impl).
    # Per-block partial results are switched off (10 000 blocks x 8.4 MB = an 84 GB slab, which the engine only buys after
    # a few jobs have paid its allocation -- since round 6 also where the alternative is this plan): this is the plan a
    # single job gets, pinned here so that every step of the bench is that job.  --n_boot / --block vary the job.
    "emboot": dict(n_ind=1000, n_sites=100_000, indep=False, evol_model=2, seed=3, n_boot=100, block=10,
                   options={"boot_partials": 0}),
}


def self_launch(n):
    """Start `n` ranks of this script under torch.distributed.run on this node (127.0.0.1 rendezvous, a free port)
    and return the launcher's exit code.  The children are new processes (never an exec of this one)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def host_cpu():
    """The host the cpu_baseline leg runs on (north_star: "the reference's --n_threads CPU path timed on the GPU box's own
    host cores, core count stated"): logical cores of the box, the CPU model, and the cores THIS process may use -- its
    affinity mask cut down to the container's CPU quota (cgroup cpu.max) where there is one -- which is the thread count
    the baseline takes (the reference clamps --n_threads to the number of pairs, ngsDist.cpp:44-52)."""
    info = {"host_cores": os.cpu_count() or 1, "cpu_model": None, "affinity_cores": None, "cgroup_cpu_quota": None}
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                info["cpu_model"] = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        info["affinity_cores"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(f).read().split()
            if f.endswith("cpu.max"):
                if parts[0] != "max":
                    info["cgroup_cpu_quota"] = float(parts[0]) / float(parts[1])
            else:
                q = float(parts[0])
                if q > 0:
                    info["cgroup_cpu_quota"] = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    usable = info["affinity_cores"] or info["host_cores"]
    if info["cgroup_cpu_quota"]:
        usable = max(1, min(usable, int(info["cgroup_cpu_quota"] + 0.5)))
    info["threads"] = int(usable)
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # the first two launches after the fill run at a lower clock
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "stream", "mfma", "em_table", "em_fast", "em_faithful"])
    ap.add_argument("--host_results", action="store_true", help="N = 1: the engine writes the job's sums straight into pinned host "
                    "memory (its reduction kernels' stores cross the link) instead of device memory + a copy")
    ap.add_argument("--n_sites", type=int, default=0, help="override the workload's n_sites (not a valid bench line)")
    ap.add_argument("--n_boot", type=int, default=-1, help="override the workload's --n_boot_rep (bootstrap workloads)")
    ap.add_argument("--block", type=int, default=0, help="override the workload's --boot_block_size (bootstrap workloads)")
    ap.add_argument("--single_image", type=int, nargs="?", const=1, default=0,
                    help="ngd_config.single_image: 0 = the engine's choice (one image in congruent coordinates + the fix-up pass "
                         "above 384 padded individuals), 1: one image, the other formed a range of sites at a time (memory for "
                         "time), 2: both operands from one image in congruent coordinates, 3: two images")
    ap.add_argument("--single_image_gb", type=float, default=0.0,
                    help="with --single_image: GB of the second image formed at a time (NGD_OPT_SINGLE_IMAGE_BYTES; 0 = 4)")
    ap.add_argument("--second_image_gb", type=float, default=0.0,
                    help="with --single_image: GB of the second image kept resident all the same (ngd_config.second_image_mib)")
    ap.add_argument("--exact_shapes", type=int, default=0,
                    help="ngd_config.exact_shapes: the MFMA kernel's block form (0 = the engine's choice; experiments)")
    ap.add_argument("--cpu_sites", type=int, default=0, help="sites in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no_cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu_threads", type=int, default=0,
                    help="threads of the cpu_baseline leg (0 = every core this process may use: affinity mask and cgroup quota)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --same_device rehearses the N>1 flow on a 1-GPU box")
    ap.add_argument("--same_device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--set_option", action="append", default=[], metavar="NAME=VALUE",
                    help="ngd_set_option on the engine before the run (A/B measurements, e.g. sign_form=0)")
    ap.add_argument("--no_preflight", action="store_true",
                    help="N > 1: skip tools/rccl_preflight.py (the job's collectives timed once before the engine is created)")
    ap.add_argument("--shard", default="auto", choices=["auto", "replicates", "sites", "pairs"],
                    help="N>1: split the site axis (each rank holds 1/N of the data, all pairs; sums are added), or deal "
                         "pair tiles over ranks (input replicated; disjoint results) -- both time the SAME job at every "
                         "N; or one matrix (bootstrap replicate) per GPU (weak scaling: the job grows with N); "
                         "auto = sites")
    ap.add_argument("--pairwise_del", action="store_true",
                    help="--pairwise_del with --miss_frac missing sites: counts differ per pair, so the N>1 flow also "
                         "reduce-scatters the valid-site counts (not a BASELINE configuration)")
    ap.add_argument("--miss_frac", type=float, default=0.0, help="fraction of exact (1/3,1/3,1/3) sites in the input")
    ap.add_argument("--split_tail", action="store_true",
                    help="N = 1 bootstrap jobs: the round-5 form of a job's tail -- the engine call ends when the sums are in "
                         "device memory, THEN they are copied out in chunks and ngd_finish_stream works them -- instead of "
                         "ngd_run_mult_batch_dist (groups of replicates copied out while the later ones are reduced, the tail "
                         "inside the call)")
    ap.add_argument("--serial_tail", action="store_true",
                    help="only the headline region (one job at a time: a job's tail -- copy out, /cnt, evolutionary model; "
                         "N > 1: the collectives too -- ends before the next job's kernels start); skip the second, "
                         "pipelined region")
    ap.add_argument("--pipelined_tail", action="store_true",
                    help="run the second (pipelined) region also where it is skipped by default: jobs of under 1e5 cells, "
                         "and N > 1 (collectives on the worker thread: rehearsed over gloo only)")
    ap.add_argument("--poll_sclk", action="store_true",
                    help="poll pp_dpm_sclk from a thread during the timed region (off by default: it shares the interpreter "
                         "with the timed loop; the clock reported is the one sampled inside the kernel, shader_clock_mhz)")
    ap.add_argument("--vary_jobs", action="store_true",
                    help="test path: odd steps compute another job than even steps (a bootstrap replicate instead of the "
                         "full-data matrix / the replicates in reverse order), every tail records a checksum of its whole "
                         "result, and the pipelined region's checksums must equal the serial region's step by step -- a "
                         "mix-up of the two buffer sets cannot pass (not a bench line)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` by itself: this process touches neither torch nor the GPU; it starts the
        # N ranks as fresh children (one process per GPU, torch.distributed.run) and relays their output and exit code
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch  # first: one HIP runtime in the process (see ngsdist_amd/_lib.py)
    import torch.distributed as dist

    import ngsdist_amd as N
    from ngsdist_amd.dist import gather_cells, gather_matrices, owned_cells, scatter_sum, share_of, unpack_cells

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (N > 1 needs a torch.distributed.run launch with "
                         "--nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the product path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            probe = torch.zeros(1, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(probe)  # the communicator really comes up here (lazy under RCCL): fail now, not in a step
            if args.backend == "nccl":
                torch.cuda.synchronize()
        except Exception as exc:  # no JSON line, no retry in this process: the launcher sees a non-zero exit
            sys.stderr.write("bench.py: rank %d of %d (cuda:%d, backend %s): collective initialisation failed: %r\n"
                             % (rank, world, local_rank, args.backend, exc))
            sys.stderr.flush()
            os._exit(3)

    W = dict(WORKLOADS[args.workload])
    if args.n_sites:
        W["n_sites"] = args.n_sites
    if args.n_boot >= 0 or args.block:
        if not W["n_boot"]:
            raise SystemExit("bench.py: --n_boot / --block vary a bootstrap workload (cfg5, emboot)")
        W["n_boot"] = args.n_boot if args.n_boot > 0 else W["n_boot"]
        W["block"] = args.block or W["block"]
    n_ind, n_sites = W["n_ind"], W["n_sites"]
    n_pairs = N.n_pairs(n_ind)
    kernel = args.kernel
    if kernel == "auto":
        kernel = "mfma" if W["indep"] else "em_table"

    shard = args.shard
    if shard == "auto":
        shard = "sites"
    if world == 1:
        shard = "none"
    by_sites, by_reps = shard == "sites", shard == "replicates"
    # pair tiles are disjoint: a rank finishes its own cells and ONE all-gather of finished cells ends the job
    by_pairs = shard == "pairs" and world > 1
    pdel = args.pairwise_del
    if pdel and by_reps:
        raise SystemExit("bench.py: --pairwise_del is wired for --shard sites|pairs")
    if by_reps:
        if W["n_boot"]:
            raise SystemExit("bench.py: --shard replicates is for the single-matrix workloads")
        # one matrix per GPU: the full-data matrix + (N-1) bootstrap replicates at the reference's default block size
        W["n_boot"], W["block"] = world - 1, 1

    # bootstrap geometry of the WHOLE data set
    n_mat = W["n_boot"] + 1
    if world > 1 and not args.no_preflight:
        # The job's collectives at their real sizes, timed BEFORE the engine exists (tools/rccl_preflight.py): no multi-GPU
        # node was available to this build, so the first hardware run prints what its reduce-scatter / all-gather cost on
        # the node it found, beside the RCCL version -- one "rccl_preflight {...}" line on rank 0's stderr
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from rccl_preflight import preflight
        pf_err = None
        try:
            preflight(dist, dev, args.backend, n_mat * n_pairs, n_pairs if by_reps else 0, repeats=3)
        except Exception as exc:  # (a collective that returns WRONG data is a SystemExit on EVERY rank and ends the run)
            pf_err = exc
        # Anything else -- an unknown attribute of this torch build, say -- must not cost the run its bench line, but a rank
        # that stopped partway through the preflight's collectives is out of step with the others: all ranks agree (MAX of
        # their flags) and, if any of them failed, all of them start over from a barrier instead of entering the job's
        # collectives one call apart.
        pf_flag = torch.tensor([1.0 if pf_err is not None else 0.0], dtype=torch.float64,
                               device=dev if args.backend == "nccl" else "cpu")
        try:
            dist.all_reduce(pf_flag, op=dist.ReduceOp.MAX)
        except Exception as exc:
            raise SystemExit("bench.py: rank %d: the ranks are out of step after the preflight (%r; first error: %r)" % (rank, exc, pf_err))
        if pf_err is not None:
            sys.stderr.write("rccl_preflight: rank %d: %r (the run goes on)\n" % (rank, pf_err))
            sys.stderr.flush()
        if pf_flag.item():
            dist.barrier()
    n_eff = n_sites - n_sites % W["block"]
    if by_sites:
        # contiguous site ranges, whole bootstrap blocks and whole 16-site groups per rank
        unit = int(np.lcm(16, W["block"]))
        n_units = n_eff // unit
        lo = (n_units * rank // world) * unit
        hi = (n_units * (rank + 1) // world) * unit if rank + 1 < world else n_sites
        eng = N.Engine(n_ind, hi - lo, indep_geno=W["indep"], kernel=kernel, device=local_rank, pairwise_del=pdel,
                       exact_shapes=args.exact_shapes, single_image=args.single_image, second_image_bytes=int(args.second_image_gb * 2**30) if args.single_image == 1 else 0)
        eng.synth_fill(W["seed"], args.miss_frac, site0=lo)
        blk_lo, blk_hi = lo // W["block"], min(hi, n_eff) // W["block"]
    else:
        lo, hi = 0, n_sites
        eng = N.Engine(n_ind, n_sites, indep_geno=W["indep"], kernel=kernel, device=local_rank, pairwise_del=pdel,
                       shard_rank=0 if by_reps else rank, shard_world=1 if by_reps else world, exact_shapes=args.exact_shapes, single_image=args.single_image, second_image_bytes=int(args.second_image_gb * 2**30) if args.single_image == 1 else 0)
        eng.synth_fill(W["seed"], args.miss_frac)
    if args.single_image == 1 and args.single_image_gb > 0:
        eng.set_option("single_image_bytes", int(args.single_image_gb * 1e9))
    for name, v in W.get("options", {}).items():
        eng.set_option(name, v)
    for kv in args.set_option:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    # what the engine holds: 3 = both operand images, 2 = ONE in congruent coordinates (+ the fix-up pass of nearly identical
    # pairs when `fixup`), 1 = one + the other formed a range at a time; 0 = not the MFMA kernel
    image_mode, has_fixup = eng.image_mode()

    torch.set_num_threads(1)  # no CPU tensor math here; keep OpenMP's spinning workers out of the way

    # bootstrap workloads: replicate r>0 draws its block map from the reference's taus stream
    rng = N.Taus(12345)
    maps = [None] + [rng.block_map(n_eff // W["block"]) for _ in range(W["n_boot"])]
    # site sharding: a rank needs the multiplicities of ITS blocks (ngd_run_mult)
    mults = [None if m is None else np.bincount(m.astype(np.int64), minlength=n_eff // W["block"]).astype(np.uint32)
             for m in maps]

    acc_ms, red_ms, tot_ms, pair_sites = [], [], [], []
    spill_t = []  # emboot: the accumulation phase kernel by kernel, per timed step (ngd_last_spill_timing)
    last = {}
    on_gpu = args.backend == "nccl"  # collectives on device tensors (RCCL) or, in the rehearsal, on host tensors (gloo)
    pin = lambda *shape: torch.empty(*shape, dtype=torch.float64).pin_memory()

    # Bootstrap workloads: all replicates of the job go to the engine in ONE call (ngd_run_mult_batch: per-block
    # partial sums once, then every replicate is a weighted reduction of them).  When the blocks cover the
    # whole data set the full-data matrix is the all-ones row of the same batch, otherwise it is its own pass.
    batched = W["n_boot"] > 0 and not by_reps
    # a small job's tail (cfg 2: 19 900 cells, ~0.1 ms) is shorter than the hand-over to a worker thread: it stays serial.
    # N > 1: the pipelined form (collectives on the worker thread) has only been rehearsed over gloo -- no multi-GPU
    # hardware was available to this build -- so the default there is the plain form, every collective on the main thread
    # the headline region is always serial (one job's latency); the second region pipelines the jobs two deep
    run_pipelined = not by_reps and not args.serial_tail and (
        args.pipelined_tail or (n_pairs * (W["n_boot"] + 1) >= 100_000 and world == 1))
    mode = {"serial": True}
    sums_seen = {"serial": [], "pipelined": []}  # --vary_jobs: one checksum per step and region
    if by_reps:
        d_sum = torch.zeros(n_pairs, dtype=torch.float64, device=dev)
        d_cnt = torch.zeros(n_pairs, dtype=torch.int64, device=dev)
        h_sum = pin(n_pairs)
        d_dist = torch.zeros(n_pairs, dtype=torch.float64, device=dev)
        h_dist = pin(n_pairs)
        d_all = torch.zeros((n_mat, n_pairs), dtype=torch.float64, device=dev)
        h_all = pin(n_mat, n_pairs)
        cnt_mine = np.full(n_pairs, n_sites if rank == 0 else n_eff, dtype=np.uint64)
    else:
        # the job's cells, flat: [n_mat][n_pairs], padded to `world` equal shares
        total = n_mat * n_pairs
        chunk, c_lo, c_hi = share_of(total, rank, world)
        # A small single-GPU job (cfg 2: 19 900 cells) has its results written straight into pinned host memory, which
        # HIP maps into the device's address space: the reduction kernel's stores cross PCIe while it runs (160 KB), and
        # the separate device-to-host copy (one more launch and one more wait per 0.3 ms job) is gone.
        zero_copy = world == 1 and (total * 8 <= (1 << 20) or args.host_results) and not pdel
        if zero_copy:
            d_flat = torch.zeros(world * chunk, dtype=torch.float64).pin_memory()
            d_cflat = torch.zeros(world * chunk, dtype=torch.int64).pin_memory()
        else:
            d_flat = torch.zeros(world * chunk, dtype=torch.float64, device=dev)  # the engine's (partial) sums
            # the engine's valid-site counts (ngsDist.cpp:362): per rank its own sites' / its own pairs' share
            d_cflat = torch.zeros(world * chunk, dtype=torch.int64, device=dev)
        d_all_1 = d_flat[:total].view(n_mat, n_pairs)
        d_call_1 = d_cflat[:total].view(n_mat, n_pairs)
        # without --pairwise_del a cell's count is the number of sites its matrix visits: no exchange needed
        cnt_flat = np.full((n_mat, n_pairs), n_eff, dtype=np.uint64)
        cnt_flat[0, :] = n_sites
        cnt_flat = cnt_flat.reshape(-1)
        if pdel:
            h_call = torch.empty(total, dtype=torch.int64).pin_memory()
            if world > 1:
                d_cmine = torch.empty(chunk, dtype=torch.int64, device=dev)
                h_cmine = torch.empty(chunk, dtype=torch.int64).pin_memory()
                h_cflat = torch.empty(world * chunk, dtype=torch.int64).pin_memory() if not on_gpu else None
        # Jobs are pipelined two deep: while a worker thread runs job k's tail -- N = 1: copy out + ngd_finish on the
        # host's libm; N > 1: reduce-scatter, every rank's share of ngd_finish, all-gather -- this thread hands the
        # GPU job k+1, which accumulates into the other set of buffers.  One worker, so the ranks' collectives keep
        # one order.
        import concurrent.futures
        tail_pool = concurrent.futures.ThreadPoolExecutor(1, initializer=lambda: torch.cuda.set_device(local_rank))
        fin_pool = concurrent.futures.ThreadPoolExecutor(1)  # (the one call of ngd_finish_stream of a job's tail)
        tail_job = [None, None]
        step_no = [0]
        d_flat_b = [d_flat, torch.zeros_like(d_flat).pin_memory() if zero_copy else torch.zeros_like(d_flat)]
        d_cflat_b = [d_cflat, torch.zeros_like(d_cflat).pin_memory() if zero_copy else torch.zeros_like(d_cflat)]
        if by_pairs:
            own_idx, own_cap = owned_cells(n_ind, n_mat, world)
            n_own = len(own_idx[rank])
            d_own_idx = torch.from_numpy(own_idx[rank]).to(dev)
            d_own = torch.zeros(own_cap, dtype=torch.float64, device=dev)
            h_own, h_own_dist = pin(own_cap), pin(own_cap)
            h_own_dist.zero_()
            d_own_cnt = torch.zeros(own_cap, dtype=torch.int64, device=dev)
            h_own_cnt = torch.zeros(own_cap, dtype=torch.int64).pin_memory()
            d_own_dist = torch.zeros(own_cap, dtype=torch.float64, device=dev)
            d_gath = torch.zeros(world * own_cap, dtype=torch.float64, device=dev)
            h_gath = pin(world * own_cap)
            dist_full = np.zeros(total)
        if world > 1:
            d_mine = torch.empty(chunk, dtype=torch.float64, device=dev)
            h_mine, h_dist_mine = pin(chunk), pin(chunk)
            h_dist_mine.zero_()
            d_dist_mine = torch.empty(chunk, dtype=torch.float64, device=dev)
            d_dist_all = torch.empty(world * chunk, dtype=torch.float64, device=dev)
            h_dist_all = pin(world * chunk)
            h_flat = pin(world * chunk) if not on_gpu else None
        else:
            h_all_b = [pin(n_mat, n_pairs), pin(n_mat, n_pairs)]
            h_call_b = [h_call, torch.empty(total, dtype=torch.int64).pin_memory()] if pdel else [None, None]
            dist_all = np.zeros((n_mat, n_pairs))
            copy_stream = torch.cuda.Stream()
            step_m = max(1, n_mat // 8)
            chunks = [(a, min(n_mat, a + step_m)) for a in range(0, n_mat, step_m)]
            chunk_ev = [[torch.cuda.Event() for _ in chunks] for _ in range(2)]
    if batched:
        n_blocks = n_eff // W["block"]
        fold0 = n_eff == n_sites
        rows = ([np.ones(n_blocks, dtype=np.uint32)] if fold0 else []) + mults[1:]
        mult_all = np.ascontiguousarray(np.stack(rows)[:, blk_lo:blk_hi] if by_sites else np.stack(rows))
    one_call = (batched and world == 1 and not by_reps and fold0 and not zero_copy and not args.split_tail and not args.vary_jobs)
    if one_call:
        import ctypes as C
        assert mult_all.shape == (n_mat, n_blocks) and mult_all.dtype == np.uint32 and dist_all.flags.c_contiguous
        one_args = (eng._h, mult_all.ctypes.data_as(C.POINTER(C.c_uint32)), n_mat, n_blocks, int(W["block"]), 0, int(W["evol_model"]),
                    dist_all.ctypes.data_as(C.POINTER(C.c_double)), mult_all, dist_all)

    # --vary_jobs: the other job (odd steps counted from the END of a region, so that a region's last step is the
    # workload's own job and the spot check below applies to it)
    if args.vary_jobs:
        if by_reps:
            raise SystemExit("bench.py: --vary_jobs is wired for the sharded and single-GPU flows, not --shard replicates")
        if batched:
            keep = 1 if fold0 else 0
            mult_alt = np.ascontiguousarray(np.concatenate([mult_all[:keep], mult_all[keep:][::-1]]))
        else:
            alt_B = 16
            alt_rng = N.Taus(777)
            alt_map = alt_rng.block_map(n_sites // alt_B)
            alt_mult_all = np.bincount(alt_map.astype(np.int64), minlength=n_sites // alt_B).astype(np.uint32)
            alt_mult = np.ascontiguousarray(alt_mult_all[lo // alt_B:(hi if hi < n_sites else n_sites) // alt_B])
            cnt_flat_alt = np.full(n_pairs, (n_sites // alt_B) * alt_B, dtype=np.uint64)

    def checksum(a):
        v = np.ascontiguousarray(a).reshape(-1).view(np.uint64)
        return int(np.bitwise_xor.reduce(v[::max(1, v.size // 65536)]))

    # (the harness's own cost sits inside the timed region: a 0.35 ms job -- cfg 2 -- notices every dict and every ctypes
    # argument conversion, so the per-step calls go straight to the C ABI with arguments built once)
    import ctypes as C
    from ngsdist_amd import _lib as _L
    L_abi = _L.load()
    t_last = _L.NgdTiming()
    t_last_ref = C.byref(t_last)
    may_spill = (not W["indep"]) and W["n_boot"] > 0  # the only plan that fills ngd_last_spill_timing

    def record_timing():
        if L_abi.ngd_last_timing(eng._h, t_last_ref) != 0:
            raise RuntimeError("ngd_last_timing failed")
        red_ms.append(t_last.ms_reduce); tot_ms.append(t_last.ms_total)
        if t_last.launches:  # replicates served from cached block partial sums launch no accumulation
            acc_ms.append(t_last.ms_accum / t_last.launches); pair_sites.append(t_last.pair_sites / t_last.launches)
        if may_spill:
            sp = eng.spill_timing()
            if sp["chunks"]:
                spill_t.append(sp)

    finish_args = {}  # (buffer set, chunk, which counts) -> the ctypes arguments of ngd_finish for that chunk

    def finish_chunk(key, sums, cnts, out):
        a = finish_args.get(key)
        if a is None:
            dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint64)
            assert sums.dtype == np.float64 and cnts.dtype == np.uint64 and out.dtype == np.float64
            assert sums.flags.c_contiguous and cnts.flags.c_contiguous and out.flags.c_contiguous and sums.size == cnts.size == out.size
            a = finish_args[key] = (sums.ctypes.data_as(dp), cnts.ctypes.data_as(up), sums.size, 0, int(W["evol_model"]),
                                    out.ctypes.data_as(dp), sums, cnts, out)  # (the arrays are kept alive beside their pointers)
        if L_abi.ngd_finish(*a[:6]) != 0:
            raise RuntimeError("ngd_finish failed")

    def step(record, variant=0):
        eng.drop_caches()  # bootstrap block partial sums are recomputed in every step (no carried work)
        if by_reps:
            # this rank's matrix, start to finish; then ONE collective puts the N finished matrices together
            eng.run_device(d_sum.data_ptr(), d_cnt.data_ptr(), maps[rank], W["block"])
            if record:
                record_timing()
            h_sum.copy_(d_sum, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            with np.errstate(all="ignore"):
                N.finish(h_sum.numpy(), cnt_mine, 0, W["evol_model"], out=h_dist.numpy())
            if on_gpu:
                d_dist.copy_(h_dist, non_blocking=True)
                gather_matrices(d_all, d_dist)
                if rank == 0:
                    h_all.copy_(d_all, non_blocking=True)
                torch.cuda.current_stream().synchronize()  # every rank: the collective is over before the next step
            else:
                gather_matrices(h_all, h_dist)
            last["dist"] = h_all[-1].numpy()
            return
        # the job's (partial) sums, every matrix: one engine call per plan
        buf = step_no[0] & 1
        step_no[0] += 1
        if tail_job[buf] is not None:  # the job that used this set of buffers two steps ago has left them
            tail_job[buf].result()
        da, dc = d_flat_b[buf][:total].view(n_mat, n_pairs), d_cflat_b[buf][:total].view(n_mat, n_pairs)
        if one_call and mode["serial"] and not variant:
            # the job and its tail in ONE call of the C ABI: a group of replicates leaves the device as soon as it is reduced
            # and the host's threads finish each chunk as it lands (engine.hip run_dist)
            if L_abi.ngd_run_mult_batch_dist(*one_args[:8]) != 0:
                raise RuntimeError("ngd_run_mult_batch_dist: " + L_abi.ngd_last_error().decode())
            if record:
                record_timing()
            last["dist"] = dist_all[-1]
            return
        if batched:
            first = 0 if fold0 else 1
            if not fold0:
                eng.run_device(da[0].data_ptr(), dc[0].data_ptr())
                if record:
                    record_timing()
            eng.run_batch(mult=mult_alt if variant else mult_all, block_size=W["block"], d_sum_ptr=da[first].data_ptr(),
                          d_cnt_ptr=dc[first].data_ptr())
        elif variant:
            eng.run_mult(alt_mult, alt_B, d_sum_ptr=da[0].data_ptr(), d_cnt_ptr=dc[0].data_ptr())
        else:
            eng.run_device(da[0].data_ptr(), dc[0].data_ptr())
        if record:
            record_timing()
        region = "serial" if mode["serial"] else "pipelined"
        cnt_job = cnt_flat_alt if (variant and not batched) else cnt_flat
        if world == 1:
            # results leave the device in chunks of matrices (the engine call has synchronised its stream); the host
            # tail (ngd_finish) of one chunk runs while the next is in flight -- on the worker thread, so that this
            # thread can hand the GPU the next job meanwhile
            ha, h_call1, evs = (da if zero_copy else h_all_b[buf]), h_call_b[buf], chunk_ev[buf]
            if not zero_copy:
                with torch.cuda.stream(copy_stream):
                    if pdel:
                        h_call1.copy_(d_cflat_b[buf][:total], non_blocking=True)
                    for c, (a, b) in enumerate(chunks):
                        ha[a:b].copy_(da[a:b], non_blocking=True)
                        evs[c].record(copy_stream)

            def tail():
                cnts = h_call1.numpy().view(np.uint64) if pdel else cnt_job
                which = 2 if pdel else (0 if cnts is cnt_flat else 1)  # (1: --vary_jobs' other job)
                if not zero_copy and total >= (1 << 20):
                    # a job of millions of cells: ONE call of ngd_finish_stream on a helper thread works the cells as the
                    # chunks land (this thread waits for the copies' events and raises the counter): one wake-up of the
                    # host's threads per job instead of one per chunk
                    key = (buf, "stream", which)
                    a_ = finish_args.get(key)
                    if a_ is None:
                        dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint64)
                        sums, out = ha.numpy().reshape(-1)[:total], dist_all.reshape(-1)
                        assert sums.flags.c_contiguous and cnts.flags.c_contiguous and out.flags.c_contiguous and cnts.size == total
                        landed = C.c_uint64(0)
                        a_ = finish_args[key] = (sums.ctypes.data_as(dp), cnts.ctypes.data_as(up), total, 0, int(W["evol_model"]),
                                                 out.ctypes.data_as(dp), C.byref(landed), landed, sums, cnts, out)
                    landed = a_[7]
                    landed.value = 0
                    fut = fin_pool.submit(L_abi.ngd_finish_stream, *a_[:7])
                    for c, (a, b) in enumerate(chunks):
                        evs[c].synchronize()
                        landed.value = b * n_pairs
                    if fut.result() != 0:
                        raise RuntimeError("ngd_finish_stream failed")
                else:
                    for c, (a, b) in enumerate(chunks):
                        if not zero_copy:
                            evs[c].synchronize()
                        key = (buf, c, which)
                        if key in finish_args:
                            finish_chunk(key, None, None, None)
                        else:
                            finish_chunk(key, ha[a:b].numpy().reshape(-1), cnts[a * n_pairs:b * n_pairs], dist_all[a:b].reshape(-1))
                last["dist"] = dist_all[-1]
                if args.vary_jobs:
                    sums_seen[region].append(checksum(dist_all))

            if mode["serial"]:
                tail()
            else:
                tail_job[buf] = tail_pool.submit(tail)
            return
        # N > 1: reduce-scatter (partial sums add / disjoint shards meet) -> every rank finishes its share of the
        # cells on its own host cores -> all-gather of the finished cells.  (The engine call has synchronised its
        # stream: this job's sums are in d_flat_b[buf]; the next job writes the other set.)
        df, dcf = d_flat_b[buf], d_cflat_b[buf]

        def tail_pairs():
            # this rank's own cells only: pack -> D2H -> gen_dist()'s tail on this host -> H2D -> ONE all-gather
            torch.index_select(df, 0, d_own_idx, out=d_own[:n_own])
            h_own.copy_(d_own, non_blocking=True)
            if pdel:
                torch.index_select(dcf, 0, d_own_idx, out=d_own_cnt[:n_own])
                h_own_cnt.copy_(d_own_cnt, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            cnts = h_own_cnt.numpy().view(np.uint64)[:n_own] if pdel else cnt_job[own_idx[rank]]
            with np.errstate(all="ignore"):
                N.finish(h_own.numpy()[:n_own], cnts, 0, W["evol_model"], out=h_own_dist.numpy()[:n_own])
            if on_gpu:
                d_own_dist.copy_(h_own_dist, non_blocking=True)
                gather_cells(d_gath, d_own_dist)
                if rank == 0:
                    h_gath.copy_(d_gath, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            else:
                gather_cells(h_gath, h_own_dist)
            if rank == 0:
                unpack_cells(h_gath.numpy().reshape(world, own_cap), own_idx, dist_full)
                last["dist"] = dist_full.reshape(n_mat, n_pairs)[-1]
                if args.vary_jobs:
                    sums_seen[region].append(checksum(dist_full))

        def tail_n():
            if on_gpu:
                scatter_sum(df, d_mine)
                h_mine.copy_(d_mine, non_blocking=True)
                if pdel:  # --pairwise_del: the valid-site counts of the ranks' site ranges add up the same way
                    scatter_sum(dcf, d_cmine)
                    h_cmine.copy_(d_cmine, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            else:
                h_flat.copy_(df)
                scatter_sum(h_flat, h_mine)
                if pdel:
                    h_cflat.copy_(dcf)
                    scatter_sum(h_cflat, h_cmine)
            cnts = h_cmine.numpy().view(np.uint64)[:c_hi - c_lo] if pdel else cnt_job[c_lo:c_hi]
            with np.errstate(all="ignore"):
                N.finish(h_mine.numpy()[:c_hi - c_lo], cnts, 0, W["evol_model"],
                         out=h_dist_mine.numpy()[:c_hi - c_lo])
            if on_gpu:
                d_dist_mine.copy_(h_dist_mine, non_blocking=True)
                gather_cells(d_dist_all, d_dist_mine)
                if rank == 0:
                    h_dist_all.copy_(d_dist_all, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            else:
                gather_cells(h_dist_all, h_dist_mine)
            last["dist"] = h_dist_all.numpy()[:total].reshape(n_mat, n_pairs)[-1]
            if args.vary_jobs and rank == 0:
                sums_seen[region].append(checksum(h_dist_all.numpy()[:total]))

        the_tail = tail_pairs if by_pairs else tail_n
        if mode["serial"]:
            the_tail()
        else:
            tail_job[buf] = tail_pool.submit(the_tail)

    def fence():
        if not by_reps:
            for j in tail_job:  # every job handed in so far is finished, tail included
                if j is not None:
                    j.result()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allred(x, op):
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())

    if world > 1:  # communicators and buffers come up outside the timed region whatever --warmup is
        if by_reps:
            gather_matrices(d_all if on_gpu else h_all, d_dist if on_gpu else h_dist)
        else:
            def comms_up():  # on the worker: it is the thread that issues the timed collectives
                if by_pairs:
                    gather_cells(d_gath if on_gpu else h_gath, d_own_dist if on_gpu else h_own_dist)
                    if on_gpu:
                        torch.cuda.current_stream().synchronize()
                    return
                scatter_sum(d_flat if on_gpu else h_flat, d_mine if on_gpu else h_mine)
                if pdel:
                    scatter_sum(d_cflat if on_gpu else h_cflat, d_cmine if on_gpu else h_cmine)
                gather_cells(d_dist_all if on_gpu else h_dist_all, d_dist_mine if on_gpu else h_dist_mine)
                if on_gpu:
                    torch.cuda.current_stream().synchronize()
            comms_up()
            if run_pipelined:  # the worker is the thread that issues the pipelined region's collectives
                tail_pool.submit(comms_up).result()
    # the shader clock while the timed kernels run: the device's own reading (pp_dpm_sclk marks the current level),
    # polled from a thread; the main thread sits in hipStreamSynchronize (GIL released) most of the time
    clk = {"mhz": [], "stop": False, "src": None}

    def poll_clock():
        import glob
        files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        if not files:
            return
        f = files[min(local_rank, len(files) - 1)]
        clk["src"] = f
        while not clk["stop"]:
            try:
                for ln in open(f).read().splitlines():
                    if ln.rstrip().endswith("*"):
                        clk["mhz"].append(float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
            except Exception:
                return
            time.sleep(0.002)

    def alt(k):  # --vary_jobs: which job step k of a region computes
        return (args.steps - 1 - k) & 1 if args.vary_jobs else 0

    for _ in range(args.warmup):
        step(False)
    fence()
    import threading
    poller = threading.Thread(target=poll_clock, daemon=True)
    if rank == 0 and args.poll_sclk:
        poller.start()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(True, alt(k))
    fence()
    dt = time.perf_counter() - t0
    clk["stop"] = True
    # second region: the same K jobs two deep (job k's tail beside job k+1's kernels) -- throughput of a stream of jobs,
    # reported apart; a real run is ONE job and cannot hide its tail behind a next one
    dt_pipe = None
    if run_pipelined:
        mode["serial"] = False
        for _ in range(min(2, args.warmup)):
            step(False)
        fence()
        tp = time.perf_counter()
        for k in range(args.steps):
            step(False, alt(k))
        fence()
        dt_pipe = time.perf_counter() - tp
        mode["serial"] = True
        if world > 1:
            dt_pipe = allred(dt_pipe, dist.ReduceOp.MAX)
    if world > 1:
        dt = allred(dt, dist.ReduceOp.MAX)
        # per-rank accumulation-kernel time: report the slowest rank's mean
        acc_mean_ms = allred(float(np.mean(acc_ms)) if acc_ms else 0.0, dist.ReduceOp.MAX)
        pair_sites_per_launch_all = allred(float(np.mean(pair_sites)) if pair_sites else 0.0, dist.ReduceOp.SUM)
    else:
        acc_mean_ms = float(np.mean(acc_ms))
        pair_sites_per_launch_all = float(np.mean(pair_sites))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt * 1e3 / args.steps
    value = n_pairs * n_mat * args.steps / dt
    # spot check against the CPU oracle: a few pairs over ALL sites of the last matrix computed
    spot = None
    cpu = None
    try:
        from oracle import oracle as O
        # 4 individuals -> 6 pairs, every site of the LAST matrix of the step (full data or last replicate)
        idx = [0, 1, n_ind // 2, n_ind - 1]
        sub = np.concatenate([O.synth_indmajor(W["seed"], n_ind, n_sites, miss_frac=args.miss_frac, i0=i, n_sub=1)
                              for i in idx])
        src = None if maps[-1] is None else O.boot_site_src(maps[-1], W["block"])
        so, co = O.all_pairs(sub, indep_geno=W["indep"], site_src=src, n_sites=n_eff if src is not None else n_sites,
                             n_threads=6, pairwise_del=pdel)
        with np.errstate(all="ignore"):
            do = O.finish(so, co, 0, W["evol_model"])
        k = 0
        worst = 0.0
        for a in range(len(idx)):
            for b in range(a + 1, len(idx)):
                g = last["dist"][N.n_pairs(n_ind) - N.n_pairs(n_ind - idx[a]) + (idx[b] - idx[a] - 1)]
                worst = max(worst, abs(g - do[k]) / abs(do[k]))
                k += 1
        spot = {"pairs": k, "sites": int(n_eff if src is not None else n_sites), "max_rel_err_vs_oracle": float(worst)}
        if not args.no_cpu and world == 1:  # the CPU baseline is an N=1 figure
            hc = host_cpu()
            # every core this process may use (ngsDist.cpp:44-52: --n_threads); the port's pool holds up to 256 threads
            cores = min(args.cpu_threads or hc["threads"], 256)
            rate_guess = (1.7e8 if W["indep"] else 2.8e6) * min(cores, 64)  # pair-sites/s per thread, measured
            # ~15 s of wall time at the measured per-thread rate, the sample held to 2 GB of likelihoods
            cs = args.cpu_sites or int(max(64, min(n_sites, 15.0 * rate_guess / n_pairs, 2e9 / (24.0 * n_ind))))
            pc = O.synth_indmajor(W["seed"], n_ind, cs)
            tc = time.perf_counter()
            O.all_pairs(pc, indep_geno=W["indep"], n_threads=cores)
            tc = time.perf_counter() - tc
            cpu_ps = n_pairs * cs / tc
            cpu = {"value": cpu_ps / n_sites, "unit": "pair-distances/s", "cores": cores, "kind": "port",
                   "threads": cores, "host_cores": hc["host_cores"], "cpu_model": hc["cpu_model"],
                   "affinity_cores": hc["affinity_cores"], "cgroup_cpu_quota": hc["cgroup_cpu_quota"],
                   "pair_sites_per_s": cpu_ps, "seconds": tc,
                   "sample": "first %d of %d sites, all %d pairs, same generator/seed; rate scaled linearly "
                             "in n_sites to one full matrix" % (cs, n_sites, n_pairs)}
            if not W["indep"]:
                # The same threaded pair loop on the reference's OWN em2() (oracle/_ref/libref_em2.so = emOptim2.cpp
                # compiled from the reference tree as it lies, no stand-ins), all host cores, on the first part of the
                # same sample; its sums must carry the bits of the restatement's.
                try:
                    if not O.use_reference_em2(True):
                        raise RuntimeError("oracle/_ref/libref_em2.so not built")
                    try:
                        cr = max(32, cs // 3)
                        pr = np.ascontiguousarray(pc[:, :cr])
                        tr = time.perf_counter()
                        sr, _ = O.all_pairs(pr, indep_geno=False, n_threads=cores)
                        tr = time.perf_counter() - tr
                    finally:
                        O.use_reference_em2(False)
                    sp, _ = O.all_pairs(pr, indep_geno=False, n_threads=cores)
                    ref_ps = n_pairs * cr / tr
                    cpu["reference_em2"] = {
                        "value": ref_ps / n_sites, "unit": "pair-distances/s", "cores": cores, "threads": cores,
                        "host_cores": hc["host_cores"], "cpu_model": hc["cpu_model"], "kind": "reference-em2",
                        "pair_sites_per_s": ref_ps, "seconds": tr, "bit_identical_to_port": bool(np.array_equal(sr, sp)),
                        "sample": "first %d of %d sites, all %d pairs: gen_dist's loop as restated in oracle/, em2() the "
                                  "reference's own (emOptim2.cpp:112-135); rate scaled linearly in n_sites"
                                  % (cr, n_sites, n_pairs)}
                except Exception as exc:
                    cpu["reference_em2"] = {"error": repr(exc)}
    except Exception as exc:  # reported in the line, which is then marked invalid (exit code 1)
        cpu = {"error": repr(exc)}
    valid = bool(spot is not None and spot["max_rel_err_vs_oracle"] <= 1e-9)

    t_acc = acc_mean_ms * 1e-3
    if W["indep"]:
        flops = FLOPS_PER_PAIR_SITE * pair_sites_per_launch_all / world  # per launch of ONE rank's kernel
        roof = {"bound": "mfma", "kernel": "k_accum_%s" % kernel, "achieved": flops / t_acc / 1e12,
                "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": flops / t_acc / 1e12 / PEAK_FP64_TFLOPS,
                "traffic": None, "ms_per_launch": acc_mean_ms,
                "algorithmic": "%.0f FP64 flop per pair-site x %.4g pair-sites per launch"
                               % (FLOPS_PER_PAIR_SITE, pair_sites_per_launch_all / world)}
        if image_mode == 1:
            roof["note"] = ("single-image engine: a pass is one launch of the kernel per range of the second operand image "
                            "plus the kernel that forms the range (k_qb_range, HBM-bound); ms_per_launch is the whole "
                            "accumulation phase of a pass, frac the pass's flops against it")
        sb = BYTES_PER_PAIR_SITE * pair_sites_per_launch_all / world / t_acc / 1e9
        roof["stream_model"] = {"bound": "hbm", "achieved": sb, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": sb / PEAK_HBM_GBS,
                                "algorithmic": "48 B per pair-site (north_star's one-wavefront-per-pair model)"}
        if kernel == "stream":
            # the 48 B model counts every re-read of an individual; all but one per tile of L2 are served on-die, so
            # this figure can exceed 1 -- it is the north_star's roofline definition, not HBM utilisation (that is
            # `traffic` / time, from the PMC pass)
            roof = dict(roof["stream_model"], kernel="k_accum_stream", traffic=None, ms_per_launch=acc_mean_ms,
                        note="algorithmic bytes (48 B per pair-site), mostly L2 / Infinity-Cache served; see traffic")
    else:
        # EM path: bound by FP64 VALU issue (HBM is irrelevant: 48 B per ~130 instructions).  The roof is the FP64 vector
        # pipe, 78.6 TFLOP/s = 39.3e12 lane-instruction slots/s x 2 flop (the same datapath and peak as FP64 MFMA,
        # profiles/r01_fp64_peak_microbench.txt).  Algorithmic work = VALU lane-instructions per pair-site, counted in
        # the ISA: the data-dependent factor is the number of EM steps, which the table kernel
        # reports as table rounds per (tile, site) and which is fixed at its measured mean for the per-pair kernels.
        ps_launch = pair_sites_per_launch_all / world
        lane_peak = PEAK_FP64_TFLOPS * 1e12 / 2  # FP64 lane-instruction slots per second (1 slot = 1 FMA = 2 flop)
        spill = None
        if spill_t:
            # the spilled-terms plan: the dominant kernel is the EM pass itself (k_accum_em_table<SPILL>, one launch per chunk
            # of sites), priced like cfg 4's; the contraction that follows every chunk has its own roofline below
            spill = {k: float(np.mean([x[k] for x in spill_t])) for k in spill_t[0]}
            acc_job_ms = acc_mean_ms
            acc_mean_ms = spill["ms_terms"] / spill["chunks"]
            t_acc = spill["ms_terms"] * 1e-3
        roof = {"bound": "valu", "kernel": "k_accum_%s" % kernel, "achieved": None, "peak": lane_peak,
                "unit": "lane-instructions/s", "frac": None, "traffic": None, "ms_per_launch": acc_mean_ms,
                "pair_sites_per_s": ps_launch / t_acc,
                "algorithmic": "VALU lane-instructions the kernel EXECUTES on active lanes (v_cmpx, v_add_u32 and moves "
                               "included: not a flop count) against the FP64 vector pipe's issue rate, 39.3e12 lane slots/s "
                               "(16 lanes per SIMD and cycle; the 78.6 TFLOP/s FP64 peak is this x 2 flop per FMA)"}
        # Work per pair-site of THIS kernel on THIS data, from the PMC pass of tools/em_pmc.sh (accepted only if it was
        # measured on the kernel source this library was built from): lanes that executed a VALU instruction
        # (SQ_THREAD_CYCLES_VALU) and issue slots taken (SQ_INSTS_VALU x 64).  `frac` is the ACTIVE-LANE figure: it cannot
        # be raised by issuing instructions whose lanes are masked off; the issue-slot figure is reported beside it.
        try:
            import hashlib
            vj = json.load(open(os.path.join(ROOT, "profiles", "valu_%s_%s.json" % (args.workload, kernel))))
            src = "accum_%s.hip" % {"em_fast": "em", "em_faithful": "em", "em_table": "em_table"}.get(kernel, kernel)
            now = hashlib.sha256(open(os.path.join(ROOT, "ngsdist_amd", "csrc", src), "rb").read()).hexdigest()[:16]
            stale = vj.get("kernel_source_sha16", {}).get(src) != now
            act, iss = vj["per_pair_site"]["active_lane_instructions"], vj["per_pair_site"]["issue_slots"]
            roof["achieved"] = act * ps_launch / t_acc
            roof["frac"] = roof["active_lane_frac"] = act * ps_launch / t_acc / lane_peak
            roof["issue_slot_frac"] = iss * ps_launch / t_acc / lane_peak
            roof["frac_kind"] = ("active lanes: %.1f executed lane-instructions per pair-site (SQ_THREAD_CYCLES_VALU) x pair-sites "
                                 "/ launch time / %.3g slots per second; issue_slot_frac prices the %.1f slots the kernel "
                                 "ISSUES per pair-site (SQ_INSTS_VALU x 64), masked lanes included" % (act, lane_peak, iss))
            roof["valu_source"] = "profiles/valu_%s_%s.json (%s)" % (args.workload, kernel, vj.get("source"))
            if stale:
                roof["valu_stale"] = "measured on another version of %s" % src
        except Exception as exc:
            roof["frac_kind"] = "no PMC pass of this kernel under profiles/ (%r)" % (exc,)
        if spill:
            # k_contract_mfma: running sums D[matrix][pair slot] += W[matrix][unit] x C[unit][pair slot] per chunk.  Algorithmic
            # bytes: every term read once per batch of 128 matrices, the running sums read and written once per chunk, the
            # weights; algorithmic flops: 2 per (matrix of the padded groups of 16, pair slot, unit).  Whichever of the two
            # takes longer at its peak is the bound: HBM up to ~32 matrices, the FP64 matrix pipe above.
            mg, sg, sgl, un, ch = (spill[k] for k in ("matrix_groups", "slot_groups", "slot_groups_live", "units", "chunks"))
            # `frac` is on USEFUL work -- the job's real matrices x real pairs -- like every other frac of this file; what the
            # kernel ISSUES (matrices padded to groups of 16, pair slots to groups of 16 and to the wavefronts' 2 or 4 groups) is
            # printed beside it as issued_frac.
            term_bytes = un * sgl * 16 * 8
            i_bytes = term_bytes * np.ceil(mg / 8) + 2 * 8 * mg * sg * 256 * ch + (un + 4 * ch) * mg * 16 * 8
            i_flops = 2.0 * (16 * mg) * (16 * sg) * un
            c_bytes = un * n_pairs * 8 * np.ceil(n_mat / 128) + 2 * 8 * n_mat * n_pairs * ch + (un + 4 * ch) * n_mat * 8
            c_flops = 2.0 * n_mat * n_pairs * un
            t_c = spill["ms_contract"] * 1e-3
            c_hbm, c_mf = c_bytes / t_c / 1e9, c_flops / t_c / 1e12
            i_hbm, i_mf = i_bytes / t_c / 1e9, i_flops / t_c / 1e12
            by_hbm = c_hbm / PEAK_HBM_GBS >= c_mf / PEAK_FP64_TFLOPS
            roof["ms_per_job"] = spill["ms_terms"]
            roof["launches_per_job"] = ch
            roof["instantiation"] = "k_accum_em_table<8, 16, 4, false, PDEL, true, 1, SPILL = true>"
            roof["terms_written_bytes_per_job"] = term_bytes
            roof["spill"] = dict(spill, ms_accumulation_phase=acc_job_ms,
                                 note="ms_*: HIP events on the engine's stream around each kernel, summed over the job's chunks, "
                                      "mean over the timed steps; a unit = unit_sites consecutive sites of one bootstrap block, "
                                      "whose terms are added up before they leave the EM kernel")
            roof["contract"] = {
                "kernel": "k_contract_mfma", "bound": "hbm" if by_hbm else "mfma",
                "achieved": c_hbm if by_hbm else c_mf, "peak": PEAK_HBM_GBS if by_hbm else PEAK_FP64_TFLOPS,
                "unit": "GB/s" if by_hbm else "TFLOP/s",
                "frac": c_hbm / PEAK_HBM_GBS if by_hbm else c_mf / PEAK_FP64_TFLOPS,
                "issued_frac": i_hbm / PEAK_HBM_GBS if by_hbm else i_mf / PEAK_FP64_TFLOPS,
                "hbm": {"achieved": c_hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": c_hbm / PEAK_HBM_GBS,
                        "issued_frac": i_hbm / PEAK_HBM_GBS},
                "mfma": {"achieved": c_mf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": c_mf / PEAK_FP64_TFLOPS,
                         "issued_frac": i_mf / PEAK_FP64_TFLOPS},
                "ms_per_job": spill["ms_contract"], "ms_per_launch": spill["ms_contract"] / spill["contract_launches"],
                "traffic": None,
                "algorithmic": "USEFUL work: %.4g B per job (%d units x %d pairs x 8 B of terms read once per 128 matrices, the "
                               "running sums of %d matrices x %d pairs read and written once per chunk (%d), the weights); %.4g "
                               "flop = 2 x %d matrices x %d pairs x %d units.  ISSUED (issued_frac): %.4g B, %.4g flop = 2 x %d "
                               "matrices (padded to 16s) x %d pair slots x %d units"
                               % (c_bytes, un, n_pairs, n_mat, n_pairs, ch, c_flops, n_mat, n_pairs, un, i_bytes, i_flops,
                                  16 * mg, 16 * sg, un)}
        if kernel == "em_table":
            tile_sites, rounds = eng.em_work()
            roof["table_rounds_per_tile_site"] = rounds / max(1, tile_sites)
            # What the closed form NEEDS per pair-site on this data, so that the ~93 lane-instructions the kernel executes have
            # a denominator (emOptim2.cpp:112-135 in the closed form of DESIGN.md section 3 K2): the search for the step T at
            # which the reference's rule stops (:127) is one compare + one count per step and cannot skip steps (the rule takes
            # the FIRST step that satisfies it, and R_t(i1) R_t(i2) is not monotone in t in general); the site's term is three
            # products and their sum against the per-individual g_T (ngsDist.cpp:351-353); the tables belong to ONE individual
            # and need building once per individual and site (the kernel rebuilds them per 64 x 64 tile of pairs: n_ind / 64
            # times), ~35 lane-instructions per step (three powers, their sum, R_t with one division, f_t, g_t = score . f_t, Q_t).
            rounds_mean = rounds / max(1, tile_sites)
            alg = 2 * MEAN_EM_STEPS + 4 + 35.0 * 16 * rounds_mean * n_ind / n_pairs
            roof["algorithmic_lane_instructions_per_pair_site"] = alg
            roof["algorithmic_frac"] = alg * ps_launch / t_acc / lane_peak
            roof["algorithmic_model"] = (
                "2 x %.1f (one compare + one count per EM step until the rule of emOptim2.cpp:127 stops; mean steps per pair-site "
                "on this data) + 4 (the three products and the sum of ngsDist.cpp:351-353 against g_T) + %.2f (tables of 16 steps x "
                "%.2f rounds, ~35 lane-instructions a step, built ONCE per individual and site and shared by its %d pairs); the "
                "kernel executes ~3 x this: it rebuilds the tables per 64 x 64 tile (%.1f tiles per individual), searches in "
                "blocks of 8 steps, and moves thresholds and operands through LDS"
                % (MEAN_EM_STEPS, 35.0 * 16 * rounds_mean * n_ind / n_pairs, rounds_mean, n_ind - 1, n_ind / 64.0))
        # SURVEY 8(d)'s algorithmic count of the reference's form (emOptim2.cpp:77-109: per EM step 36 mul + 37 add + 18 div
        # + 1 log, + the first lik2 and the 18-flop scoring per site), at this data set's mean of 11.7 steps per pair-site
        ref_flop = MEAN_EM_STEPS * (36 + 37 + 18 + 1) + (27 + 1) + 18
        roof["reference_form_model"] = {"flop_per_pair_site": ref_flop, "achieved_TFLOPs": ref_flop * ps_launch / t_acc / 1e12,
                                        "note": "what the reference's arithmetic would need at this rate; the kernel's closed "
                                                "form and per-individual tables remove most of it (DESIGN.md section 3, K2)"}
        # the same launch priced with the per-pair algorithm's count (round 1's kernel): > 1 means work removed
        roof["per_pair_model"] = {"lane_instructions_per_pair_site": EMFAST_OPS_PER_PAIR_SITE,
                                  "frac": EMFAST_OPS_PER_PAIR_SITE * ps_launch / t_acc / lane_peak}

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure is the one
    # the last tools/profile.sh run of this same command left in profiles/ -- accepted only if the kernel source it was
    # measured on is the one this library was built from (sha256 recorded by tools/pmc_summary.py)
    try:
        import hashlib
        tname = "traffic_%s_%s.json" % (args.workload, "single_image%d" % image_mode if image_mode in (1, 2) else kernel)
        tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
        src = "accum_%s.hip" % {"em_fast": "em", "em_faithful": "em", "em_table": "em_table"}.get(kernel, kernel)
        now = hashlib.sha256(open(os.path.join(ROOT, "ngsdist_amd", "csrc", src), "rb").read()).hexdigest()[:16]
        if tj.get("kernel_source_sha16", {}).get(src) != now:
            roof["traffic_stale"] = "profiles/%s was measured on another version of %s" % (tname, src)
        else:
            for kname, v in tj["per_launch"].items():
                if "accum" in kname and roof.get("traffic") is None:
                    roof["traffic"] = v.get("read_bytes", 0) + v.get("write_bytes", 0)
                    roof["traffic_source"] = "profiles/%s (%s)" % (tname, tj["source"])
                if "contract" in kname and "contract" in roof:
                    roof["contract"]["traffic"] = v.get("read_bytes", 0) + v.get("write_bytes", 0)
                    roof["contract"]["traffic_source"] = "profiles/%s (%s), per launch" % (tname, tj["source"])
    except Exception:
        pass

    # the dominant kernel's launches inside the timed region (HIP events on the engine's stream) and the shader clock
    # the device reported meanwhile: what a rocprofv3 pass of this command must reproduce (tools/profile.sh keeps its
    # kernel trace only if its own ms_per_step is within 2 % of an unprofiled line of the same lease)
    if spill_t:
        per = [x["ms_terms"] / x["chunks"] for x in spill_t]
        roof["ms_per_launch_min"], roof["ms_per_launch_median"] = float(np.min(per)), float(np.median(per))
        roof["launches_timed"] = int(sum(x["chunks"] for x in spill_t))
    elif acc_ms:
        roof["ms_per_launch_min"], roof["ms_per_launch_median"] = float(np.min(acc_ms)), float(np.median(acc_ms))
        roof["launches_timed"] = len(acc_ms)
    # the clock the dominant kernel ran at, sampled INSIDE its last launch (one wavefront reads the shader-cycle and the
    # constant-rate counter around its work: ngd_last_shader_clock); the peaks above are quoted at 2400 MHz
    mhz = eng.shader_clock_mhz()
    roof["shader_clock_mhz"] = mhz if mhz > 0 else None
    if mhz > 0 and roof.get("frac") is not None:
        roof["frac_of_peak_at_that_clock"] = roof["frac"] * 2400.0 / mhz
    roof["sclk_mhz"] = ({"median": float(np.median(clk["mhz"])), "min": float(np.min(clk["mhz"])),
                         "max": float(np.max(clk["mhz"])), "samples": len(clk["mhz"]),
                         "source": "%s (the level marked current), polled every 2 ms during the timed region; on some boxes "
                                   "this file reports an idle level throughout -- shader_clock_mhz is the figure to use" % clk["src"]}
                        if clk["mhz"] else None)
    roof["sclk_poller_active_during_value"] = bool(args.poll_sclk)
    pipeline_check = None
    if args.vary_jobs:
        a, b = sums_seen["serial"][-args.steps:], sums_seen["pipelined"][-args.steps:]
        pipeline_check = {"steps_compared": len(b), "jobs_differ": len(set(a)) > 1,
                          "ok": bool(b) and a == b and len(set(a)) > 1}
        valid = valid and (pipeline_check["ok"] or not run_pipelined)

    out = {
        "metric": "pair-distances/sec", "value": value, "unit": "pair-distances/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak" if by_reps or (world == 1 and args.shard == "replicates") else "strong", "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: n_ind=%d n_sites=%d %s evol_model=%d n_boot_rep=%d boot_block_size=%d"
                               % (args.workload, n_ind, n_sites, "--indep_geno" if W["indep"] else "EM",
                                  W["evol_model"], W["n_boot"], W["block"])
                               + (" --pairwise_del (%.3g of the sites missing; not a BASELINE configuration)" % args.miss_frac
                                  if pdel else ""),
                   "kernel": kernel, "matrices_per_step": n_mat, "n_pairs": n_pairs,
                   "device_bytes": eng.device_bytes(), "single_image": args.single_image,
                   "image_mode": {"resolved": image_mode, "fixup_pass": has_fixup, "last_fixup": eng.fixup(),
                                  "meaning": "3 = two operand images, 2 = one image in congruent coordinates (fixup_pass: pairs "
                                             "below 1e-6 per site are recomputed the two-operand way), 1 = one image + the other "
                                             "formed a range at a time, 0 = not the MFMA kernel"},
                   "second_image_gb": args.second_image_gb if args.single_image == 1 else None,
                   "results": ("written by the reduction kernel straight into pinned host memory (mapped into the device's "
                               "address space): no separate copy" if world == 1 and not by_reps and zero_copy else
                               "device buffers, copied to pinned host memory"),
                   "host_tail": ("inside the engine call (ngd_run_mult_batch_dist): a group of 32 replicates is copied out as soon "
                                 "as it is reduced, beside the later groups' reductions, and the host's threads finish each "
                                 "chunk as it lands; " if one_call else "") +
                                "serial: a job's copy-out and ngd_finish (N > 1: the collectives too) end before the next "
                                "job's kernels start -- ms_per_step and value are ONE job's latency",
                   "pair_sites_per_s": n_pairs * float(n_eff if W["n_boot"] else n_sites) * n_mat * args.steps / dt,
                   "sharding": ("site axis split over %d ranks (each holds 1/%d of the data, all pairs): one RCCL "
                                "reduce-scatter adds the sums, every rank finishes its 1/%d of the cells on its host, one "
                                "RCCL all-gather" % (world, world, world)) if by_sites else
                               ("one matrix per GPU (full data + %d bootstrap replicates, block size 1), data set "
                                "resident on every GPU, one RCCL all-gather of the finished matrices" % (world - 1))
                               if by_reps else
                               ("pair tiles dealt over %d rank(s), input replicated; every rank finishes its own cells on "
                                "its host, ONE all-gather of the finished cells" % world)},
        "pipelined": None if dt_pipe is None else {
            "ms_per_step": dt_pipe * 1e3 / args.steps, "value": n_pairs * n_mat * args.steps / dt_pipe,
            "note": "a second region of the same K jobs, two in flight: job k's tail (copy-out and ngd_finish; N > 1: "
                    "reduce-scatter, each rank's share of ngd_finish, all-gather) on a worker thread beside job k+1's "
                    "kernels, every tail inside the region -- throughput of a stream of jobs, not the headline"},
        "roofline": roof, "cpu_baseline": cpu, "spot_check": spot, "valid": valid,
        "pipeline_check": pipeline_check, "device_bytes": eng.device_bytes(),
        "ms_reduce": float(np.mean(red_ms)), "ms_engine_total": float(np.mean(tot_ms)),
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if not valid:
        sys.stderr.write("bench.py: the spot check against the CPU oracle failed or did not run: %r\n" % (spot or cpu,))
        sys.exit(1)


if __name__ == "__main__":
    main()
