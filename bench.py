#!/usr/bin/env python3
"""bench.py -- MC steps/s of the polarizable 4096-atom box on N x MI355X (one walker per GPU).

One "step" = one single-molecule displacement (translate + rotate, reference
src/mc/mc_moves.c:378-488) + one energy() on the device through the C ABI + Metropolis
(reference src/mc/mc.c:294-353), exactly what the reference does per step; the loop itself is the
C host layer of mpmc_amd/host/ (system_t, energy(), checkpoint/make_move/restore), called via ctypes.
energy() is evaluated incrementally -- everything pairwise is resident and only what the moved
molecule touches is recomputed, bit-identical to a full evaluation (the reference caches per pair
too, pairs.c:238-249) -- and `full_rebuild_steps_per_s` in the line is the same chain with every
cached unit rebuilt every step.  Workload: the 4096-atom PCN-61 cell + 416 BSSP H2 of
tests/golden/pcn61_bssp_4096.npz with the flags of the reference's sample_configs_gpu/3_PCN61/iter.inp
run as NVT (Jacobi x4, cutoff 8 A, FH 4th order) -- BASELINE.json configs[3]; `--workload` selects
the synthetic boxes instead.

Walkers are independent: rank r runs its own chain with seed + r on GPU r; the only
collective is the sum of a small observable vector every `corrtime` steps, through the C ABI's RCCL
entry (mpmc_hip_allreduce_observables_begin/_end: what the reference's C mc() would call in place of
MPI_Gather, mc.c:417-432).  torch.distributed is only the launcher's barrier / max-over-ranks clock.

`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks itself (child
processes, before this process has imported torch or touched HIP) and relays rank 0's JSON line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.
`--gpus` must equal WORLD_SIZE.  Prints ONE JSON line on rank 0.

Measurement layout: the timed region carries NO events or probes (value does not depend on how the
roofline is sampled); afterwards, untimed: (1) a roofline pass -- HIP events on the engine's own stream
around every launch of the dominant kernel, raw mean, nothing subtracted; (2) a pass with every kernel
class timed; (3) one full (non-incremental) evaluation for the pairs/s of the VALU-bound kernels; (4) the
same chain with --full-rebuild semantics; (5) the CPU baseline (rank 0, N = 1 only).

A/B switches (none changes what is computed; see include/mpmc_hip.h for the engine options behind them):
  --expanded-matrix / --full-sweep / --full-rebuild, --uvt, --walkers-per-gpu W, and the environment variables
  MPMC_OVERLAP=0|1 (second stream), MPMC_SIDE_AFTER=n (where the side stream is fed), MPMC_STEP_GRAPH=1 (HIP-graph
  replay), MPMC_SYM_MODE, MPMC_GS_DEBUG, MPMC_RESIDENT / MPMC_RESIDENT_FOLD (one-launch dipole solve, with / without
  finisher workgroups), MPMC_SIDE_MOVES / MPMC_SPLIT_RECORD / MPMC_FUSE_FIELD / MPMC_RANK_LATE (=0: fork event, join event,
  coefficient update as its own launch, ranking work enqueued first), MPMC_WALKERS_ONE_THREAD=1, MPMC_HIP_HOST_PROFILE=1
  (host-side timing of energy(): printed to stderr when the contexts are destroyed).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK = 78.6e12   # MI355X fp64 vector peak, FLOP/s (256 CUs x 4 SIMD x 16 lanes x 2 x 2.4 GHz)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="pcn61_4096")
    ap.add_argument("--corrtime", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0,
                    help="seconds of single-core CPU work for the cpu_baseline sample (20 for the headline; the per-size "
                         "lines of profiles/round3_profile.sh use 8)")
    ap.add_argument("--cpu-all-cores", type=int, default=1,
                    help="also time one CPU walker per core on all cores (0 = skip)")
    ap.add_argument("--walkers-per-gpu", type=int, default=1,
                    help="independent walkers driven by each rank on its GPU, interleaved (default 1 = the headline "
                         "definition: one Markov chain per GPU); value counts the steps of all of them")
    ap.add_argument("--uvt", action="store_true",
                    help="grand-canonical chain (insert / remove / displace) with the flags of the reference's "
                         "3_PCN61/iter.inp (insert_probability 0.666, pressure 70 atm as the fugacity) instead of NVT")
    ap.add_argument("--full-sweep", action="store_true", help="A/B: stream the full matrix instead of its upper triangle")
    ap.add_argument("--full-rebuild", action="store_true", help="A/B: rebuild every cached unit from scratch every step")
    ap.add_argument("--expanded-matrix", action="store_true",
                    help="A/B: sweep over the expanded 3N x 3N matrix (72 B per pair) instead of pair coefficients (16 B)")
    ap.add_argument("--deadline", type=float, default=900.0,
                    help="seconds the self-started ranks of --gpus N get, rank 0 included; after that exactly the PIDs "
                         "started here are killed and the exit code is non-zero")
    ap.add_argument("--launch-fault", default="",
                    help="test hook of --launch-check: 'exit:R' makes rank R exit with code 3 before the rendezvous, "
                         "'hang:R' makes it sleep past any deadline")
    ap.add_argument("--launch-check", action="store_true",
                    help="exercise the rank launcher and the seed/pooling bookkeeping WITHOUT a GPU (gloo, no energy is "
                         "evaluated, value is null): what the CPU tests run")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: N ranks as child processes, started before this process touches torch or HIP
# ---------------------------------------------------------------------------------------------------
def visible_gpus():
    """GPUs this process could use, counted from sysfs (no HIP call, no torch import)."""
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0 and int(props.get("gfx_target_version", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args):
    """Start args.gpus ranks of this script and relay rank 0's JSON line.  Returns the exit code."""
    n = args.gpus
    if not args.launch_check:
        have = visible_gpus()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d needs %d GPUs, %d visible on this host (one walker per GPU; "
                             "nothing was run)\n" % (n, n, have))
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile

    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MPMC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
    # ONE deadline for all ranks, rank 0 included: a rank stuck in a collective (ncclCommInitRank on a node where one
    # peer died, a rendezvous nobody joins) must end as a non-zero exit with a message, not as a run the caller's own
    # time limit kills with nothing written.
    deadline = time.time() + args.deadline
    out, timed_out = b"", []
    try:
        out, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        timed_out.append(0)
    rcs = [procs[0].returncode]
    # once rank 0 is done (or out of time) the others get a short grace period: they leave the last barrier together
    grace = time.time() + (20.0 if not timed_out else 0.0)
    for r, p in enumerate(procs[1:], start=1):
        try:
            rcs.append(p.wait(timeout=max(0.5, min(deadline, grace) - time.time())))
        except subprocess.TimeoutExpired:
            timed_out.append(r)
            rcs.append(None)
    for r, p in enumerate(procs):
        if p.poll() is None:
            p.kill()  # exactly the PIDs started above
            p.wait()
            if r == 0:
                try:
                    out = p.stdout.read() or b""
                except Exception:  # noqa: BLE001
                    out = b""
            rcs[r] = -9
    text = out.decode(errors="replace") if out else ""
    sys.stdout.write(text)
    sys.stdout.flush()
    failed = timed_out or any(rc != 0 for rc in rcs)
    for r, f in enumerate(errs):
        f.seek(0)
        tail = f.read().decode(errors="replace").splitlines()[-(12 if failed else 3):]
        f.close()
        if tail and (failed or r == 0):
            sys.stderr.write("".join("[rank %d] %s\n" % (r, ln) for ln in tail))
    if failed:
        sys.stderr.write("bench.py: rank exit codes %s%s\n" % (rcs, ("; ranks %s were still running at the %.0f s deadline "
                         "and were killed" % (timed_out, args.deadline)) if timed_out else ""))
        return 1
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if len(lines) != 1 or json.loads(lines[0]).get("n_gpus") != n:
        sys.stderr.write("bench.py: expected one JSON line with n_gpus = %d from rank 0\n" % n)
        return 1
    return 0


# ---------------------------------------------------------------------------------------------------
# workloads, CPU baseline
# ---------------------------------------------------------------------------------------------------
def load_workload(name):
    import numpy as np
    from mpmc_amd import synth

    if name == "pcn61_4096":
        s = dict(np.load(os.path.join(ROOT, "tests", "golden", "pcn61_bssp_4096.npz")))
        flags = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, pbc_cutoff=8.0,
                     feynman_hibbs=1, feynman_hibbs_order=4)
        label = "PCN-61 cell + 416 BSSP H2, 4096 atoms (2016 frozen), polarizable, Jacobi x4, rc 8 A, FH4"
        return s, flags, label
    if name == "pcn61_21183":
        # the reference's GPU sample in full: sample_configs_gpu/3_PCN61/{iter.inp, input.pdb} (tests/data/pcn61_full)
        from mpmc_amd import pqr

        s = pqr.read_pqr(os.path.join(ROOT, "tests", "data", "pcn61_full", "input.pdb.gz"),
                         np.diag([128.388, 42.796, 42.796]))
        flags = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, pbc_cutoff=8.0,
                     feynman_hibbs=1, feynman_hibbs_order=4)
        return s, flags, "PCN-61 3x1x1 + 3027 BSSP H2, 21183 atoms (6048 frozen), polarizable, Jacobi x4, rc 8 A, FH4"
    kind, n = name.rsplit("_", 1)
    n = int(n)
    if kind == "spol":
        return synth.s_pol(n), dict(synth.FLAGS_POL_JACOBI), "S-POL(%d): BSSP H2 box, Jacobi x10, FH4" % n
    if kind == "spolprod":
        return synth.s_pol(n), dict(synth.FLAGS_POL_PRODUCTION), "S-POL(%d): production flags (wolf, GS-ranked, Palmo)" % n
    if kind == "ses":
        return synth.s_es(n), dict(synth.FLAGS_ES), "S-ES(%d): LJ + Ewald dimers" % n
    if kind == "slj":
        return synth.s_lj(n), dict(synth.FLAGS_LJ), "S-LJ(%d): LJ only" % n
    raise SystemExit("unknown workload " + name)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_walker(workload, budget_s, core, max_steps=200):
    """One CPU walker of the workload on one pinned core: the oracle (C restatement of the reference CPU path
    with its per-pair caching between steps) driven with single-molecule moves.  Returns (steps, seconds)."""
    import numpy as np
    from oracle import oracle

    if core is not None:
        try:
            os.sched_setaffinity(0, {core})
        except OSError:
            pass
    system, flags, _ = load_workload(workload)
    n = len(system["charge"])
    mol = np.asarray(system["molecule"])
    frozen = np.asarray(system["frozen"])
    starts = np.flatnonzero(np.r_[True, mol[1:] != mol[:-1]])
    ends = np.r_[starts[1:], n]
    movable = [(int(a), int(b)) for a, b in zip(starts, ends) if not frozen[a]]
    rng = np.random.default_rng(7 + (core or 0))
    s = dict(system)
    s["pos"] = np.array(system["pos"], dtype=np.float64)
    cache = oracle.Cache(n)
    oracle.energy(s, flags, cache=cache)  # first call computes every pair (untimed, like the GPU warm-up)
    t0 = time.perf_counter()
    nstep = 0
    while True:
        a, b = movable[rng.integers(len(movable))]
        s["pos"][a:b] += 0.05 * (rng.random(3) - 0.5)
        oracle.energy(s, flags, cache=cache)
        nstep += 1
        el = time.perf_counter() - t0
        if el > budget_s or nstep >= max_steps:
            break
    cache.close()
    return nstep, el


def cpu_baseline(workload, budget_s=20.0, all_cores=True):
    """The CPU path timed on this host (reported baseline, not the target): one pinned core, and -- clearly
    labelled -- one walker per core on all cores."""
    cores = sorted(os.sched_getaffinity(0))
    saved = set(cores)
    nstep, el = cpu_walker(workload, budget_s, cores[0])
    os.sched_setaffinity(0, saved)
    out = dict(value=nstep / el, unit="MC steps/s", cores=1, kind="port", cpu_model=cpu_model(),
               pinned_core=cores[0],
               sample="%d single-molecule moves + energy() of the same workload by oracle/ (C restatement of the "
                      "reference CPU path with its per-pair caching, gcc -O3, 1 thread pinned to core %d), %.1f s"
                      % (nstep, cores[0], el))
    if all_cores and len(cores) > 1:
        # throughput-equivalent line: one independent walker per core, all cores of this job's CPU share at once
        # (fresh interpreters: nothing of this process's GPU state is inherited).  The pool is sized to the share
        # (16 cores per GPU on the bench boxes, whatever the affinity mask says) AND to memory: every walker holds
        # its own pair cache and A matrix (~3 GB at 4096 atoms), and a host-memory overrun ends the whole box.
        import resource

        rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0 + 0.5  # this process ran one walker
        avail_gb = 0.0
        try:
            for line in open("/proc/meminfo"):
                if line.startswith("MemAvailable"):
                    avail_gb = float(line.split()[1]) / 1048576.0
        except OSError:
            pass
        nproc = min(len(cores), 16, max(1, int(0.25 * avail_gb / rss_gb)))
        use = cores[:nproc]
        code = ("import sys, json; sys.path.insert(0, %r); import bench; "
                "n, el = bench.cpu_walker(%r, %r, int(sys.argv[1]), 60); print(json.dumps([n, el]))"
                % (ROOT, workload, budget_s * 0.6))
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-c", code, str(c)], stdout=subprocess.PIPE,
                                  env=dict(os.environ, OMP_NUM_THREADS="1")) for c in use]
        rates = []
        for p in procs:
            o, _ = p.communicate()
            if p.returncode == 0:
                n, e = json.loads(o.decode().strip().splitlines()[-1])
                rates.append(n / e)
        out["all_cores"] = dict(value=sum(rates), unit="MC steps/s", cores=len(rates),
                                cores_in_affinity_mask=len(cores),
                                note="one independent CPU walker per core x %d cores at once (sum of their rates; "
                                     "not a parallel energy(); pool = min(cores in the mask, 16 = the CPU share of one "
                                     "GPU, memory / 4)); wall %.1f s" % (len(rates), time.perf_counter() - t0))
    return out


# ---------------------------------------------------------------------------------------------------
# the launch check: launcher + seeds + pooling, no GPU
# ---------------------------------------------------------------------------------------------------
def launch_check(args, rank, world):
    import numpy as np
    import torch.distributed as dist
    from mpmc_amd import host, synth
    from mpmc_amd.walkers import TorchReducer, WalkerAverages, walker_seed

    if args.launch_fault:
        kind, _, who = args.launch_fault.partition(":")
        if int(who or -1) == rank:
            if kind == "exit":
                sys.stderr.write("launch-check: rank %d exits on request\n" % rank)
                sys.exit(3)
            if kind == "hang":
                time.sleep(3600.0)
    if world > 1:
        import datetime

        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=min(60.0, args.deadline)))
    seed = walker_seed(args.seed, rank)
    h = host.HostSystem(synth.s_pol(10), synth.FLAGS_POL_JACOBI, seed=seed)
    avg = WalkerAverages(reducer=TorchReducer(dist) if world > 1 else None)
    draws = []
    for k in range(args.steps):
        u = h.lib.host_get_rand(h.ptr)  # the walker's own stream (std::mt19937 seeded with seed + rank)
        draws.append(u)
        avg.add(u, 0.0, 0.0, 0.0, 0.0, 1.0)
        if (k + 1) % args.corrtime == 0:
            avg.reduce()
    avg.reduce()
    summ = avg.summary()
    h.close()
    first = [draws[0]]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, draws[0])
        first = gathered
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "MC steps/sec (polarizable, 4096 atoms)", "value": None, "unit": "MC steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "launch_check": True,
                          "seeds": [walker_seed(args.seed, r) for r in range(world)], "first_draws": first,
                          "pooled_samples": summ["samples"], "pooled_mean": summ["energy"]}))
    if world > 1:
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))  # nothing GPU-related has been imported or initialised yet

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE = %d ranks (one walker per GPU: "
                         "they must agree)" % (args.gpus, world))
    if args.launch_check:
        sys.exit(launch_check(args, rank, world))

    import numpy as np
    import torch

    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d needs GPU %d; %d visible (this engine has no CPU fallback)"
                         % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # under torchrun, also with a single rank
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from mpmc_amd import host
    from mpmc_amd.walkers import AbiReducer, WalkerAverages, walker_seed

    system, flags, label = load_workload(args.workload)
    n = len(system["charge"])
    # host control stays in C: system_t + energy() + the NVT chain of mpmc_amd/host/ drive the engine through the C ABI
    extra = {"ensemble": "uvt", "insert_probability": 0.666, "pressure": 70.0} if args.uvt else None
    W = max(1, args.walkers_per_gpu)
    chains = [host.HostSystem(system, flags, device=local_rank, seed=walker_seed(args.seed, rank, W, w), extra=extra)
              for w in range(W)]
    chain = chains[0]
    for ch in chains:
        ch.energy()  # creates the device context, uploads the configuration
    # ---- walker pooling through the C ABI: rank 0 makes the RCCL id, the launcher's process group hands it round
    rccl_ranks = 1
    pooled = dist is not None  # under a launcher the collective runs also with one rank (same code path as N ranks)
    collective = "none (single walker, no launcher)"
    reducer = None
    if pooled:
        # Every rank tries the C ABI's communicator; they then agree (one all-reduce of the launcher's group) on whether
        # ALL of them have it.  If any could not (librccl missing, an init error), all fall back together to the
        # launcher's own all-reduce and the line says so -- a labelled number instead of a crashed scaling run.
        # ncclCommInitRank is itself a collective: a rank that fails BEFORE it (librccl not loadable, no context)
        # would leave the others blocked inside it.  So first every rank proves locally that it can enter -- making an
        # id of its own loads librccl and runs it once -- and the ranks agree (MIN over the launcher's group); only then
        # do all of them call walkers_init with rank 0's id.
        err = ""
        my_id = None
        try:
            my_id = host.walkers_unique_id()
        except Exception as e:  # noqa: BLE001 -- reported in the line
            err = repr(e)
        pre = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(pre, op=dist.ReduceOp.MIN)
        if float(pre.item()) > 0.5:
            ids = [my_id if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            try:
                chain.walkers_init(world, rank, ids[0])
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        else:
            err = err or "another rank cannot load / run librccl"
        okt = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if float(okt.item()) > 0.5:
            reducer = AbiReducer(chain)
            rccl_ranks = world
            collective = "mpmc_hip_allreduce_observables_begin/_end (C ABI, RCCL) every corrtime"
        else:
            from mpmc_amd.walkers import TorchReducer

            reducer = TorchReducer(dist, device=dev)
            rccl_ranks = 0
            collective = ("torch.distributed all_reduce (nccl = RCCL) every corrtime -- FALLBACK: the C ABI's communicator "
                          "could not be made on every rank (%s)" % (err or "another rank failed"))
    avg = WalkerAverages(reducer=reducer)

    for var, opt in (("MPMC_OVERLAP", "overlap_streams"), ("MPMC_SIDE_AFTER", "side_after"),
                     ("MPMC_STEP_GRAPH", "step_graph"), ("MPMC_SYM_MODE", "sym_mode"), ("MPMC_GS_DEBUG", "persistent_gs"),
                     ("MPMC_RESIDENT", "resident_jacobi"), ("MPMC_RESIDENT_FOLD", "resident_fold"), ("MPMC_SIDE_MOVES", "side_moves"), ("MPMC_SPLIT_RECORD", "split_record"), ("MPMC_FUSE_FIELD", "fuse_field"), ("MPMC_RANK_LATE", "rank_late"), ("MPMC_FUSE_RECIP", "fuse_recip"), ("MPMC_FUSE_TENSOR", "fuse_tensor"), ("MPMC_GS_FOLD_UPPER", "gs_fold_upper"), ("MPMC_RANK_VIEW_SIDE", "rank_view_side"), ("MPMC_SWEEP_ALTERNATE", "sweep_alternate"),
                     ("MPMC_SWEEP_NT", "sweep_nt"), ("MPMC_SWEEP_SPLIT", "sweep_split"), ("MPMC_GS_FUSE_MOVES", "gs_fuse_moves"), ("MPMC_GS_LAGS", "gs_lags"), ("MPMC_GS_BUILD_FORK", "gs_build_fork"), ("MPMC_GS_SIDE_WAVES", "gs_side_waves"), ("MPMC_FUSE_MOVES", "fuse_moves"), ("MPMC_GS_FOLD_FINISH", "gs_fold_finish")):
        if os.environ.get(var):
            chain.set_option(opt, int(os.environ[var]))
    if args.full_sweep or args.full_rebuild or args.expanded_matrix:
        for ch in chains:
            ch.set_option("symmetric_sweep", 0 if args.full_sweep else 1)
            ch.set_option("incremental_amatrix", 0 if args.full_rebuild else 1)
            ch.set_option("incremental_pairs", 0 if args.full_rebuild else 1)
            ch.set_option("pair_coefficients", 0 if (args.expanded_matrix or args.full_sweep) else 1)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    pool = None
    if W > 1 and not os.environ.get("MPMC_WALKERS_ONE_THREAD"):
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(W)

    def run(nsteps):
        done = 0
        while done < nsteps:
            k = min(args.corrtime, nsteps - done)
            if W == 1:
                acc = chain.mc_steps(k)  # raises on a device / ABI failure: never counted as rejected moves
            elif pool is None:
                acc = host.mc_steps_multi(chains, k)  # one host thread feeds all walkers
            else:
                acc = sum(pool.map(lambda ch: ch.mc_steps(k), chains))  # one host thread per walker (ctypes drops the GIL)
            done += k
            for ch in chains:
                o = ch.observables()
                avg.add(o["energy"], o["rd_energy"], o["coulombic_energy"], o["polarization_energy"],
                        o["polar_iterations"], acc / float(k * W))
            # walker averaging every corrtime (reference mc.c:417-432: MPI_Gather of observables)
            avg.reduce()

    for ch in chains:
        ch.set_option("timing", 0)  # no events anywhere near the timed region
    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    avg.summary()  # completes the last (asynchronous) walker all-reduce inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_own = elapsed
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # ---- untimed pass 1: the dominant kernel under HIP events on the engine's stream, every launch, raw mean
    nprobe = 96
    chain.set_option("timing", 1)
    chain.set_option("timing_interval", 1)
    chain.enable_timing(True)
    chain.mc_steps(nprobe)
    acc = chain.timings()
    # ---- untimed pass 2: every kernel class
    nb = 32
    chain.set_option("timing", 2)
    chain.enable_timing(True)
    chain.mc_steps(nb)
    brk = chain.timings()
    # ---- untimed pass 3: one full (non-incremental) evaluation: the VALU-bound pair / field kernels over all tiles
    chain.set_option("incremental_pairs", 0)
    chain.enable_timing(True)
    chain.mc_steps(4)
    full = chain.timings()
    chain.set_option("incremental_pairs", 1)
    # ---- untimed pass 4: the same chain with every cached unit rebuilt every step
    chain.set_option("timing", 0)
    chain.set_option("incremental_amatrix", 0)
    chain.set_option("incremental_pairs", 0)
    nfr = max(10, min(200, args.steps // 4))
    chain.mc_steps(5)
    torch.cuda.synchronize(dev)
    tf0 = time.perf_counter()
    chain.mc_steps(nfr)
    torch.cuda.synchronize(dev)
    full_rebuild_rate = nfr / (time.perf_counter() - tf0)

    # engine counters that would silently change what the headline measures (cumulative over this rank's whole run)
    ctr = chain.timings()
    rank_rec = {"rank": rank, "device": local_rank, "steps_per_s": W * args.steps / elapsed_own, "rccl_ok": rccl_ranks == world,
                "own_mean_energy": avg.own_mean_energy(), "resident_fallbacks": int(ctr["resident_fallbacks"]),
                "spec_rank_redos": int(ctr["spec_rank_redos"]), "resident_calls": int(ctr["resident_calls"])}
    ranks = [rank_rec]
    if dist is not None and world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_rec)
    # the preset_seeds trap (mersenne.cpp:13-14: every rank the same seed): walkers that are copies of each other would
    # scale "perfectly" and average nothing.  Two ranks with the same mean energy over the run = the run is void.
    means = [r["own_mean_energy"] for r in ranks]
    identical = world > 1 and len(set(means)) < len(means)

    if rank == 0:
        value = world * W * args.steps / elapsed
        sweep_raw_ms = acc["sweep_ms"] / max(1, acc["sweep_count"])
        event_pair_ms = acc["event_pair_ms"] / max(1, acc["event_pair_count"])
        # algorithmic bytes of one sweep launch (pair_sweep_kernel): one {c3, c5} coefficient pair (16 B, fp64)
        # per unordered pair of polarizable sites, read once, + coordinates and dipoles in, field out.
        # (sites with alpha = 0 carry no dipole, so neither their rows nor their columns exist)
        n_pol_initial = int(np.count_nonzero(np.asarray(system["alpha"]) != 0.0))
        n_pol, n_final = n_pol_initial, n
        if args.uvt:
            # a grand-canonical chain changes its own size: the sweep that was timed ran over the configuration the chain
            # had reached by then (at 70 atm the PCN-61 cell takes up ~15 % more H2 than it starts with), so its bytes are
            # counted on THAT many polarizable sites -- pricing it at the initial count made the same kernel look 18 %
            # slower under uvt than under nvt (round 2: 19.4 vs 16.4 us "on the same atoms", which they were not)
            fin = chain.system(system["basis"])
            n_final = len(fin["charge"])
            n_pol = int(np.count_nonzero(fin["alpha"] != 0.0))
        m3 = 3.0 * n_pol
        gs = bool(flags.get("polar_gs") or flags.get("polar_gs_ranked"))
        expanded = (args.expanded_matrix or args.full_sweep) and not gs
        sym_off = ((n_pol + 127) // 128 * 128) < 2048  # size threshold of the symmetric expanded-matrix kernel
        nblk = (n_pol + 63) // 64
        if gs:
            # gs_chain_kernel (the persistent lower-triangle launch of one exact Gauss-Seidel sweep): the {c3, c5}
            # coefficients of every tile (t, s <= t-2) (16 B per pair), the sub-diagonal tiles (t, t-1) as expanded
            # tensors (48 B per pair), the cached inverse of every diagonal block (32 x 9 x 64 doubles) and the vectors.
            # (Round 2's accounting, kept so that the fraction stays comparable: round 3's kernel reads MORE than this --
            # the cached matrices P_t, Q_t at 72 B per pair of the tiles (t, t-1), (t, t-2) -- by design, to shorten the
            # chain; the PMC traffic in profiles/ shows what it really moves.)
            sweep_bytes = (max(nblk - 1, 0) * max(nblk - 2, 0) / 2) * 4096 * 16 + max(nblk - 1, 0) * 4096 * 48 \
                + nblk * 18432 * 8 + 6 * m3 * 8
            kernel_name = "gs_chain_kernel"
        elif not expanded and acc.get("resident_calls", 0) > 0 and acc["sweep_count"] <= nprobe + 1:
            # small views: the whole solve is ONE launch (jacobi_folded_kernel up to 16 blocks, jacobi_resident_kernel up
            # to 21) whose tiles stay in registers; priced
            # at the bytes the sweeps of the solve need algorithmically (n_iter passes over the coefficients), so this
            # is an EFFECTIVE bandwidth -- the launch itself reads the coefficients from memory once
            n_iter = int(flags.get("polar_max_iter", 10))
            sweep_bytes = n_iter * (n_pol * (n_pol - 1) / 2 * 16 + 3 * m3 * 8)
            kernel_name = ("%s (%d sweeps in one launch, tiles held in registers: effective bandwidth)"
                           % ("jacobi_folded_kernel" if nblk <= 16 and not os.environ.get("MPMC_RESIDENT_FOLD")
                              else "jacobi_resident_kernel / jacobi_folded_kernel", n_iter))
        elif not expanded:
            sweep_bytes = n_pol * (n_pol - 1) / 2 * 16 + 3 * m3 * 8
            kernel_name = "pair_sweep_kernel"
        elif args.full_sweep or sym_off:
            sweep_bytes = m3 * m3 * 8 + 5 * m3 * 8
            kernel_name = "sweep_kernel<Jacobi>"
        else:  # upper triangle of the expanded matrix
            sweep_bytes = m3 * (m3 + 1) / 2 * 8 + 3 * m3 * 8
            kernel_name = "symv_kernel"
        achieved = sweep_bytes / (sweep_raw_ms * 1e-3) / 1e9 if sweep_raw_ms > 0 else 0.0
        # what the committed rocprofv3 profile of this command says (profiles/run_profile.sh writes it)
        traffic = None
        rocprof_ms = None
        pmc = os.path.join(ROOT, "profiles", "sweep_pmc_latest.json")
        if os.path.exists(pmc) and args.workload == "pcn61_4096":
            try:
                rec = json.load(open(pmc))
                if kernel_name.split("<")[0].split(" ")[0] in rec.get("kernel", ""):
                    traffic = rec.get("hbm_bytes_per_launch")
                    rocprof_ms = rec.get("rocprof_avg_launch_ms")
            except Exception:
                traffic = None
        # VALU-bound kernels over ALL tiles (pass 3): pairs/s and an estimate of the fp64 vector peak they use.
        # Per pair the screen costs ~31 fp32 operations (counted as 15.5 fp64-equivalents at the 2:1 rate); the ~3 %
        # of pairs inside the cutoff + margin pay the exact path: minimum image 46 + LJ/FH4 ~70 + erfc/exp ~60 fp64
        # operations for the pair kernel, 46 + ~25 for the bare-field kernel.  Estimates, stated as such.
        npair = n * (n - 1) / 2.0
        pair_s = full["pair_ms"] / 4 * 1e-3
        field_s = full["field_ms"] / 4 * 1e-3
        valu = {}
        if pair_s > 0:
            valu["pair_rd_es_kernel"] = dict(pairs_per_s=npair / pair_s, ms_full_pass=pair_s * 1e3,
                                             est_flop_per_pair=15.5 + 0.03 * 176,
                                             est_frac_fp64_valu_peak=npair * (15.5 + 0.03 * 176) / pair_s / FP64_VALU_PEAK)
        if field_s > 0:
            valu["static_field_kernel"] = dict(pairs_per_s=2 * npair / field_s, ms_full_pass=field_s * 1e3,
                                               est_flop_per_pair=15.5 + 0.03 * 71,
                                               est_frac_fp64_valu_peak=2 * npair * (15.5 + 0.03 * 71) / field_s / FP64_VALU_PEAK)
        # what binds the dominant kernel: its coefficient set is re-read by every sweep; below the 256 MB Infinity Cache
        # those re-reads can be served on-die, so the 8 TB/s HBM figure is the price list, not a proven roof, at that size
        # (the same kernel at 16 384 atoms, 773 MB of coefficients, is the beyond-cache figure: DESIGN.md section 5)
        bound_label = "hbm" if sweep_bytes > 256e6 else "hbm (set fits Infinity Cache)"
        if gs:
            bound_label = "hbm (latency chain: see DESIGN.md section 3)"
        out = {
            "metric": "MC steps/sec (polarizable, 4096 atoms)",
            "value": value,
            "unit": "MC steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if args.workload != "pcn61_4096" else
                    "reference sample geometry (PCN-61 cell carved from sample_configs_gpu/3_PCN61/input.pdb), "
                    "random MC moves",
            "config": {"workload": label + (" [UVT: insert/remove/displace]" if args.uvt else "") +
                                   ("; energy() rebuilt from scratch every step" if args.full_rebuild else
                                    "; energy() incremental (bit-identical to full)"),
                       "n_atoms": n, "n_polarizable": n_pol_initial, "n_atoms_final": n_final,
                       "n_polarizable_final": n_pol, "walkers": world * W, "corrtime": args.corrtime,
                       "parallelism": "%d independent walkers, %d per GPU" % (world * W, W),
                       "collective": collective,
                       "rccl_ranks": rccl_ranks},
            "full_rebuild_steps_per_s": full_rebuild_rate,
            # a dedicated box shows 0 / 0: a resident (one-launch) solve that lost a hand-off is repeated launch by launch
            # and the context stays on that path; a mis-speculated ranked walk is repeated with the host sorting
            "resident_fallbacks": sum(r["resident_fallbacks"] for r in ranks),
            "spec_rank_redos": sum(r["spec_rank_redos"] for r in ranks),
            "resident_calls": sum(r["resident_calls"] for r in ranks),
            "ranks": ranks,
            "roofline": {"kernel": kernel_name + " (Thole field / dipole sweep)", "bound": bound_label,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": sweep_raw_ms, "launches": acc["sweep_count"],
                         "event_pair_ms": event_pair_ms,
                         "rocprof_avg_launch_ms": rocprof_ms,
                         "algorithmic_bytes_per_launch": sweep_bytes,
                         # SURVEY 8(d) asks for these two beside a sweep that does not stream a stored matrix:
                         "pairs_per_s": (n_pol * (n_pol - 1) / 2) / (sweep_raw_ms * 1e-3) if sweep_raw_ms > 0 else 0.0,
                         "effective_GBps_reference_layout": ((m3 * m3 * 8) / (sweep_raw_ms * 1e-3) / 1e9
                                                             if sweep_raw_ms > 0 else 0.0),
                         "note": "avg_launch_ms = raw mean of HIP-event pairs on the engine's stream around every launch "
                                 "of the kernel in a separate untimed pass of %d steps (nothing subtracted: an event pair "
                                 "itself adds event_pair_ms, so this is an upper bound of the kernel time and frac a lower "
                                 "bound; rocprof_avg_launch_ms is the committed rocprofv3 mean of the same command); "
                                 "achieved = bytes this design moves (16 B per unordered pair); "
                                 "effective_GBps_reference_layout prices the same time at the reference's (3N)^2 x 8 B "
                                 "matrix and is not a bandwidth claim; the 87 MB coefficient set is re-read every sweep and "
                                 "fits the 256 MB Infinity Cache, so part of the traffic may be served on-die (FETCH_SIZE "
                                 "counts memory-side requests including such hits)" % nprobe},
            "valu_kernels": valu,
            "device_ms_per_step": dict({k: brk[k] / nb for k in
                                        ("pair_ms", "recip_ms", "field_ms", "amatrix_ms", "sweep_ms", "palmo_ms",
                                         "other_ms", "total_ms")},
                                       note="separate untimed pass of %d steps with every kernel class timed" % nb),
            "walker_averages": avg.summary(),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, budget_s=args.cpu_budget, all_cores=bool(args.cpu_all_cores))
        print(json.dumps(out))
    if identical:
        sys.stderr.write("bench.py: walkers with IDENTICAL averages %s -- the ranks ran the same chain (seed + rank not "
                         "applied?): the run is void\n" % means)
    for ch in chains:
        ch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if identical:
        sys.exit(4)


if __name__ == "__main__":
    main()
