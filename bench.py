#!/usr/bin/env python3
"""bench.py -- MC steps/s of the polarizable 4096-atom box on N x MI355X (one walker per GPU).

One "step" = one single-molecule displacement (translate + rotate, reference
src/mc/mc_moves.c:378-488) + one energy() on the device through the C ABI + Metropolis
(reference src/mc/mc.c:294-353), exactly what the reference does per step; the loop itself is the
C host layer of mpmc_amd/host/ (system_t, energy(), checkpoint/make_move/restore), called via ctypes.  Workload: the
4096-atom PCN-61 cell + 416 BSSP H2 of tests/golden/pcn61_bssp_4096.npz with the flags of the
reference's sample_configs_gpu/3_PCN61/iter.inp run as NVT (Jacobi x4, cutoff 8 A, FH 4th order)
-- BASELINE.json configs[3]; `--workload` selects the synthetic boxes instead.

Walkers are independent: rank r runs its own chain with seed+r on GPU r; the only
collective is the sum of a small observable vector every `corrtime` steps (RCCL over xGMI via
torch.distributed, backend "nccl").  Prints ONE JSON line on rank 0.

A/B switches (none changes what is computed; see include/mpmc_hip.h for the engine options behind them):
  --expanded-matrix / --full-sweep / --full-rebuild, --uvt, --walkers-per-gpu W, and the environment variables
  MPMC_OVERLAP=0|1 (second stream), MPMC_SIDE_AFTER=n (where the side stream is fed), MPMC_STEP_GRAPH=1 (HIP-graph
  replay), MPMC_SYM_MODE, MPMC_GS_DEBUG, MPMC_WALKERS_ONE_THREAD=1, MPMC_HIP_HOST_PROFILE=1 (host-side timing of
  energy(): printed to stderr when the contexts are destroyed).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def load_workload(name):
    from mpmc_amd import synth

    if name == "pcn61_4096":
        s = dict(np.load(os.path.join(ROOT, "tests", "golden", "pcn61_bssp_4096.npz")))
        flags = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, pbc_cutoff=8.0,
                     feynman_hibbs=1, feynman_hibbs_order=4)
        label = "PCN-61 cell + 416 BSSP H2, 4096 atoms (2016 frozen), polarizable, Jacobi x4, rc 8 A, FH4"
        return s, flags, label
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


def cpu_baseline(system, flags, budget_s=20.0):
    """The CPU oracle (a port of the reference path, including its per-pair caching between steps)
    timed on this host, 1 core, bounded sample of the same workload: single-molecule moves + energy()."""
    from oracle import oracle

    n = len(system["charge"])
    mol = np.asarray(system["molecule"])
    frozen = np.asarray(system["frozen"])
    starts = np.flatnonzero(np.r_[True, mol[1:] != mol[:-1]])
    ends = np.r_[starts[1:], n]
    movable = [(int(a), int(b)) for a, b in zip(starts, ends) if not frozen[a]]
    rng = np.random.default_rng(7)
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
        if el > budget_s or nstep >= 200:
            break
    cache.close()
    return dict(value=nstep / el, unit="MC steps/s", cores=1, kind="port",
                sample="%d single-molecule moves + energy() of the same workload by oracle/ (C restatement of the "
                       "reference CPU path with its per-pair caching, gcc -O3, 1 thread), %.1f s" % (nstep, el))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="pcn61_4096")
    ap.add_argument("--corrtime", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--walkers-per-gpu", type=int, default=1,
                    help="independent walkers driven by each rank on its GPU, interleaved (default 1 = the headline "
                         "definition: one Markov chain per GPU); value counts the steps of all of them")
    ap.add_argument("--uvt", action="store_true",
                    help="grand-canonical chain (insert / remove / displace) with the flags of the reference's "
                         "3_PCN61/iter.inp (insert_probability 0.666, pressure 70 atm as the fugacity) instead of NVT")
    ap.add_argument("--full-sweep", action="store_true", help="A/B: stream the full matrix instead of its upper triangle")
    ap.add_argument("--full-rebuild", action="store_true", help="A/B: rebuild A from scratch every step")
    ap.add_argument("--expanded-matrix", action="store_true",
                    help="A/B: sweep over the expanded 3N x 3N matrix (72 B per pair) instead of pair coefficients (16 B)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # under torchrun, also with a single rank
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from mpmc_amd import host
    from mpmc_amd.walkers import WalkerAverages

    system, flags, label = load_workload(args.workload)
    n = len(system["charge"])
    # host control stays in C: system_t + energy() + the NVT chain of mpmc_amd/host/ drive the engine through the C ABI
    extra = {"ensemble": "uvt", "insert_probability": 0.666, "pressure": 70.0} if args.uvt else None
    W = max(1, args.walkers_per_gpu)
    chains = [host.HostSystem(system, flags, device=local_rank, seed=args.seed + rank * W + w, extra=extra)
              for w in range(W)]
    chain = chains[0]
    avg = WalkerAverages(dist=dist, device=dev)
    for ch in chains:
        ch.energy()  # creates the device context, uploads the configuration
    if os.environ.get("MPMC_OVERLAP"):
        chain.energy()
        chain.set_option("overlap_streams", int(os.environ["MPMC_OVERLAP"]))
    if os.environ.get("MPMC_SIDE_AFTER"):
        chain.energy()
        chain.set_option("side_after", int(os.environ["MPMC_SIDE_AFTER"]))
    if os.environ.get("MPMC_STEP_GRAPH"):
        chain.energy()
        chain.set_option("step_graph", int(os.environ["MPMC_STEP_GRAPH"]))
    if os.environ.get("MPMC_SYM_MODE"):
        chain.energy()
        chain.set_option("sym_mode", int(os.environ["MPMC_SYM_MODE"]))
    if os.environ.get("MPMC_GS_DEBUG"):
        chain.energy()
        chain.set_option("persistent_gs", int(os.environ["MPMC_GS_DEBUG"]))
    if args.full_sweep or args.full_rebuild or args.expanded_matrix:
        chain.energy()  # creates the device context
        chain.set_option("symmetric_sweep", 0 if args.full_sweep else 1)
        chain.set_option("incremental_amatrix", 0 if args.full_rebuild else 1)
        chain.set_option("pair_coefficients", 0 if (args.expanded_matrix or args.full_sweep) else 1)

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
                acc = chain.mc_steps(k)
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

    run(args.warmup)
    # timed region: HIP events around the dominant (sweep) kernel only -- every event pair costs a few
    # microseconds of stream time, so the per-class breakdown is taken in a separate, untimed pass below
    chain.set_option("timing", 1)
    # sampled calls cost ~80 us extra (event pairs on the stream + reading them back): every 32nd call, or every
    # 4th in short runs so that a handful of sweeps is still sampled
    chain.set_option("timing_interval", 32 if args.steps >= 128 else 4)
    chain.enable_timing(True)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    avg.summary()  # completes the last (asynchronous) walker all-reduce inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    acc = chain.timings()
    nb = max(10, min(50, args.steps))
    chain.set_option("timing", 2)
    chain.enable_timing(True)
    chain.mc_steps(nb)
    brk = chain.timings()
    chain.set_option("timing", 1)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        value = world * W * args.steps / elapsed
        # event pairs bracket the sweep kernel on every 32nd call of the timed region (every 4th in runs < 128 steps); an EMPTY pair recorded on
        # the same calls measures what the two event records themselves add, and is subtracted
        sweep_raw_ms = acc["sweep_ms"] / max(1, acc["sweep_count"])
        event_pair_ms = acc["event_pair_ms"] / max(1, acc["event_pair_count"])
        sweep_avg_ms = max(sweep_raw_ms - event_pair_ms, 0.0)
        # algorithmic bytes of one sweep launch (pair_sweep_kernel): one {c3, c5} coefficient pair (16 B, fp64)
        # per unordered pair of polarizable sites, read once, + coordinates and dipoles in, field out.
        # (sites with alpha = 0 carry no dipole, so neither their rows nor their columns exist)
        n_pol = int(np.count_nonzero(np.asarray(system["alpha"]) != 0.0))
        m3 = 3.0 * n_pol
        gs = bool(flags.get("polar_gs") or flags.get("polar_gs_ranked"))
        expanded = args.expanded_matrix or args.full_sweep or gs
        sym_off = ((n_pol + 127) // 128 * 128) < 2048  # size threshold of the symmetric expanded-matrix kernel
        if not expanded:
            sweep_bytes = n_pol * (n_pol - 1) / 2 * 16 + 3 * m3 * 8
            kernel_name = "pair_sweep_kernel"
        elif gs:  # exact Gauss-Seidel walks the expanded matrix: upper GEMV + persistent lower-triangle kernel
            sweep_bytes = m3 * m3 * 8 + 5 * m3 * 8
            kernel_name = "gs_upper_kernel + gs_persistent_kernel"
        elif args.full_sweep or sym_off:
            sweep_bytes = m3 * m3 * 8 + 5 * m3 * 8
            kernel_name = "sweep_kernel<Jacobi>"
        else:  # upper triangle of the expanded matrix
            sweep_bytes = m3 * (m3 + 1) / 2 * 8 + 3 * m3 * 8
            kernel_name = "symv_kernel"
        achieved = sweep_bytes / (sweep_avg_ms * 1e-3) / 1e9 if sweep_avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "sweep_pmc_latest.json")
        if os.path.exists(pmc) and args.workload == "pcn61_4096":
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_launch") if kernel_name.split("<")[0] in rec.get("kernel", "") else None
            except Exception:
                traffic = None
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
            "config": {"workload": label + (" [UVT: insert/remove/displace]" if args.uvt else ""), "n_atoms": n, "n_polarizable": n_pol, "walkers": world * W, "corrtime": args.corrtime,
                       "parallelism": "%d independent walkers, %d per GPU" % (world * W, W)},
            "roofline": {"kernel": kernel_name + " (Thole field / dipole sweep)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": sweep_avg_ms, "avg_launch_ms_with_event_pair": sweep_raw_ms,
                         "event_pair_ms": event_pair_ms, "launches": acc["sweep_count"],
                         "algorithmic_bytes_per_launch": sweep_bytes,
                         # SURVEY 8(d) asks for these two beside a sweep that does not stream a stored matrix:
                         "pairs_per_s": (n_pol * (n_pol - 1) / 2) / (sweep_avg_ms * 1e-3) if sweep_avg_ms > 0 else 0.0,
                         "effective_GBps_reference_layout": ((m3 * m3 * 8) / (sweep_avg_ms * 1e-3) / 1e9
                                                             if sweep_avg_ms > 0 else 0.0),
                         "note": "achieved = bytes this design moves (16 B per unordered pair); "
                                 "effective_GBps_reference_layout prices the same time at the reference's (3N)^2 x 8 B "
                                 "matrix and is not a bandwidth claim"},
            "device_ms_per_step": dict({k: brk[k] / nb for k in
                                        ("pair_ms", "recip_ms", "field_ms", "amatrix_ms", "sweep_ms", "palmo_ms",
                                         "other_ms", "total_ms")},
                                       note="separate untimed pass of %d steps with every kernel class timed" % nb),
            "walker_averages": avg.summary(),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(system, flags)
        print(json.dumps(out))
    for ch in chains:
        ch.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
