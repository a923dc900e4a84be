"""GPU tests of the C host layer: energy(system_t*) and the NVT chain driven from C, against the
oracle; the stand-alone driver on a reference-style input directory."""
import os
import subprocess

import numpy as np
import pytest

from mpmc_amd import engine, host, synth
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_energy_matches_oracle_and_engine():
    s = synth.s_pol(640)
    p = dict(synth.FLAGS_POL_JACOBI)
    h = host.HostSystem(s, p, seed=3)
    e = h.energy()
    want = oracle.energy(s, p)
    assert abs(e - want["energy"]) < 1e-9 * abs(want["energy"])
    o = h.observables()
    assert abs(o["polarization_energy"] - want["polarization_energy"]) < 1e-9 * abs(want["polarization_energy"])
    assert o["N"] == 128 and o["polar_iterations"] == 10
    d = h.dipoles()
    wv = oracle.energy(s, p, want_vectors=True)
    assert np.abs(d["mu"] - wv["mu"]).max() < 1e-10 * np.abs(wv["mu"]).max()
    h.close()


def test_host_chain_tracks_the_oracle():
    """Run the C Markov chain; after a run of accepted and rejected moves the energy the chain
    carries must equal the oracle's energy of the chain's current configuration."""
    s = synth.s_pol(320)
    p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, polar_gs=1, polar_palmo=1)
    h = host.HostSystem(s, p, seed=11, move_factor=0.05, rot_factor=0.05)
    acc = h.mc_steps(60)
    o = h.observables()
    assert o["accept"] == acc and o["accept"] + o["reject"] == 60 and 0 < acc
    s2 = dict(s)
    s2["pos"] = h.positions()
    want = oracle.energy(s2, p)
    assert abs(o["energy"] - want["energy"]) < 1e-9 * abs(want["energy"])
    h.close()


@pytest.mark.parametrize("production", [False, True])
def test_interleaved_walkers_take_the_steps_they_would_take_alone(production):
    """host_mc_steps_multi drives several walkers from one process (their kernels interleave on the device).
    Each walker has its own random-number stream and engine context: its trajectory must be, bit for bit, the
    one it produces when it runs alone.  With the production flags three persistent Gauss-Seidel chain kernels share
    the device at once: nothing in them assumes co-residency (a workgroup takes its block from a ticket counter when
    it starts), so they make progress however the workgroups of the three launches are interleaved."""
    s = synth.s_pol(640 if production else 320)
    p = dict(synth.FLAGS_POL_PRODUCTION) if production else dict(synth.FLAGS_POL_JACOBI, polar_max_iter=4)
    seeds = (7, 8, 9)
    alone = []
    for sd in seeds:
        h = host.HostSystem(s, p, seed=sd, move_factor=0.05, rot_factor=0.05)
        h.mc_steps(40)
        alone.append((h.observables()["energy"], h.observables()["accept"], h.positions()))
        h.close()
    walkers = [host.HostSystem(s, p, seed=sd, move_factor=0.05, rot_factor=0.05) for sd in seeds]
    total = 0
    for _ in range(4):
        total += host.mc_steps_multi(walkers, 10)
    assert total == sum(a[1] for a in alone)
    for w, a in zip(walkers, alone):
        o = w.observables()
        assert o["energy"] == a[0] and o["accept"] == a[1] and np.array_equal(w.positions(), a[2])
        w.close()


def test_long_chain_on_the_headline_box_has_no_drift():
    """600 MC steps on the 4096-atom PCN-61 box, everything maintained incrementally (pair coefficients,
    pair / field tile partials, view coordinates): the energy the chain carries at the end must be, bit for
    bit, what a fresh context computes from scratch for the chain's final configuration."""
    s = dict(np.load(os.path.join(ROOT, "tests", "golden", "pcn61_bssp_4096.npz")))
    p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, pbc_cutoff=8.0,
             feynman_hibbs=1, feynman_hibbs_order=4)
    h = host.HostSystem(s, p, seed=5)
    acc = h.mc_steps(600)
    assert 0 < acc < 600
    o = h.observables()
    s2 = dict(s)
    s2["pos"] = h.positions()
    eng = engine.Engine(len(s["charge"]))
    eng.load_system(s2, p)
    fresh = eng.energy()
    for key in ("energy", "rd_energy", "coulombic_energy", "polarization_energy"):
        assert o[key] == fresh[key], key
    eng.close()
    h.close()


def test_same_seed_same_chain():
    s = synth.s_pol(160)
    p = dict(synth.FLAGS_POL_JACOBI)
    out = []
    for _ in range(2):
        h = host.HostSystem(s, p, seed=99, move_factor=0.05, rot_factor=0.05)
        h.mc_steps(30)
        out.append((h.observables()["energy"], h.positions()))
        h.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


def test_uvt_chain_inserts_removes_and_tracks_the_oracle():
    """Grand-canonical chain (insert / remove / displace, reference mc.c:44-104, mc_moves.c:583-697):
    N must fluctuate, and the energy the chain carries must equal the oracle's energy of whatever
    configuration (and atom count) it ended in."""
    s = synth.s_pol(160, spacing=4.5)
    p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4)
    h = host.HostSystem(s, p, seed=21, move_factor=0.05, rot_factor=0.05,
                        extra={"ensemble": "uvt", "insert_probability": 0.6, "pressure": 300.0})
    n0 = h.natoms()
    seen = set()
    for _ in range(12):
        h.mc_steps(10)
        seen.add(h.natoms())
    o = h.observables()
    assert len(seen) > 1, "N never changed"
    assert all((n - n0) % 5 == 0 for n in seen)
    final = h.system(s["basis"])
    assert len(final["charge"]) == h.natoms() and o["N"] == h.natoms() // 5
    want = oracle.energy(final, p)
    assert abs(o["energy"] - want["energy"]) < 1e-9 * max(1.0, abs(want["energy"]))
    h.close()


def test_uvt_incremental_edits_match_full_reuploads():
    """Insertions and removals reach the device as edits of the resident configuration
    (mpmc_hip_insert_molecule / remove_molecule: holes, reused slots, appended view slots).  The same chain
    with those edits disabled -- every N change a full upload -- must walk the same path: identical accept /
    reject decisions and atom counts, energies equal to rounding (the two keep different atom orders), and the
    per-atom dipoles mapped back through the slots must match the oracle."""
    s = synth.s_pol(160, spacing=4.5)
    p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, feynman_hibbs=1,
             feynman_hibbs_order=4)
    # one chain after the other: the host layer has ONE Mersenne twister per process, like the reference
    chains, trace = [], [[], []]
    for k, incremental in enumerate((1, 0)):
        h = host.HostSystem(s, p, seed=33, move_factor=0.05, rot_factor=0.05,
                            extra={"ensemble": "uvt", "insert_probability": 0.7, "pressure": 300.0})
        h.energy()  # creates the context
        h.set_option("incremental_amatrix", incremental)  # 0: the engine asks for re-uploads instead of edits
        for _ in range(15):
            acc = h.mc_steps(8)
            trace[k].append((acc, h.natoms(), h.observables()["energy"]))
        chains.append(h)
    ns = {t[1] for t in trace[0]}
    assert len(ns) > 2, "N hardly changed"
    for a, b in zip(*trace):
        assert a[0] == b[0] and a[1] == b[1]
        assert abs(a[2] - b[2]) <= 1e-10 * max(1.0, abs(b[2]))
    for h in chains:
        final = h.system(s["basis"])
        want = oracle.energy(final, p, want_vectors=True)
        o = h.observables()
        assert abs(o["energy"] - want["energy"]) < 1e-9 * max(1.0, abs(want["energy"]))
        e = h.energy()  # the device may still hold a rejected trial: evaluate the configuration the chain kept
        assert abs(e - want["energy"]) < 1e-9 * max(1.0, abs(want["energy"]))
        d = h.dipoles()
        assert np.abs(d["mu"] - want["mu"]).max() <= 1e-9 * np.abs(want["mu"]).max()
        assert np.abs(d["ef_static"] - want["ef_static"]).max() <= 1e-9 * np.abs(want["ef_static"]).max()
        h.close()


def test_uvt_gauss_seidel_edits_follow_the_list_order():
    """Grand-canonical chain under the reference's production flags (ranked Gauss-Seidel: the sweep ORDER is part of
    the result).  Insertions and removals edit the resident configuration and the host states the new list order of
    the polarizable sites (mpmc_hip_set_sweep_order); the same chain with edits disabled re-uploads the whole
    configuration in list order at every change of N.  Same decisions, same atom counts, energies equal to rounding,
    and the configuration the chain ends on matches the oracle -- which sweeps in list order like the reference."""
    s = synth.s_pol(160, spacing=4.5)
    p = dict(synth.FLAGS_POL_PRODUCTION)
    chains, trace = [], [[], []]
    for k, incremental in enumerate((1, 0)):
        h = host.HostSystem(s, p, seed=33, move_factor=0.05, rot_factor=0.05,
                            extra={"ensemble": "uvt", "insert_probability": 0.7, "pressure": 300.0})
        h.energy()  # creates the context
        h.set_option("incremental_amatrix", incremental)  # 0: the engine asks for re-uploads instead of edits
        for _ in range(12):
            acc = h.mc_steps(8)
            trace[k].append((acc, h.natoms(), h.observables()["energy"]))
        chains.append(h)
    ns = {t[1] for t in trace[0]}
    assert len(ns) > 2, "N hardly changed"
    for a, b in zip(*trace):
        assert a[0] == b[0] and a[1] == b[1]
        assert abs(a[2] - b[2]) <= 1e-10 * max(1.0, abs(b[2]))
    for h in chains:
        final = h.system(s["basis"])
        want = oracle.energy(final, p, want_vectors=True)
        e = h.energy()  # the device may still hold a rejected trial: evaluate the configuration the chain kept
        assert abs(e - want["energy"]) < 1e-9 * max(1.0, abs(want["energy"]))
        o = h.observables()
        assert abs(o["polarization_energy"] - want["polarization_energy"]) < 1e-9 * max(1e-3, abs(want["polarization_energy"]))
        d = h.dipoles()
        assert np.abs(d["mu"] - want["mu"]).max() <= 1e-9 * np.abs(want["mu"]).max()
        h.close()



def test_uvt_chain_is_the_same_with_and_without_the_resident_solver():
    """A grand-canonical chain on a small box (the view grows and shrinks across 64-atom block boundaries) with the
    dipole solve as one resident launch and with a launch pair per sweep: bit-identical results mean the SAME chain --
    every decision, every atom count, every energy to the last bit."""
    s = synth.s_pol(160, spacing=4.5)
    p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=5, polar_palmo=1, feynman_hibbs=1,
             feynman_hibbs_order=4)
    traces = []
    for resident in (1, 0):
        h = host.HostSystem(s, p, seed=57, move_factor=0.05, rot_factor=0.05,
                            extra={"ensemble": "uvt", "insert_probability": 0.7, "pressure": 400.0})
        h.energy()  # creates the context
        h.set_option("resident_jacobi", resident)
        h.enable_timing(True)  # (the host layer reads the engine's counters only then)
        tr = []
        for _ in range(20):
            acc = h.mc_steps(6)
            tr.append((acc, h.natoms(), h.observables()["energy"], h.observables()["polarization_energy"]))
        used = h.timings().get("resident_calls", 0)
        h.close()  # one context at a time: the resident kernel wants the device to itself
        traces.append((tr, used))
    (a, used_a), (b, used_b) = traces
    assert used_a > 100 and used_b <= 1  # (the call that created the second context ran before the option was set)
    assert len({t[1] for t in a}) > 2, "N hardly changed"
    assert a == b

def test_uvt_chain_without_polarization():
    """Grand-canonical chain of charged LJ dimers (no polarization): insertions / removals only touch the pair
    tiles, the reciprocal-space block partials, the long-range-correction tiles and the cached self term."""
    s = synth.s_es(256)
    p = dict(synth.FLAGS_ES)
    h = host.HostSystem(s, p, seed=17, move_factor=0.05, rot_factor=0.05,
                        extra={"ensemble": "uvt", "insert_probability": 0.7, "pressure": 2000.0})
    seen = set()
    for _ in range(20):
        h.mc_steps(20)
        seen.add(h.natoms())
    assert len(seen) > 3
    final = h.system(s["basis"])
    want = oracle.energy(final, p)
    o = h.observables()
    for key in ("energy", "rd_energy", "coulombic_energy"):
        assert abs(o[key] - want[key]) < 1e-9 * max(1.0, abs(want[key]), abs(want["rd_energy"])), key
    assert abs(h.energy() - want["energy"]) < 1e-9 * max(1.0, abs(want["rd_energy"]))
    h.close()


def test_uvt_chain_that_outgrows_its_context():
    """A box that fills up tenfold under a very high fugacity: thousands of insertions and removals as edits
    of the resident configuration (reused holes, a growing sweep view, pair / field tile grids that change
    shape), and along the way the context runs out of slots, so the host re-creates it and uploads again.
    At the end the energy the chain carries, a fresh evaluation and the oracle must agree, per-atom dipoles
    included (they come back through the slot map)."""
    s = synth.s_pol(160, spacing=6.0)
    p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, feynman_hibbs=1,
             feynman_hibbs_order=4)
    h = host.HostSystem(s, p, seed=5, move_factor=0.05, rot_factor=0.05,
                        extra={"ensemble": "uvt", "insert_probability": 0.8, "pressure": 50000.0})
    h.mc_steps(6000)
    assert h.natoms() > 160 + 80 + 1024, "the context (N + N/2 + 1024 slots) was never outgrown"
    final = h.system(s["basis"])
    want = oracle.energy(final, p, want_vectors=True)
    carried = h.observables()["energy"]
    fresh = h.energy()
    # (not bitwise: a re-upload in between re-orders the atoms, and with them the order of the tile sums)
    assert abs(carried - fresh) <= 1e-12 * abs(fresh)
    assert abs(fresh - want["energy"]) < 1e-10 * abs(want["energy"])
    d = h.dipoles()
    assert np.abs(d["mu"] - want["mu"]).max() <= 1e-10 * np.abs(want["mu"]).max()
    h.close()


def test_driver_executable_on_reference_style_input():
    """mpmc_hip <input> on the 10-atom box: the step-0 line of its energy_output must carry the
    reference's golden numbers (sample_configs_gpu/cuda_pol.small/noncuda_control/small.energy.dat:2)."""
    out = "/tmp/mpmc_hip_small.energy.dat"
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([host.EXE_PATH, os.path.join(ROOT, "tests", "data", "bssp_small", "input")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(out).read().splitlines()
    assert lines[0].startswith("#step #energy #coulombic #rd #polar")
    t = lines[1].split()
    assert t[:5] == ["0", "-22.738394", "1.993547", "-24.673480", "-0.058461"]
    assert [l.split()[0] for l in lines[1:]] == ["0", "10", "20", "30", "40"]
