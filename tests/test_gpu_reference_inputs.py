"""The reference's real sample inputs end to end on the device: keyword file + PDB -> the
driver executable `mpmc_hip` -> reference-format energy_output.

  * tests/data/socmof: the 1228-atom In-soc-MOF + BSSP H2 run of sample_configs_gpu/cuda_pol/noncuda_control; the
    driver's step-0 line must be the reference's own (socMOF+BSSP.energy.dat:2): the polarization energy digit for
    digit, the pair sums to one unit of the last printed digit (summation order).
  * tests/data/pcn61_full: sample_configs_gpu/3_PCN61 in full -- 21 183 atoms (15 129 polarizable), `ensemble uvt`,
    4 steps as iter.inp asks.  The reference holds no output for it, so parity at this size is carried by what does
    not depend on size: the two non-polarization terms against the oracle (no O(N^2) memory needed there), the pair
    sums invariant under a lattice translation, every term invariant under a re-ordering of the molecules, the
    coefficient sweep against the expanded-matrix sweep, and the chain's carried energy after its insert / remove / displace steps
    equal to a fresh upload of the final configuration.
"""
import os
import subprocess

import numpy as np
import pytest

from mpmc_amd import engine, host, pqr
from oracle import oracle
from test_reference_inputs import DATA, SOCMOF_LINE, pcn61_dir

pytestmark = pytest.mark.gpu

PCN_BASIS = np.diag([128.388, 42.796, 42.796])
PCN_FLAGS = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, pbc_cutoff=8.0, feynman_hibbs=1,
                 feynman_hibbs_order=4)


def test_driver_reproduces_the_reference_step0_line_from_its_own_pdb():
    out = "/tmp/mpmc_hip_socmof.energy.dat"
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([host.EXE_PATH, os.path.join(DATA, "socmof", "input")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(out).read().splitlines()
    assert lines[0].startswith("#step #energy #coulombic #rd #polar")
    got = lines[1].split()[:5]
    # The reference's printed digits (%f, six decimals of numbers of 1e4 ... 1e5 K).  The device adds its tile partials in
    # another order than the reference's pair loop, and the Ewald column is a difference of parts of +-1e7 K (real-space
    # sum vs point self term): a column may round its LAST printed digit the other way (1e-11 relative).  The CPU oracle,
    # which keeps the reference's order, prints the line exactly (tests/test_reference_inputs.py).
    assert got[0] == "0"
    for g, w in zip(got[1:], SOCMOF_LINE[1:]):
        assert abs(float(g) - float(w)) < 1.5e-6, (g, w)
    assert got[4] == SOCMOF_LINE[4]  # the polarization energy: every digit
    assert lines[1].split()[8] == "156.000000"  # N: the movable molecules
    assert [l.split()[0] for l in lines[1:]] == ["0", "10", "20"]


def test_pcn61_full_runs_through_the_driver(tmp_path):
    d = pcn61_dir(tmp_path)
    out = "/tmp/mpmc_hip_pcn61_full.energy.dat"
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([host.EXE_PATH, os.path.join(d, "input")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = open(out).read().splitlines()
    assert [l.split()[0] for l in lines[1:]] == ["0", "2", "4"]
    for l in lines[1:]:
        v = [float(x) for x in l.split()[1:5]]
        assert all(np.isfinite(v)) and v[0] < 0 and v[3] < 0  # bound system, negative polarization energy
        assert abs(v[0] - (v[1] + v[2] + v[3])) < 1e-5 * abs(v[0])  # %f columns add up


def test_pcn61_full_parity_by_size_independent_properties():
    s = pqr.read_pqr(os.path.join(DATA, "pcn61_full", "input.pdb.gz"), PCN_BASIS)
    n = len(s["charge"])
    eng = engine.Engine(n)
    eng.load_system(s, PCN_FLAGS)
    r0 = eng.energy()
    # (1) LJ (+ long-range correction, Feynman-Hibbs) and Ewald real / reciprocal / self at FULL size against the oracle
    want = oracle.energy(s, dict(PCN_FLAGS, polarization=0))
    for k in ("rd_energy", "es_real", "es_recip", "es_self"):
        assert abs(r0[k] - want[k]) <= 1e-10 * max(1.0, abs(want[k])), (k, r0[k], want[k])
    # (2) the pair sums under a lattice translation of all atoms.  (Not the polarization energy: the framework has atom
    # pairs EXACTLY half a cell apart in one coordinate, where the minimum image is a tie that rint() breaks on the last
    # bit of the coordinates; the dipole tensor of such a pair -- the A matrix has no cut-off -- differs between the two
    # images in its off-diagonal elements, so a translation that re-rounds the coordinates legitimately moves U_pol in
    # the sixth digit.  The reference has the same sensitivity; tests/test_gpu_parity.py pins which image is taken.)
    shift = PCN_BASIS[0] * 1 + PCN_BASIS[1] * (-2) + PCN_BASIS[2] * 3
    eng2 = engine.Engine(n)
    eng2.load_system(dict(s, pos=s["pos"] + shift), PCN_FLAGS)
    r1 = eng2.energy()
    eng2.close()
    for k in ("rd_energy", "es_real", "es_recip", "es_self"):
        assert abs(r1[k] - r0[k]) <= 2e-9 * max(1.0, abs(r0[k])), (k, r0[k], r1[k])
    assert r0["polarization_energy"] < 0 and r0["polar_iterations"] == 4
    # (2b) every term, polarization included, with the movable molecules in another ORDER (same coordinates to the bit:
    # no tie is touched; Jacobi sweeps do not depend on the order, the tiling of every kernel does)
    mol = s["molecule"]
    nfro = int(np.flatnonzero(s["frozen"] == 0)[0])
    ids = np.unique(mol[nfro:])
    perm_ids = np.random.default_rng(11).permutation(ids)
    order = np.concatenate([np.arange(nfro)] + [np.flatnonzero(mol == m) for m in perm_ids])
    sp = {k: (v[order] if k != "basis" else v) for k, v in s.items()}
    sp["molecule"] = np.concatenate([mol[:nfro], np.repeat(np.arange(len(ids)) + mol[nfro], 5)]).astype(np.int32)
    eng3 = engine.Engine(n)
    eng3.load_system(sp, PCN_FLAGS)
    r2 = eng3.energy()
    # (2c) ... and the dipole sweep on the expanded 3N x 3N matrix (16.5 GB here) instead of the 16-byte pair coefficients
    eng3.set_option("pair_coefficients", 0)
    r3 = eng3.energy()
    eng3.close()
    for k in ("rd_energy", "es_real", "es_recip", "es_self", "polarization_energy", "energy"):
        assert abs(r2[k] - r0[k]) <= 1e-9 * max(1.0, abs(r0[k])), (k, r0[k], r2[k])
    assert abs(r3["polarization_energy"] - r2["polarization_energy"]) <= 1e-10 * abs(r2["polarization_energy"])
    # (3) single-molecule moves: incremental evaluation == fresh context, bit for bit
    rng = np.random.default_rng(3)
    pos = s["pos"].copy()
    movable = np.flatnonzero(s["frozen"] == 0)
    for step in range(3):
        a = int(rng.choice(movable))
        idx = np.flatnonzero(s["molecule"] == s["molecule"][a])
        pos[idx] += rng.normal(scale=0.2, size=3)
        eng.update_atoms(int(idx[0]), pos[idx])
        ri = eng.energy()
    fresh = engine.Engine(n)
    fresh.load_system(dict(s, pos=pos), PCN_FLAGS)
    rf = fresh.energy()
    fresh.close()
    eng.close()
    for k in ("energy", "rd_energy", "coulombic_energy", "polarization_energy"):
        assert ri[k] == rf[k], k


def test_pcn61_full_uvt_chain_carries_the_energy_of_its_configuration():
    """4 grand-canonical steps (the run iter.inp asks for) + 40 more through the host layer: insertions / removals as
    edits of the resident 21 183-atom configuration; the carried energy = a fresh evaluation of the final lists."""
    s = pqr.read_pqr(os.path.join(DATA, "pcn61_full", "input.pdb.gz"), PCN_BASIS)
    h = host.HostSystem(s, PCN_FLAGS, seed=752498, move_factor=0.001, rot_factor=0.01,
                        extra={"ensemble": "uvt", "insert_probability": 0.666, "pressure": 70.0})
    seen = set()
    for _ in range(11):
        h.mc_steps(4)
        seen.add(h.natoms())
    assert len(seen) > 1  # molecules did enter / leave
    carried = h.observables()
    final = h.system(PCN_BASIS)
    eng = engine.Engine(len(final["charge"]))
    eng.load_system(final, PCN_FLAGS)
    fresh = eng.energy()
    eng.close()
    h.close()
    for k in ("energy", "rd_energy", "coulombic_energy", "polarization_energy"):
        assert abs(carried[k] - fresh[k]) <= 1e-10 * max(1.0, abs(fresh[k])), (k, carried[k], fresh[k])
