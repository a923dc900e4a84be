"""Checks that do NOT go through the oracle's reading of the solver (known-answer tests):

* the converged dipoles of every solver variant against a direct solve of A mu = E_static, with A from
  mpmc_hip_download_amatrix and E_static from the engine's own static-field kernel;
* the fp32 pair / field screen guarded against coordinates far from the origin (the reference never wraps
  atom->pos, src/io/output.c:142-183): the 1228-atom reference golden with every sorbate molecule moved by 10^4
  lattice vectors must still give the reference's printed digits.
"""
import json
import os

import numpy as np
import pytest

from mpmc_amd import engine, synth
from oracle import oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FX = json.load(open(os.path.join(GOLD, "fixtures.json")))
DEBYE2SKA = 85.10597636  # reference src/include/defines.h

SOLVERS = {
    "jacobi": dict(),
    "jacobi_sor": dict(polar_sor=1, polar_gamma=0.8),
    "gs": dict(polar_gs=1),
    "gs_ranked": dict(polar_gs_ranked=1),
    "gs_ranked_palmo_gamma": dict(polar_gs_ranked=1, polar_palmo=1, polar_gamma=1.03),
}


@pytest.mark.parametrize("solver", sorted(SOLVERS))
def test_converged_dipoles_solve_the_linear_system(solver):
    """polar_precision 1e-10 Debye: the SCF fixed point is the solution of A mu = E_static (thole_iterative.c is an
    iteration for exactly that; polar.c:91-95 solves it by inversion when polar_iterative is off).  Independent of
    how the oracle reads the sweep: only A, E_static and numpy.linalg.solve."""
    s = synth.s_pol(640)  # 384 polarizable sites = 6 blocks of 64: the multi-block Gauss-Seidel path
    prec = 1e-10
    p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=0, polar_precision=prec)
    p.update(SOLVERS[solver])
    eng = engine.Engine(640)
    try:
        eng.load_system(s, p)
        r = eng.energy()
        d = eng.dipoles()
        A = eng.amatrix()
    finally:
        eng.close()
    assert r["iter_success"] == 0 and 3 < r["polar_iterations"] < 128
    pol = np.repeat(s["alpha"] != 0.0, 3)
    # non-polarizable sites carry no dipole (thole_iterative.c:34-39); the polarizable block is a closed system
    mu_direct = np.zeros(3 * 640)
    mu_direct[pol] = np.linalg.solve(A[np.ix_(pol, pol)], d["ef_static"].reshape(-1)[pol])
    mu = d["mu"].reshape(-1)
    assert np.all(mu[~pol] == 0.0)
    allowed = prec * DEBYE2SKA
    # the stopping rule bounds the last CHANGE by `allowed`; the distance to the fixed point is a small multiple of it
    assert np.abs(mu - mu_direct).max() <= 20.0 * allowed, (solver, np.abs(mu - mu_direct).max(), allowed)
    assert np.abs(mu_direct).max() > 1e6 * allowed  # the comparison resolves 6+ digits of the dipoles
    # U_pol = -1/2 mu . E_static (polar.c:107-116); the Palmo term mu . dE_ind vanishes at the fixed point
    upol = -0.5 * float(mu_direct @ d["ef_static"].reshape(-1))
    assert abs(r["polarization_energy"] - upol) <= 1e-7 * abs(upol)


def shifted(s, nbox):
    """every movable molecule moved by its own multiple (~nbox) of lattice vectors"""
    s2 = dict(s)
    pos = s["pos"].copy()
    rng = np.random.default_rng(11)
    mol = np.asarray(s["molecule"])
    for m in np.unique(mol[np.asarray(s["frozen"]) == 0]):
        k = rng.integers(-nbox, nbox + 1, size=3).astype(np.float64)
        k[rng.integers(3)] = float(nbox)  # at least one component at the full distance
        pos[mol == m] += k @ s["basis"]
    s2["pos"] = pos
    return s2


@pytest.mark.parametrize("nbox", [40, 10000])
def test_screen_survives_coordinates_far_from_the_origin(nbox):
    """|x| up to ~1e3 A stays on the fp32 screen (inside its stated bound), ~2e5 A switches to the fp64 screen.
    Moving molecules by lattice vectors changes no energy: reference digits (the stored coordinates carry
    |x| * 2^-53 of rounding, ~3e-11 A at 10^4 boxes, far below the printed digits) and the oracle on the very
    same shifted coordinates."""
    name = "socmof_bssp_1228"
    fx = FX[name]
    s = shifted(dict(np.load(os.path.join(GOLD, name + ".npz"))), nbox)
    assert (np.abs(s["pos"]).max() > 2048.0) == (nbox == 10000)
    eng = engine.Engine(1228)
    try:
        eng.load_system(s, fx["params"])
        got = eng.energy()
        key = {"energy": "energy", "coulombic": "coulombic_energy", "rd": "rd_energy", "polar": "polarization_energy"}
        for k, want in fx["expected"].items():
            if k in key:
                tol = max(0.5000001 * 10 ** (-fx["decimals"]), 1e-9 * abs(want))
                assert abs(got[key[k]] - want) <= tol, (k, got[key[k]], want)
        want = oracle.energy(s, fx["params"])
        for k in ("rd_energy", "es_real", "polarization_energy"):
            assert abs(got[k] - want[k]) <= 1e-10 * abs(want[k]), (k, got[k], want[k])
        # a move that crosses the bound switches the screen at update time, not only at upload
        first = 1228 - 5
        far = s["pos"][first:] + np.array([7.0, -3.0, 5.0]) * 1e4 @ s["basis"] + np.array([0.2, -0.1, 0.15])
        eng.update_atoms(first, far)
        e1 = eng.energy()
        s3 = dict(s)
        s3["pos"] = s["pos"].copy()
        s3["pos"][first:] = far
        w1 = oracle.energy(s3, fx["params"])
        for k in ("rd_energy", "es_real", "polarization_energy"):
            assert abs(e1[k] - w1[k]) <= 1e-10 * abs(w1[k]), (k, e1[k], w1[k])
    finally:
        eng.close()
