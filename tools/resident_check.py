"""Resident Jacobi solver (kernels_resident.h) against the multi-launch path: bit-identity of every output over the
solver variants and sizes, a move / restore sequence, and the energy() rate of both.  Run on the GPU box:
    python tools/resident_check.py [quick]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mpmc_amd import engine, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"

VARIANTS = {
    "jacobi4": dict(polar_max_iter=4),
    "jacobi1": dict(polar_max_iter=1),
    "jacobi10_gamma": dict(polar_max_iter=10, polar_gamma=1.03),
    "sor": dict(polar_max_iter=6, polar_sor=1, polar_gamma=0.8),
    "esor": dict(polar_max_iter=6, polar_esor=1, polar_gamma=0.9),
    "palmo": dict(polar_max_iter=4, polar_palmo=1),
    "palmo_rrms": dict(polar_max_iter=5, polar_palmo=1, polar_rrms=1),
    "wolf": dict(polar_max_iter=4, polar_wolf=1, polar_wolf_alpha=0.13),
}
KEYS = ("energy", "polarization_energy", "dipole_rrms", "rd_energy", "coulombic_energy")
VEC = ("mu", "ef_induced", "ef_induced_change")


def systems():
    out = [("S-POL(1024)", synth.s_pol(1024)), ("S-POL(4096)", synth.s_pol(4096))]
    f = os.path.join(ROOT, "tests", "golden", "pcn61_bssp_4096.npz")
    if os.path.exists(f):
        out.append(("PCN-61(4096)", dict(np.load(f))))
    if not quick:
        out.insert(0, ("S-POL(320)", synth.s_pol(320)))
        out.insert(0, ("S-POL(40)", synth.s_pol(40)))
    return out


def one(s, p, resident, moves=0, seed=5):
    n = len(s["charge"])
    e = engine.Engine(n)
    try:
        e.load_system(s, p)
        e.set_option("resident_jacobi", resident)
        r = e.energy()
        r.update(e.dipoles())
        hist = []
        rng = np.random.default_rng(seed)
        pos = s["pos"].copy()
        movable = np.where(~s["frozen"].astype(bool))[0] if "frozen" in s else np.arange(n)
        mol = s["molecule"]
        for k in range(moves):
            a = int(rng.choice(movable))
            if mol is not None:
                idx = np.where(mol == mol[a])[0]
                a, cnt = int(idx[0]), len(idx)
            else:
                cnt = 1
            e.update_atoms(a, pos[a:a + cnt] + rng.normal(scale=0.1, size=3))
            rr = e.energy()
            hist.append(rr["energy"])
            if k % 2:
                e.update_atoms(a, pos[a:a + cnt])
        t = e.timings()
        return r, hist, t
    finally:
        e.close()


bad = 0
for name, s in systems():
    for vn, flags in VARIANTS.items():
        if quick and vn not in ("jacobi4", "palmo_rrms", "sor"):
            continue
        p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, feynman_hibbs=1, feynman_hibbs_order=4)
        if name.startswith("PCN"):
            p["pbc_cutoff"] = 8.0
        p.update(flags)
        a, ha, ta = one(s, p, 1, moves=6)
        b, hb, tb = one(s, p, 0, moves=6)
        same = all(a[k] == b[k] for k in KEYS) and all(np.array_equal(a[k], b[k]) for k in VEC) and ha == hb
        used = ta["resident_calls"]
        print("%-13s %-15s resident calls %d fallbacks %d  bit-identical: %s   U_pol %.12e" %
              (name, vn, used, ta["resident_fallbacks"], same, a["polarization_energy"]), flush=True)
        eligible = not name.startswith("PCN") and not name.startswith("S-POL(4096)")  # views of up to 21 blocks
        if not same or (eligible and used == 0) or ta["resident_fallbacks"]:
            bad += 1
            for k in KEYS:
                if a[k] != b[k]:
                    print("    ", k, a[k], b[k])
            for k in VEC:
                if not np.array_equal(a[k], b[k]):
                    print("    ", k, "max diff", np.abs(a[k] - b[k]).max())

# ---- rates (energy() after a single-molecule move, everything else resident)
for name, s in systems():
    if name.startswith("S-POL(40)") or name.startswith("S-POL(320)"):
        continue
    p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, feynman_hibbs=1, feynman_hibbs_order=4,
             polar_max_iter=4 if name.startswith("PCN") else 10)
    if name.startswith("PCN"):
        p["pbc_cutoff"] = 8.0
    n = len(s["charge"])
    for resident, side in ((1, 0), (1, 1), (0, 0)):
        e = engine.Engine(n)
        e.load_system(s, p)
        e.set_option("resident_jacobi", resident)
        e.set_option("resident_side", side)
        e.set_option("timing", 0)
        e.energy()
        rng = np.random.default_rng(1)
        pos = s["pos"].copy()
        movable = np.where(~s["frozen"].astype(bool))[0] if "frozen" in s else np.arange(n)
        mol = s["molecule"]
        firsts = []
        for a in rng.choice(movable, size=400):
            idx = np.where(mol == mol[a])[0]
            firsts.append((int(idx[0]), len(idx)))
        def run(m):
            for a, cnt in firsts[:m]:
                e.update_atoms(a, pos[a:a + cnt] + rng.normal(scale=0.05, size=3))
                e.energy()
        run(50)
        t0 = time.perf_counter()
        run(400)
        dt = time.perf_counter() - t0
        print("%-13s resident %d side_first %d: %.0f energy() calls/s (%.1f us)" % (name, resident, side, 400 / dt, 1e6 * dt / 400),
              flush=True)
        e.close()
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
