"""A/B of engine options on the incremental energy() path: rate and mean sweep-kernel time (kernel-own timestamps).
    python tools/sweep_policy.py <workload> <option> <value> [<value> ...]
workload: pcn61 | <n> (S-POL(n), Jacobi x10)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpmc_amd import engine, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, opt = sys.argv[1], sys.argv[2]
values = [int(v) for v in sys.argv[3:]]
p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, feynman_hibbs=1, feynman_hibbs_order=4)
if name == "pcn61":
    s = dict(np.load(os.path.join(ROOT, "tests", "golden", "pcn61_bssp_4096.npz")))
    p.update(polar_max_iter=4, pbc_cutoff=8.0)
else:
    s = synth.s_pol(int(name))
    p.update(polar_max_iter=10)
n = len(s["charge"])
mol = s["molecule"]
movable = np.where(~s["frozen"].astype(bool))[0]
rng = np.random.default_rng(1)
firsts = []
for a in rng.choice(movable, size=600):
    idx = np.where(mol == mol[a])[0]
    firsts.append((int(idx[0]), len(idx)))
pos = s["pos"].copy()
for rep in range(2):
    for v in values:
        e = engine.Engine(n)
        e.load_system(s, p)
        e.set_option(opt, v)
        e.set_option("timing", 0)
        e.energy()
        r2 = np.random.default_rng(2)
        def run(lo, hi):
            for a, cnt in firsts[lo:hi]:
                e.update_atoms(a, pos[a:a + cnt] + r2.normal(scale=0.05, size=3))
                e.energy()
        run(0, 100)
        t0 = time.perf_counter()
        run(100, 600)
        dt = time.perf_counter() - t0
        e.set_option("timing", 1)
        e.set_option("timing_interval", 1)
        sw, cnt = 0.0, 0
        for a, c in firsts[:40]:
            e.update_atoms(a, pos[a:a + c] + r2.normal(scale=0.05, size=3))
            last = e.energy()
            t = e.timings()
            sw += t["sweep_ms"]
            cnt += t["sweep_count"]
        print("%s %s=%d: %.0f energy()/s (%.1f us), sweep launch %.2f us (n=%d)   U_pol %r" %
              (name, opt, v, 500 / dt, 2000 * dt, 1000 * sw / max(1, cnt), cnt, last["polarization_energy"]), flush=True)
        e.close()
