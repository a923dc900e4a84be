"""Large-view check of the Gauss-Seidel chain kernel (S-POL(16384): 9830 polarizable sites, 154 blocks): the chain
kernel (cached block inverses) against the literal forward substitution on the expanded matrix (persistent_gs = 0),
production flags, plus the step rate.  The CPU oracle is out of reach at this size (a 19 GB matrix)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpmc_amd import engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
s = synth.s_pol(n)
p = dict(synth.FLAGS_POL_PRODUCTION)
res = []
for persistent in (1, 0):
    e = engine.Engine(n)
    e.load_system(s, p)
    e.set_option("persistent_gs", persistent)
    t0 = time.time()
    r = e.energy()
    r.update(e.dipoles())
    print("persistent_gs", persistent, "U_pol %.12e" % r["polarization_energy"], "first call %.2f s" % (time.time() - t0))
    if persistent:
        pos = s["pos"].copy()
        rng = np.random.default_rng(3)
        t0 = time.time()
        for k in range(30):
            m = 5 * int(rng.integers(0, n // 5))
            e.update_atoms(m, pos[m:m + 5] + rng.normal(scale=0.05, size=3))
            e.energy()
            e.update_atoms(m, pos[m:m + 5])
        print("  %.1f energy() calls per second (moves + restores)" % (30 / (time.time() - t0)))
        again = e.energy()
        assert again["polarization_energy"] == r["polarization_energy"], "restored configuration differs"
    res.append(r)
    e.close()
scale = np.abs(res[1]["mu"]).max()
print("max |mu - mu_literal| / max|mu| = %.2e" % (np.abs(res[0]["mu"] - res[1]["mu"]).max() / scale))
print("rel dU_pol = %.2e" % (abs(res[0]["polarization_energy"] - res[1]["polarization_energy"]) / abs(res[1]["polarization_energy"])))
