"""Timing-only ablations of gs_chain_kernel on ONE box (option gs_ablate; the results of ablated sweeps are wrong by
construction): which part of a block's ~2.6 us is the source loop, which the critical section, which the hand-off.
bits: 1 no source loop, 2 no tile loads, 4 no polls in the loop, 8 critical section only republishes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpmc_amd import engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
masks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 4, 6, 8, 9, 10, 12, 14]
s = synth.s_pol(n)
p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_gs=1, polar_max_iter=4)
e = engine.Engine(n)
if os.environ.get("MPMC_GS_LAGS"):
    e.set_option("gs_lags", int(os.environ["MPMC_GS_LAGS"]))
e.load_system(s, p)
e.energy()
pos = s["pos"].copy()
rng = np.random.default_rng(1)
e.set_option("timing", 1)
e.set_option("timing_interval", 1)
res = {m: [] for m in masks}
for rnd in range(5):
    for m in masks:
        e.set_option("gs_ablate", m)
        for k in range(4):
            a = 5 * int(rng.integers(0, n // 5))
            e.update_atoms(a, pos[a:a + 5] + rng.normal(scale=0.05, size=3))
            try:
                e.energy()
            except engine.EngineError as ex:
                print("mask", m, "error:", str(ex)[:80])
                break
            t = e.timings()
            if k >= 1 and t["sweep_count"]:
                res[m].append(1e3 * t["sweep_ms"] / t["sweep_count"])
            e.update_atoms(a, pos[a:a + 5])
e.set_option("gs_ablate", 0)
for m in masks:
    a = np.array(res[m]) if res[m] else np.array([float("nan")])
    print("ablate %2d: chain kernel %.1f us mean, %.1f min, n=%d" % (m, a.mean(), a.min(), len(a)))
e.close()
