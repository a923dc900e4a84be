"""Long chains through the C host layer on every solver path (resident Jacobi launch, launch-per-sweep, ranked
Gauss-Seidel chain, grand-canonical edits under both): step rate, and the counters that would show a lost hand-off (resident
fallbacks) or a repeated speculative call.  python tools/soak.py  (about a minute on an MI355X)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from mpmc_amd import host, synth
def run(name, s, p, steps, extra=None):
    h = host.HostSystem(s, p, seed=7, extra=extra)
    h.energy()
    for kv in os.environ.get("SOAK_OPTS", "").split(","):  # e.g. SOAK_OPTS=rank_view_side=0,split_record=0 (A/B: same energies)
        if "=" in kv:
            h.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    h.enable_timing(True)
    h.set_option("timing", 0)
    t0 = time.time()
    done = 0
    while done < steps:
        h.mc_steps(1000)
        done += 1000
    dt = time.time() - t0
    t = h.timings()
    o = h.observables()
    print(name, "steps", steps, "%.0f steps/s" % (steps / dt), "resident_calls", t["resident_calls"], "fallbacks", t["resident_fallbacks"],
          "spec_redos", t["spec_rank_redos"], "E %.6f" % o["energy"], flush=True)
    h.close()
only = os.environ.get("SOAK_ONLY", "")
_run = run
def run(name, *a, **k):
    if only and only not in name:
        return
    _run(name, *a, **k)
run("S-POL(1024) jacobi10", synth.s_pol(1024), dict(synth.FLAGS_POL_JACOBI), 150000)
run("S-POL(320) palmo sor", synth.s_pol(320), dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=6, polar_sor=1, polar_gamma=0.8, polar_palmo=1), 100000)
run("S-POL(1024) uvt", synth.s_pol(1024), dict(synth.FLAGS_POL_JACOBI), 50000, extra={"ensemble": "uvt", "insert_probability": 0.5, "pressure": 100.0})
run("S-POL(4096) production", synth.s_pol(4096), dict(synth.FLAGS_POL_PRODUCTION), 20000)
run("S-POL(1024) production uvt", synth.s_pol(1024), dict(synth.FLAGS_POL_PRODUCTION), 30000, extra={"ensemble": "uvt", "insert_probability": 0.5, "pressure": 100.0})
run("S-POL(4096) production uvt", synth.s_pol(4096), dict(synth.FLAGS_POL_PRODUCTION), 10000, extra={"ensemble": "uvt", "insert_probability": 0.5, "pressure": 100.0})
pc = dict(np.load("tests/golden/pcn61_bssp_4096.npz"))
run("PCN-61", pc, dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, pbc_cutoff=8.0, feynman_hibbs=1, feynman_hibbs_order=4), 60000)
