"""Time line of the resident Jacobi launch (option "resident_stamps"): python tools/resident_stamps.py [workload] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpmc_amd import engine, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "pcn61"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, feynman_hibbs=1, feynman_hibbs_order=4)
if name == "pcn61":
    s = dict(np.load(os.path.join(ROOT, "tests", "golden", "pcn61_bssp_4096.npz")))
    p.update(polar_max_iter=4, pbc_cutoff=8.0)
else:
    s = synth.s_pol(int(name))
    p.update(polar_max_iter=10)
e = engine.Engine(len(s["charge"]))
e.load_system(s, p)
e.set_option("resident_jacobi", int(os.environ.get("RESIDENT", "1")))
if os.environ.get("RESIDENT_FOLD"):
    e.set_option("resident_fold", int(os.environ["RESIDENT_FOLD"]))  # 0 = with finisher workgroups
for k in range(3):
    e.energy()
e.set_option("resident_stamps", launches)
for k in range(launches):
    e.energy()
e.close()
