import os
import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from mpmc_amd import engine, synth
s = synth.s_pol(4096)
p = dict(synth.FLAGS_POL_PRODUCTION)
e = engine.Engine(4096)
if os.environ.get("MPMC_GS_LAGS"):
    e.set_option("gs_lags", int(os.environ["MPMC_GS_LAGS"]))
e.load_system(s, p)
e.energy()
pos = s["pos"].copy()
for k in range(3):
    e.update_atoms(5*k, pos[5*k:5*k+5] + 0.05)
    e.energy()
e.set_option("gs_stamps", 4)
e.update_atoms(50, pos[50:55] + 0.05)
e.energy()
e.close()
