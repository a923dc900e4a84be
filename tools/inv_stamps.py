"""Where gs_block_inverse_kernel's time goes (option inv_stamps): a few single-molecule moves under the production flags."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpmc_amd import engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = synth.s_pol(n)
p = dict(synth.FLAGS_POL_PRODUCTION)
e = engine.Engine(n)
e.load_system(s, p)
e.energy()
pos = s["pos"].copy()
for k in range(3):
    e.update_atoms(5 * k, pos[5 * k:5 * k + 5] + 0.05)
    e.energy()
e.set_option("inv_stamps", 6)
for k in range(6):
    a = 5 * (37 + 61 * k)
    e.update_atoms(a, pos[a:a + 5] + 0.05)
    e.energy()
e.close()
