"""mpmc_amd -- MI355X (gfx950) per-step energy engine for MPMC.

The product is the C-ABI shared library mpmc_amd/csrc/libmpmc_hip.so (include/mpmc_hip.h);
this package only holds its sources, a ctypes binding for tests/bench (engine.py) and the
synthetic-input generators (synth.py).
"""
