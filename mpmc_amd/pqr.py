"""The reference's whitespace PQR / PDB input format (src/io/read_pqr.c:155-389) as arrays, for bench.py and the tests:
`ATOM id type molecule-type F|M molecule-id x y z mass charge alpha epsilon sigma [...]`; lines whose molecule type
is BOX are cell markers the reference skips (read_pqr.c:221-224); charges are stored in sqrt(K A) (e * 408.7816,
read_pqr.c:249).  The C reader of the host layer (mpmc_amd/host/input.c) is what the driver uses; this is the same
rule set in numpy, checked against it in tests/test_reference_inputs.py."""
import gzip

import numpy as np

E2REDUCED = 408.7816


def read_pqr(path, basis):
    op = gzip.open if str(path).endswith(".gz") else open
    pos, mass, q, al, ep, sg, mol, frz = [], [], [], [], [], [], [], []
    last, mi = None, -1
    with op(path, "rt") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0].upper().startswith("END"):
                break
            if t[0].upper() != "ATOM" or t[3].upper() == "BOX":
                continue
            if last != t[5]:
                last, mi = t[5], mi + 1
            pos.append([float(t[6]), float(t[7]), float(t[8])])
            mass.append(float(t[9]))
            q.append(float(t[10]) * E2REDUCED)
            al.append(float(t[11]))
            ep.append(float(t[12]))
            sg.append(float(t[13]))
            mol.append(mi)
            frz.append(1 if t[4].upper() == "F" else 0)
    return dict(pos=np.array(pos), mass=np.array(mass), charge=np.array(q), alpha=np.array(al), epsilon=np.array(ep),
                sigma=np.array(sg), molecule=np.array(mol, dtype=np.int32), frozen=np.array(frz, dtype=np.int32),
                basis=np.asarray(basis, dtype=np.float64).reshape(3, 3))
