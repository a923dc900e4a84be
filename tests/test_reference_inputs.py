"""The reference's real sample inputs through this repository's readers.  CPU part: the C
reader of the host layer (mpmc_amd/host/input.c: keyword file + PQR/PDB) on the two systems whose input data the
reference ships -- tests/data/socmof (1228 atoms + 8 BOX marker lines; the reference holds its step-0 energies) and
tests/data/pcn61_full (21 183 atoms, uvt; inputs only) -- against the numpy restatement of the format, and the oracle
on the 1228-atom one (its golden line, digit for digit: the PDB path, not the .npz of it).  GPU part:
tests/test_gpu_reference_inputs.py."""
import gzip
import os
import shutil

import numpy as np
import pytest

from mpmc_amd import host, pqr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "data")
SOCMOF_LINE = ["0", "-129043.736570", "-8861.486645", "-93916.094138", "-26266.155787"]  # socMOF+BSSP.energy.dat:2


def pcn61_dir(tmp_path):
    """the 21 183-atom input unpacked beside its keyword file"""
    d = tmp_path / "pcn61_full"
    d.mkdir()
    shutil.copy(os.path.join(DATA, "pcn61_full", "input"), d / "input")
    with gzip.open(os.path.join(DATA, "pcn61_full", "input.pdb.gz"), "rb") as f, open(d / "input.pdb", "wb") as g:
        shutil.copyfileobj(f, g)
    return str(d)


def host_arrays(path):
    lib = host.load()
    ptr = lib.setup_system(path.encode())
    assert ptr, "setup_system failed for " + path
    n = lib.host_natoms(ptr)
    f = {k: np.zeros(n) for k in ("charge", "alpha", "epsilon", "sigma", "mass")}
    pos = np.zeros((n, 3))
    mol = np.zeros(n, dtype=np.int32)
    frz = np.zeros(n, dtype=np.int32)
    lib.host_get_system(ptr, pos.ctypes.data, f["charge"].ctypes.data, f["alpha"].ctypes.data, f["epsilon"].ctypes.data,
                        f["sigma"].ctypes.data, f["mass"].ctypes.data, mol.ctypes.data, frz.ctypes.data)
    lib.free_system(ptr)
    return dict(pos=pos, molecule=mol, frozen=frz, **f)


def test_socmof_pdb_through_the_c_reader_and_the_oracle():
    got = host_arrays(os.path.join(DATA, "socmof", "input"))
    want = pqr.read_pqr(os.path.join(DATA, "socmof", "socMOF+BSSP.initial.pdb"), 22.4567 * np.eye(3))
    assert len(got["charge"]) == 1228  # 1236 ATOM lines, 8 of them BOX markers
    assert got["molecule"].max() + 1 == 157 and got["frozen"].sum() == 448
    for k in ("pos", "charge", "alpha", "epsilon", "sigma", "mass", "molecule", "frozen"):
        assert np.array_equal(got[k], want[k]), k
    # the oracle on what the reader produced: the reference's own step-0 line to every printed digit
    from oracle import oracle

    flags = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=10, feynman_hibbs=1, feynman_hibbs_order=4)
    e = oracle.energy(dict(got, basis=22.4567 * np.eye(3)), flags)
    line = ["0"] + ["%f" % e[k] for k in ("energy", "coulombic_energy", "rd_energy", "polarization_energy")]
    assert line == SOCMOF_LINE


def test_pcn61_full_pdb_through_the_c_reader(tmp_path):
    d = pcn61_dir(tmp_path)
    got = host_arrays(os.path.join(d, "input"))
    want = pqr.read_pqr(os.path.join(DATA, "pcn61_full", "input.pdb.gz"), np.diag([128.388, 42.796, 42.796]))
    n = len(got["charge"])
    assert n == 21183 and int(got["frozen"].sum()) == 6048
    assert got["molecule"].max() + 1 == 3 + 3027  # three framework cells (2 016 atoms each) + 3 027 five-site H2
    assert int(np.count_nonzero(got["alpha"])) == 6048 + 3 * 3027
    for k in ("pos", "charge", "alpha", "epsilon", "sigma", "mass", "molecule", "frozen"):
        assert np.array_equal(got[k], want[k]), k
