#!/usr/bin/env python3
"""Fixture G7: the snapshots of the reference's scaling test
sample_configs/inputs/012-3D-crystal-replay/replay.pqr -- a crystal of two-site polarizable molecules in
sheared cells of 1^3 ... 8^3 unit cells (2 ... 1024 atoms) -- and the keyword sets of its five input files.

Run in the build container only (needs /root/reference); the GPU box uses the committed outputs.  Only DATA
files of the reference are read (a PQR trajectory and keyword files); the reference holds no outputs for this
test (its energy.* files are not checked in): what it checks by eye is that energy / N is size-independent
(scale.sh), which tests/test_crystal_replay.py asserts.

Outputs: crystal_replay_012.npz (per snapshot k: pos_k, basis_k + per-atom parameters) and
crystal_replay_012.json (flag sets in C-ABI naming, provenance).
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference/sample_configs/inputs/012-3D-crystal-replay"
OUT = os.path.dirname(os.path.abspath(__file__))
E2REDUCED = 408.7816  # reference src/include/defines.h:45, applied in src/io/read_pqr.c:249


def frames(path):
    atoms, basis = [], {}
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "ATOM":
            atoms.append(t)
        elif t[0] == "REMARK" and t[1] == "BOX":
            basis[int(t[2][6])] = [float(t[4]), float(t[5]), float(t[6])]
        elif t[0] == "END":
            yield atoms, [basis[0], basis[1], basis[2]]
            atoms, basis = [], {}


def main():
    arrays, sizes = {}, []
    for k, (atoms, basis) in enumerate(frames(os.path.join(REF, "replay.pqr"))):
        # ATOM id type moltype F|M molid x y z mass charge alpha epsilon sigma ... (read_pqr.c:201-333)
        arrays["pos_%d" % k] = np.array([[float(a[6]), float(a[7]), float(a[8])] for a in atoms])
        arrays["basis_%d" % k] = np.array(basis)
        arrays["molecule_%d" % k] = np.array([int(a[5]) for a in atoms], dtype=np.int32)
        arrays["charge_%d" % k] = np.array([float(a[10]) * E2REDUCED for a in atoms])
        assert all(a[4] == "M" for a in atoms)
        par = {(float(a[9]), float(a[11]), float(a[12]), float(a[13])) for a in atoms}
        assert par == {(1.0, 0.1, 0.1, 0.6)}, par  # mass alpha epsilon sigma: the same for every site
        sizes.append(len(atoms))
    np.savez_compressed(os.path.join(OUT, "crystal_replay_012.npz"), **arrays)
    common = dict(polarization=1, polar_damp=2.1304, polar_palmo=1, polar_gamma=1.03, rd_lrc=1)
    ranked = dict(polar_gs_ranked=1, polar_precision=1e-5, polar_max_iter=0)
    flagsets = {
        "polar.in": dict(common, **ranked),
        "polar_wolf.in": dict(common, polar_wolf=1, **ranked),
        "polar_wolf_alpha.in": dict(common, polar_wolf=1, polar_wolf_alpha=0.13, **ranked),
        "polar_ewald.in": dict(common, polar_ewald=1, polar_ewald_alpha_set=1, polar_ewald_alpha=0.15, polar_max_iter=5),
        "wolf_wolf.in": dict(common, wolf=1, polar_wolf=1, **ranked),
    }
    meta = dict(
        source="reference sample_configs/inputs/012-3D-crystal-replay/replay.pqr (7 snapshots) and the keywords of "
               "polar.in / polar_wolf.in / polar_wolf_alpha.in / polar_ewald.in / wolf_wolf.in",
        kind="reference-input (no reference output exists for this test; invariant = energy / N, scale.sh)",
        not_carried="rd_crystal / rd_crystal_order (image sums of the repulsion-dispersion term: out of scope, the rd "
                    "column is not compared); wrapall (output only); temperature is not set by these inputs",
        atoms_per_snapshot=sizes, site=dict(mass=1.0, alpha=0.1, epsilon=0.1, sigma=0.6),
        flagsets=flagsets)
    json.dump(meta, open(os.path.join(OUT, "crystal_replay_012.json"), "w"), indent=1, sort_keys=True)
    print(sizes)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; the fixture is already committed")
    main()
