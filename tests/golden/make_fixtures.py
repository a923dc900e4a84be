#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from DATA files of the reference.

Run in the build container only (needs /root/reference); the GPU box uses the
committed outputs.  Nothing of the reference's SOURCE is read or copied: inputs
are the reference's sample PQR/PDB geometry files and config keyword files, and
expected values are the numbers the reference itself wrote into its checked-in
run outputs (or, where marked `survey`, the numbers the survey stage recorded in
SURVEY.md section 8c from running the reference binary in this container).

Outputs:
  <name>.npz       arrays: pos charge alpha epsilon sigma mass molecule frozen basis
  fixtures.json    params (reference config keywords), expected energies, provenance
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
E2REDUCED = 408.7816  # reference src/include/defines.h:45, applied in src/io/read_pqr.c:249


def read_pqr(path):
    """Whitespace PQR as the reference reads it (src/io/read_pqr.c:201-333):
    ATOM id type moltype F|M molid x y z mass charge alpha epsilon sigma [omega gwp_alpha c6 c8 c10 c9]"""
    rows = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0].upper().startswith("END"):
                break
            if t[0].upper() != "ATOM" or t[3].upper() == "BOX":
                continue
            t = t + ["0"] * (20 - len(t))
            rows.append(
                dict(
                    atomtype=t[2],
                    moltype=t[3],
                    frozen=1 if t[4].upper() == "F" else 0,
                    molecule=int(t[5]),
                    pos=[float(t[6]), float(t[7]), float(t[8])],
                    mass=float(t[9]),
                    charge=float(t[10]) * E2REDUCED,
                    alpha=float(t[11]),
                    epsilon=float(t[12]),
                    sigma=float(t[13]),
                )
            )
    return rows


def rows_to_arrays(rows, basis):
    return dict(
        pos=np.array([r["pos"] for r in rows], dtype=np.float64),
        charge=np.array([r["charge"] for r in rows], dtype=np.float64),
        alpha=np.array([r["alpha"] for r in rows], dtype=np.float64),
        epsilon=np.array([r["epsilon"] for r in rows], dtype=np.float64),
        sigma=np.array([r["sigma"] for r in rows], dtype=np.float64),
        mass=np.array([r["mass"] for r in rows], dtype=np.float64),
        molecule=np.array([r["molecule"] for r in rows], dtype=np.int32),
        frozen=np.array([r["frozen"] for r in rows], dtype=np.int32),
        basis=np.array(basis, dtype=np.float64).reshape(3, 3),
    )


def read_energy_dat_line(path, lineno):
    """energy_output columns (reference src/io/output.c:988-1006):
    #step #energy #coulombic #rd #polar #vdw #kinetic #kin_temp #N #spin_ratio #volume #core_temp"""
    with open(path) as f:
        lines = f.read().splitlines()
    t = lines[lineno - 1].split()
    return dict(step=int(t[0]), energy=float(t[1]), coulombic=float(t[2]), rd=float(t[3]), polar=float(t[4]),
                N=float(t[8]), volume=float(t[10]))


POLAR_JACOBI10 = dict(
    temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=10,
    feynman_hibbs=1, feynman_hibbs_order=4,
)


def main():
    fixtures = {}

    # G1: 10-atom polarizable NVT box (two BSSP H2)
    d = f"{REF}/sample_configs_gpu/cuda_pol.small/noncuda_control"
    rows = read_pqr(f"{d}/small.initial.pdb")
    basis = [[22.4567, 0, 0], [0, 22.4567, 0], [0, 0, 22.4567]]
    np.savez_compressed(f"{OUT}/bssp_small_10.npz", **rows_to_arrays(rows, basis))
    fixtures["bssp_small_10"] = dict(
        n=len(rows), params=POLAR_JACOBI10,
        expected=read_energy_dat_line(f"{d}/small.energy.dat", 2), decimals=6,
        source="reference run output sample_configs_gpu/cuda_pol.small/noncuda_control/small.energy.dat:2 "
               "(input small.initial.pdb, config `input`)", kind="reference-output")

    # G5: In-soc-MOF + 156 BSSP H2, N = 1236
    d = f"{REF}/sample_configs_gpu/cuda_pol/noncuda_control"
    rows = read_pqr(f"{d}/socMOF+BSSP.initial.pdb")
    np.savez_compressed(f"{OUT}/socmof_bssp_1228.npz", **rows_to_arrays(rows, basis))
    fixtures["socmof_bssp_1228"] = dict(
        n=len(rows), params=POLAR_JACOBI10,
        expected=read_energy_dat_line(f"{d}/socMOF+BSSP.energy.dat", 2), decimals=6,
        source="reference run output sample_configs_gpu/cuda_pol/noncuda_control/socMOF+BSSP.energy.dat:2 "
               "(= output:51-54)", kind="reference-output")

    # G2-G4: MOF-5 + one H2 (Buch / BSS / BSSP); expected values recorded by the survey stage
    basis5 = [[25.669, 0, 0], [0, 25.669, 0], [0, 0, 25.669]]
    d = f"{REF}/sample_configs/inputs"
    rows = read_pqr(f"{d}/001-h2_buch_bulk_MOF-5/input.pqr")
    np.savez_compressed(f"{OUT}/mof5_buch_425.npz", **rows_to_arrays(rows, basis5))
    fixtures["mof5_buch_425"] = dict(
        n=len(rows), params=dict(temperature=77.0, rd_only=1),
        expected=dict(rd=-59.57861), decimals=5,
        source="SURVEY.md 8c G2: reference binary run by the survey stage on sample_configs/inputs/001 as NVT",
        kind="survey")
    rows = read_pqr(f"{d}/002-h2_bss_bulk_MOF-5/input.pqr")
    np.savez_compressed(f"{OUT}/mof5_bss_429.npz", **rows_to_arrays(rows, basis5))
    fixtures["mof5_bss_429"] = dict(
        n=len(rows), params=dict(temperature=77.0),
        expected=dict(coulombic=2.94125, rd=-61.08768), decimals=5,
        source="SURVEY.md 8c G3: reference binary run by the survey stage on sample_configs/inputs/002 as NVT",
        kind="survey")
    rows = read_pqr(f"{d}/003-h2_bssp_bulk_MOF-5/input.pqr")
    np.savez_compressed(f"{OUT}/mof5_bssp_429.npz", **rows_to_arrays(rows, basis5))
    fixtures["mof5_bssp_429"] = dict(
        n=len(rows),
        params=dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_wolf=1, polar_wolf_alpha=0.13,
                    polar_palmo=1, polar_gs_ranked=1, polar_gamma=1.03, polar_max_iter=4),
        expected=dict(polar=-0.00917), decimals=5,
        source="SURVEY.md 8c G4: reference binary run by the survey stage on sample_configs/inputs/003 as NVT",
        kind="survey")

    # G6: PCN-61 single cell carved from the 3x1x1 supercell + first 416 BSSP H2 in that cell = 4096 atoms
    rows = read_pqr(f"{REF}/sample_configs_gpu/3_PCN61/input.pdb")
    half = 21.398
    frame = [r for r in rows if r["frozen"] and -half <= r["pos"][0] < half]
    mols = {}
    order = []
    for r in rows:
        if r["frozen"]:
            continue
        if r["molecule"] not in mols:
            mols[r["molecule"]] = []
            order.append(r["molecule"])
        mols[r["molecule"]].append(r)
    h2 = []
    nmol = 0
    for m in order:
        if -half <= mols[m][0]["pos"][0] < half:
            h2.extend(mols[m])
            nmol += 1
            if nmol == 416:
                break
    carved = frame + h2
    basis61 = [[42.796, 0, 0], [0, 42.796, 0], [0, 0, 42.796]]
    np.savez_compressed(f"{OUT}/pcn61_bssp_4096.npz", **rows_to_arrays(carved, basis61))
    fixtures["pcn61_bssp_4096"] = dict(
        n=len(carved),
        params=dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, pbc_cutoff=8.0,
                    feynman_hibbs=1, feynman_hibbs_order=4),
        expected=dict(energy=-179655.447957, coulombic=-10833.887297, rd=-139212.042820, polar=-29609.517840),
        decimals=6,
        source="SURVEY.md 8c G6 / BASELINE.md section 2: reference binary run by the survey stage on the PCN-61 "
               "single cell carved from sample_configs_gpu/3_PCN61/input.pdb (flags of iter.inp, NVT)",
        kind="survey", frame_atoms=len(frame), h2_molecules=nmol)

    with open(f"{OUT}/fixtures.json", "w") as f:
        json.dump(fixtures, f, indent=1, sort_keys=True)
    for k, v in fixtures.items():
        print(k, v["n"])


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are already committed")
    main()
