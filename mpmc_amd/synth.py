"""Synthetic periodic boxes: S-LJ(N), S-ES(N), S-POL(N).

Deterministic (seeded) inputs of the shapes BASELINE.json names, used by bench.py and the
parity tests.  Units as the reference reads them: Angstrom, K, charges already multiplied by
E2REDUCED (reference src/io/read_pqr.c:249).
"""
import numpy as np

E2REDUCED = 408.7816

# BSSP H2, parameters from the reference's sample_configs_gpu/cuda_pol.small/small.initial.pdb:1-5
# (site, offset along the molecular axis, mass, charge/e, alpha, epsilon, sigma)
BSSP_SITES = [
    ("H2G", 0.0, 0.0, -0.7464, 0.69380, 12.76532, 3.15528),
    ("H2E", 0.371, 1.008, 0.3732, 0.00044, 0.0, 0.0),
    ("H2E", -0.371, 1.008, 0.3732, 0.00044, 0.0, 0.0),
    ("H2N", 0.363, 0.0, 0.0, 0.0, 2.16726, 2.37031),
    ("H2N", -0.363, 0.0, 0.0, 0.0, 2.16726, 2.37031),
]


def _lattice(nmol, spacing, jitter, rng):
    m = int(np.ceil(nmol ** (1.0 / 3.0) - 1e-9))
    L = spacing * m
    idx = np.arange(m ** 3)
    ijk = np.stack([idx // (m * m), (idx // m) % m, idx % m], axis=1)[:nmol]
    com = (ijk + 0.5) * spacing - 0.5 * L
    com = com + rng.uniform(-jitter, jitter, size=com.shape)
    return com, L


def _random_axes(n, rng):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _finish(pos, q, alpha, eps, sig, mass, mol, frozen, L):
    return dict(
        pos=np.ascontiguousarray(pos, dtype=np.float64),
        charge=np.asarray(q, dtype=np.float64) * E2REDUCED,
        alpha=np.asarray(alpha, dtype=np.float64),
        epsilon=np.asarray(eps, dtype=np.float64),
        sigma=np.asarray(sig, dtype=np.float64),
        mass=np.asarray(mass, dtype=np.float64),
        molecule=np.asarray(mol, dtype=np.int32),
        frozen=np.asarray(frozen, dtype=np.int32),
        basis=np.diag([L, L, L]).astype(np.float64),
    )


def s_lj(n, seed=None):
    """N single-site LJ atoms (eps 120 K, sigma 3.4 A) on a jittered cubic lattice, spacing 3.8 A."""
    rng = np.random.default_rng(1234 + n if seed is None else seed)
    com, L = _lattice(n, 3.8, 0.3, rng)
    z = np.zeros(n)
    return _finish(com, z, z, np.full(n, 120.0), np.full(n, 3.4), np.full(n, 39.948), np.arange(1, n + 1), z, L)


def s_es(n, seed=None):
    """N/2 rigid dimers (bond 1.0 A, q = +-0.4 e, LJ on site 1), spacing 3.8 A."""
    rng = np.random.default_rng(2234 + n if seed is None else seed)
    nmol = n // 2
    com, L = _lattice(nmol, 3.8, 0.3, rng)
    ax = _random_axes(nmol, rng)
    pos = np.empty((2 * nmol, 3))
    pos[0::2] = com + 0.5 * ax
    pos[1::2] = com - 0.5 * ax
    q = np.tile([0.4, -0.4], nmol)
    eps = np.tile([120.0, 0.0], nmol)
    sig = np.tile([3.4, 0.0], nmol)
    mass = np.tile([20.0, 20.0], nmol)
    mol = np.repeat(np.arange(1, nmol + 1), 2)
    z = np.zeros(2 * nmol)
    return _finish(pos, q, z, eps, sig, mass, mol, z, L)


def s_pol(n, seed=None, spacing=3.6):
    """floor(N/5) five-site BSSP H2 molecules (+ N mod 5 single LJ sites so that the atom count is
    exactly N), COMs on a jittered cubic lattice (spacing 3.6 A), random orientations."""
    rng = np.random.default_rng(3234 + n if seed is None else seed)
    nmol = n // 5
    extra = n - 5 * nmol
    com, L = _lattice(nmol + extra, spacing, 0.3, rng)
    ax = _random_axes(nmol, rng)
    pos, q, alpha, eps, sig, mass, mol = [], [], [], [], [], [], []
    for m in range(nmol):
        for (_, off, ms, qq, al, ep, sg) in BSSP_SITES:
            pos.append(com[m] + off * ax[m])
            q.append(qq)
            alpha.append(al)
            eps.append(ep)
            sig.append(sg)
            mass.append(ms)
            mol.append(m + 1)
    for e in range(extra):
        pos.append(com[nmol + e])
        q.append(0.0)
        alpha.append(0.0)
        eps.append(10.22)
        sig.append(2.556)
        mass.append(4.0026)
        mol.append(nmol + e + 1)
    return _finish(np.array(pos), q, alpha, eps, sig, mass, mol, np.zeros(n), L)


# flag sets (reference config keywords) used with the synthetic polarizable boxes
FLAGS_POL_JACOBI = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=10,
                        feynman_hibbs=1, feynman_hibbs_order=4)
FLAGS_POL_PRODUCTION = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_wolf=1,
                            polar_wolf_alpha=0.13, polar_gs_ranked=1, polar_palmo=1, polar_gamma=1.03,
                            polar_max_iter=4)
FLAGS_LJ = dict(temperature=100.0, rd_only=1)
FLAGS_ES = dict(temperature=100.0)
