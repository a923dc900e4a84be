"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under mpmc_amd/ does.  See oracle/mpmc_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPMC_ORACLE_LIB: the sanitizer build (make -C oracle asan) for tests/run_asan.sh
_LIB = os.environ.get("MPMC_ORACLE_LIB") or os.path.join(_HERE, "libmpmc_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("temperature", C.c_double),
        ("rd_only", C.c_int),
        ("rd_lrc", C.c_int),
        ("feynman_hibbs", C.c_int),
        ("feynman_hibbs_order", C.c_int),
        ("pbc_cutoff", C.c_double),
        ("ewald_alpha_set", C.c_int),
        ("ewald_alpha", C.c_double),
        ("ewald_kmax", C.c_int),
        ("polarization", C.c_int),
        ("polar_damp", C.c_double),
        ("polar_max_iter", C.c_int),
        ("polar_precision", C.c_double),
        ("polar_gamma", C.c_double),
        ("polar_gs", C.c_int),
        ("polar_gs_ranked", C.c_int),
        ("polar_sor", C.c_int),
        ("polar_esor", C.c_int),
        ("polar_palmo", C.c_int),
        ("polar_rrms", C.c_int),
        ("polar_zodid", C.c_int),
        ("polar_wolf", C.c_int),
        ("polar_wolf_alpha", C.c_double),
        ("polar_ewald", C.c_int),
        ("polar_ewald_alpha_set", C.c_int),
        ("polar_ewald_alpha", C.c_double),
        ("wolf", C.c_int),
    ]


class OrcSystem(C.Structure):
    _fields_ = [
        ("n", C.c_int),
        ("pos", C.c_void_p),
        ("charge", C.c_void_p),
        ("alpha", C.c_void_p),
        ("epsilon", C.c_void_p),
        ("sigma", C.c_void_p),
        ("mass", C.c_void_p),
        ("molecule", C.c_void_p),
        ("frozen", C.c_void_p),
        ("basis", C.c_double * 9),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("energy", C.c_double),
        ("rd_energy", C.c_double),
        ("coulombic_energy", C.c_double),
        ("es_real", C.c_double),
        ("es_recip", C.c_double),
        ("es_self", C.c_double),
        ("polarization_energy", C.c_double),
        ("volume", C.c_double),
        ("cutoff", C.c_double),
        ("ewald_alpha", C.c_double),
        ("polar_ewald_alpha", C.c_double),
        ("dipole_rrms", C.c_double),
        ("polar_iterations", C.c_int),
        ("iter_success", C.c_int),
    ]


class OrcVectors(C.Structure):
    _fields_ = [
        ("ef_static", C.c_void_p),
        ("ef_induced", C.c_void_p),
        ("ef_induced_change", C.c_void_p),
        ("mu", C.c_void_p),
        ("rank_metric", C.c_void_p),
        ("ranked_array", C.c_void_p),
        ("A_matrix", C.c_void_p),
    ]


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
        os.path.join(_HERE, "mpmc_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_energy.restype = C.c_int
        _lib.orc_energy.argtypes = [C.POINTER(OrcSystem), C.POINTER(OrcParams), C.POINTER(OrcResult), C.c_void_p]
        _lib.orc_energy_cached.restype = C.c_int
        _lib.orc_energy_cached.argtypes = [C.POINTER(OrcSystem), C.POINTER(OrcParams), C.POINTER(OrcResult), C.c_void_p,
                                           C.c_void_p]
        _lib.orc_cache_create.restype = C.c_void_p
        _lib.orc_cache_create.argtypes = [C.c_int]
        _lib.orc_cache_free.argtypes = [C.c_void_p]
        _lib.orc_default_params.argtypes = [C.POINTER(OrcParams)]
        _lib.orc_kvector_count.restype = C.c_int
        _lib.orc_kvector_count.argtypes = [C.c_int]
    return _lib


PARAM_NAMES = [f[0] for f in OrcParams._fields_]


def make_params(**kw):
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        if k not in PARAM_NAMES:
            raise KeyError(k)
        setattr(p, k, v)
    return p


class Cache:
    """Per-pair state kept between calls, like the reference's pair list (see mpmc_oracle.h)."""

    def __init__(self, n):
        self.ptr = C.c_void_p(lib().orc_cache_create(int(n)))

    def close(self):
        if self.ptr:
            lib().orc_cache_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        self.close()


def energy(system, params, want_vectors=False, want_A=False, cache=None):
    """system: dict with pos[n,3], charge, alpha, epsilon, sigma, mass, molecule, frozen, basis[3,3].
    params: dict of reference config keywords (see OrcParams).  Returns dict."""
    n = int(len(system["charge"]))
    arrs = {
        "pos": np.ascontiguousarray(system["pos"], dtype=np.float64).reshape(n, 3),
        "charge": np.ascontiguousarray(system["charge"], dtype=np.float64),
        "alpha": np.ascontiguousarray(system["alpha"], dtype=np.float64),
        "epsilon": np.ascontiguousarray(system["epsilon"], dtype=np.float64),
        "sigma": np.ascontiguousarray(system["sigma"], dtype=np.float64),
        "mass": np.ascontiguousarray(system["mass"], dtype=np.float64),
        "molecule": np.ascontiguousarray(system["molecule"], dtype=np.int32),
        "frozen": np.ascontiguousarray(system["frozen"], dtype=np.int32),
    }
    s = OrcSystem()
    s.n = n
    for k, a in arrs.items():
        setattr(s, k, a.ctypes.data)
    b = np.ascontiguousarray(system["basis"], dtype=np.float64).reshape(9)
    for i in range(9):
        s.basis[i] = b[i]
    p = params if isinstance(params, OrcParams) else make_params(**params)
    r = OrcResult()
    out = {}
    vec_ptr = None
    if want_vectors or want_A:
        v = OrcVectors()
        out["ef_static"] = np.zeros((n, 3))
        out["ef_induced"] = np.zeros((n, 3))
        out["ef_induced_change"] = np.zeros((n, 3))
        out["mu"] = np.zeros((n, 3))
        out["rank_metric"] = np.zeros(n)
        out["ranked_array"] = np.zeros(n, dtype=np.int32)
        for k in ("ef_static", "ef_induced", "ef_induced_change", "mu", "rank_metric", "ranked_array"):
            setattr(v, k, out[k].ctypes.data)
        if want_A:
            out["A_matrix"] = np.zeros((3 * n, 3 * n))
            v.A_matrix = out["A_matrix"].ctypes.data
        vec_ptr = C.addressof(v)
    if cache is not None:
        rc = lib().orc_energy_cached(C.byref(s), C.byref(p), C.byref(r), vec_ptr, cache.ptr)
    else:
        rc = lib().orc_energy(C.byref(s), C.byref(p), C.byref(r), vec_ptr)
    if rc != 0:
        raise RuntimeError("orc_energy failed: %d" % rc)
    for f, _ in OrcResult._fields_:
        out[f] = getattr(r, f)
    return out
