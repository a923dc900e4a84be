"""ctypes binding of the C ABI in include/mpmc_hip.h (libmpmc_hip.so).

This is plumbing for tests and bench.py; the product is the shared library.  There is no
fallback: if the library or a gfx950 device is missing every call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmpmc_hip.so")

# every symbol include/mpmc_hip.h declares
EXPORTS = [
    "mpmc_hip_last_error", "mpmc_hip_abi_version", "mpmc_hip_device_count", "mpmc_hip_create",
    "mpmc_hip_destroy", "mpmc_hip_set_option", "mpmc_hip_default_params", "mpmc_hip_set_params", "mpmc_hip_set_box",
    "mpmc_hip_upload", "mpmc_hip_update_atoms", "mpmc_hip_insert_molecule", "mpmc_hip_remove_molecule",
    "mpmc_hip_slot_count", "mpmc_hip_set_sweep_order", "mpmc_hip_energy", "mpmc_hip_energy_begin", "mpmc_hip_energy_end",
    "mpmc_hip_download_dipoles",
    "mpmc_hip_download_amatrix", "mpmc_hip_download_ranking", "mpmc_hip_get_timings",
    "mpmc_hip_comm_unique_id", "mpmc_hip_comm_create", "mpmc_hip_comm_size", "mpmc_hip_comm_rank",
    "mpmc_hip_allreduce_observables", "mpmc_hip_allreduce_observables_begin", "mpmc_hip_allreduce_observables_end",
    "mpmc_hip_gather_observables", "mpmc_hip_comm_destroy",
]


class Params(C.Structure):
    _fields_ = [
        ("temperature", C.c_double),
        ("rd_only", C.c_int),
        ("rd_lrc", C.c_int),
        ("feynman_hibbs", C.c_int),
        ("feynman_hibbs_order", C.c_int),
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


class Result(C.Structure):
    _fields_ = [
        ("energy", C.c_double),
        ("rd_energy", C.c_double),
        ("coulombic_energy", C.c_double),
        ("polarization_energy", C.c_double),
        ("es_real", C.c_double),
        ("es_recip", C.c_double),
        ("es_self", C.c_double),
        ("dipole_rrms", C.c_double),
        ("volume", C.c_double),
        ("cutoff", C.c_double),
        ("ewald_alpha", C.c_double),
        ("polar_ewald_alpha", C.c_double),
        ("polar_iterations", C.c_int),
        ("iter_success", C.c_int),
        ("n_atoms", C.c_int),
        ("status", C.c_int),
    ]


class Timings(C.Structure):
    _fields_ = [
        ("pair_ms", C.c_float),
        ("recip_ms", C.c_float),
        ("field_ms", C.c_float),
        ("amatrix_ms", C.c_float),
        ("sweep_ms", C.c_float),
        ("palmo_ms", C.c_float),
        ("other_ms", C.c_float),
        ("total_ms", C.c_float),
        ("sweep_count", C.c_int),
        ("amatrix_count", C.c_int),
        ("graph_steps", C.c_int),
        ("event_pair_ms", C.c_float),
        ("event_pair_count", C.c_int),
        ("spec_rank_redos", C.c_int),
        ("resident_calls", C.c_int),
        ("resident_fallbacks", C.c_int),
    ]


PARAM_NAMES = [f[0] for f in Params._fields_]

_lib = None


def load():
    """Load libmpmc_hip.so and declare prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmpmc_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
            "there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    lib.mpmc_hip_last_error.restype = C.c_char_p
    lib.mpmc_hip_abi_version.restype = C.c_int
    lib.mpmc_hip_device_count.restype = C.c_int
    lib.mpmc_hip_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    lib.mpmc_hip_destroy.argtypes = [vp]
    lib.mpmc_hip_destroy.restype = None
    lib.mpmc_hip_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.mpmc_hip_default_params.argtypes = [C.POINTER(Params)]
    lib.mpmc_hip_default_params.restype = None
    lib.mpmc_hip_set_params.argtypes = [vp, C.POINTER(Params)]
    lib.mpmc_hip_set_box.argtypes = [vp, dp, C.c_double]
    lib.mpmc_hip_upload.argtypes = [vp, C.c_int] + [vp] * 10
    lib.mpmc_hip_update_atoms.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    lib.mpmc_hip_insert_molecule.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.mpmc_hip_remove_molecule.argtypes = [vp, C.c_int, C.c_int]
    lib.mpmc_hip_slot_count.argtypes = [vp]
    lib.mpmc_hip_set_sweep_order.argtypes = [vp, C.c_int, vp]
    lib.mpmc_hip_energy.argtypes = [vp, C.POINTER(Result)]
    lib.mpmc_hip_energy_begin.argtypes = [vp]
    lib.mpmc_hip_energy_end.argtypes = [vp, C.POINTER(Result)]
    lib.mpmc_hip_download_dipoles.argtypes = [vp, vp, vp, vp, vp]
    lib.mpmc_hip_download_amatrix.argtypes = [vp, vp]
    lib.mpmc_hip_download_ranking.argtypes = [vp, vp, vp]
    lib.mpmc_hip_get_timings.argtypes = [vp, C.POINTER(Timings)]
    lib.mpmc_hip_comm_unique_id.argtypes = [vp]
    lib.mpmc_hip_comm_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, vp]
    lib.mpmc_hip_allreduce_observables.argtypes = [vp, vp, C.c_int]
    lib.mpmc_hip_allreduce_observables_begin.argtypes = [vp, vp, C.c_int]
    lib.mpmc_hip_allreduce_observables_end.argtypes = [vp, vp]
    lib.mpmc_hip_gather_observables.argtypes = [vp, vp, C.c_int, vp]
    lib.mpmc_hip_comm_size.argtypes = [vp]
    lib.mpmc_hip_comm_rank.argtypes = [vp]
    lib.mpmc_hip_comm_destroy.argtypes = [vp]
    lib.mpmc_hip_comm_destroy.restype = None
    _lib = lib
    return lib


class EngineError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise EngineError(load().mpmc_hip_last_error().decode())


def make_params(**kw):
    p = Params()
    load().mpmc_hip_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "pbc_cutoff":
            continue  # a box property at this boundary (set_box)
        if k not in PARAM_NAMES:
            raise KeyError(k)
        setattr(p, k, v)
    return p


class Engine:
    """One device context = one MC walker's energy engine."""

    def __init__(self, max_atoms, device=0):
        self.lib = load()
        self.ctx = C.c_void_p()
        _chk(self.lib.mpmc_hip_create(C.byref(self.ctx), device, int(max_atoms)))
        self.n = 0

    def close(self):
        if self.ctx:
            self.lib.mpmc_hip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, **kw):
        p = kw.pop("_struct", None) or make_params(**kw)
        _chk(self.lib.mpmc_hip_set_params(self.ctx, C.byref(p)))

    def set_option(self, name, value):
        _chk(self.lib.mpmc_hip_set_option(self.ctx, name.encode(), int(value)))

    def set_box(self, basis, pbc_cutoff=0.0):
        b = np.ascontiguousarray(basis, dtype=np.float64).reshape(9)
        _chk(self.lib.mpmc_hip_set_box(self.ctx, b.ctypes.data_as(C.POINTER(C.c_double)), float(pbc_cutoff)))

    def upload(self, system):
        """system: dict with pos[n,3], charge, alpha, epsilon, sigma, mass, molecule, frozen."""
        pos = np.ascontiguousarray(system["pos"], dtype=np.float64)
        n = pos.shape[0]
        cols = [np.ascontiguousarray(pos[:, k]) for k in range(3)]
        arrs = cols + [np.ascontiguousarray(system[k], dtype=np.float64)
                       for k in ("charge", "alpha", "epsilon", "sigma", "mass")]
        mol = np.ascontiguousarray(system["molecule"], dtype=np.int32)
        frz = np.ascontiguousarray(system["frozen"], dtype=np.uint8)
        _chk(self.lib.mpmc_hip_upload(self.ctx, n, *[a.ctypes.data for a in arrs], mol.ctypes.data, frz.ctypes.data))
        self.n = n

    def load_system(self, system, params):
        """Convenience: params (incl. optional pbc_cutoff) + box + atoms."""
        self.set_params(**params)
        self.set_box(system["basis"], params.get("pbc_cutoff", 0.0))
        self.upload(system)

    def update_atoms(self, first, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        cols = [np.ascontiguousarray(pos[:, k]) for k in range(3)]
        _chk(self.lib.mpmc_hip_update_atoms(self.ctx, int(first), pos.shape[0], *[a.ctypes.data for a in cols]))

    def insert_molecule(self, pos, charge, alpha, epsilon, sigma, mass, frozen=False):
        """One molecule enters the resident configuration (mpmc_hip_insert_molecule).  Returns its first slot, or
        None when the engine asks for a full upload instead."""
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        cols = [np.ascontiguousarray(pos[:, k]) for k in range(3)]
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (charge, alpha, epsilon, sigma, mass)]
        first = C.c_int(-1)
        rc = self.lib.mpmc_hip_insert_molecule(self.ctx, pos.shape[0], *[a.ctypes.data for a in cols + arrs],
                                               int(bool(frozen)), C.byref(first))
        if rc < 0:
            _chk(rc)
        if rc > 0:
            return None
        self.n = int(self.lib.mpmc_hip_slot_count(self.ctx))
        return first.value

    def remove_molecule(self, first, count):
        """The molecule in slots [first, first + count) leaves (mpmc_hip_remove_molecule); False = upload instead."""
        rc = self.lib.mpmc_hip_remove_molecule(self.ctx, int(first), int(count))
        if rc < 0:
            _chk(rc)
        return rc == 0

    def set_sweep_order(self, slots):
        """Gauss-Seidel modes after insert / remove: device slots of the polarizable atoms in the caller's atom order."""
        a = np.ascontiguousarray(slots, dtype=np.int32)
        _chk(self.lib.mpmc_hip_set_sweep_order(self.ctx, len(a), a.ctypes.data))

    def energy(self):
        r = Result()
        _chk(self.lib.mpmc_hip_energy(self.ctx, C.byref(r)))
        return {f: getattr(r, f) for f, _ in Result._fields_}

    def energy_begin(self):
        _chk(self.lib.mpmc_hip_energy_begin(self.ctx))

    def energy_end(self):
        r = Result()
        _chk(self.lib.mpmc_hip_energy_end(self.ctx, C.byref(r)))
        return {f: getattr(r, f) for f, _ in Result._fields_}

    def dipoles(self):
        out = {k: np.zeros((self.n, 3)) for k in ("mu", "ef_static", "ef_induced", "ef_induced_change")}
        _chk(self.lib.mpmc_hip_download_dipoles(self.ctx, out["mu"].ctypes.data, out["ef_static"].ctypes.data,
                                                out["ef_induced"].ctypes.data,
                                                out["ef_induced_change"].ctypes.data))
        return out

    def amatrix(self):
        A = np.zeros((3 * self.n, 3 * self.n))
        _chk(self.lib.mpmc_hip_download_amatrix(self.ctx, A.ctypes.data))
        return A

    def timings(self):
        t = Timings()
        _chk(self.lib.mpmc_hip_get_timings(self.ctx, C.byref(t)))
        return {f: getattr(t, f) for f, _ in Timings._fields_}

    def ranking(self):
        rank = np.zeros(self.n)
        order = np.zeros(self.n, dtype=np.int32)
        _chk(self.lib.mpmc_hip_download_ranking(self.ctx, rank.ctypes.data, order.ctypes.data))
        return rank, order

    def timings(self):
        t = Timings()
        _chk(self.lib.mpmc_hip_get_timings(self.ctx, C.byref(t)))
        return {f: getattr(t, f) for f, _ in Timings._fields_}
