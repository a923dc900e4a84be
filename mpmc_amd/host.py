"""ctypes binding of the C host layer (mpmc_amd/host/libmpmc_host.so): system_t, energy(), mc() -- the mirror
of the reference's host interface that sits above the C ABI.  Plumbing for tests and bench.py."""
import ctypes as C
import os

import numpy as np

from . import engine

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPMC_HOST_LIB: the sanitizer build (make -C mpmc_amd/host asan) for tests/run_asan.sh
LIB_PATH = os.environ.get("MPMC_HOST_LIB") or os.path.join(_HERE, "host", "libmpmc_host.so")
EXE_PATH = os.path.join(_HERE, "host", "mpmc_hip")

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("mpmc_amd/host/libmpmc_host.so is not built (run __graft_entry__.build())")
    engine.load()  # libmpmc_hip.so first (also found through the rpath)
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.system_from_arrays.restype = vp
    lib.system_from_arrays.argtypes = [C.c_int] + [vp] * 9
    lib.setup_system.restype = vp
    lib.setup_system.argtypes = [C.c_char_p]
    lib.free_system.argtypes = [vp]
    lib.free_system.restype = None
    lib.host_apply_config.argtypes = [vp, C.c_char_p]
    lib.host_mc_steps.argtypes = [vp, C.c_int]
    lib.host_set_device.argtypes = [vp, C.c_int]
    lib.host_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.host_enable_timing.argtypes = [vp, C.c_int]
    lib.host_get_timings.argtypes = [vp, C.POINTER(engine.Timings)]
    lib.host_get_observables.argtypes = [vp, vp]
    lib.host_get_positions.argtypes = [vp, vp]
    lib.host_get_dipoles.argtypes = [vp, vp, vp, vp]
    lib.host_natoms.argtypes = [vp]
    lib.host_get_system.argtypes = [vp] * 9
    lib.host_seed.argtypes = [vp, C.c_uint]
    lib.host_mc_steps_multi.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    lib.host_mc_steps_multi.restype = C.c_int
    lib.host_get_rand.argtypes = [vp]
    lib.host_get_rand.restype = C.c_double
    lib.energy.argtypes = [vp]
    lib.energy.restype = C.c_double
    lib.mc.argtypes = [vp]
    lib.countNatoms.argtypes = [vp]
    lib.translate.argtypes = [vp, vp, vp, C.c_double]
    lib.checkpoint.argtypes = [vp]
    lib.make_move.argtypes = [vp]
    lib.restore.argtypes = [vp]
    lib.walkers_unique_id.argtypes = [vp]
    lib.walkers_init.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.walkers_pool_begin.argtypes = [vp, vp, C.c_int]
    lib.walkers_pool_end.argtypes = [vp, vp, C.c_int]
    lib.walkers_finalize.argtypes = [vp]
    lib.walkers_finalize.restype = None
    lib.host_device_error.argtypes = [vp]
    _lib = lib
    return lib


def config_text(flags, extra=None):
    """flags in C-ABI / oracle naming -> the reference's keyword lines."""
    onoff = {"rd_only", "rd_lrc", "feynman_hibbs", "polarization", "polar_gs", "polar_gs_ranked", "polar_sor",
             "polar_esor", "polar_palmo", "polar_rrms", "polar_zodid", "polar_wolf", "polar_ewald", "wolf"}
    lines = []
    for k, v in flags.items():
        if k in onoff:
            lines.append("%s %s" % (k, "on" if v else "off"))
        elif k in ("ewald_alpha_set", "polar_ewald_alpha_set"):
            continue
        else:
            lines.append("%s %r" % (k, v))
    if flags.get("polarization"):
        lines.append("polar_damp_type exponential")
        lines.append("polar_iterative on")
    for k, v in (extra or {}).items():
        lines.append("%s %s" % (k, v))
    return "\n".join(lines) + "\n"


def walkers_unique_id():
    """128-byte RCCL id (made on rank 0, handed to the other ranks by the launcher)."""
    buf = (C.c_ubyte * 128)()
    if load().walkers_unique_id(buf) != 0:
        raise engine.EngineError("walkers_unique_id: " + engine.load().mpmc_hip_last_error().decode())
    return bytes(buf)


def mc_steps_multi(walkers, nsteps):
    """Advance several HostSystem walkers together (one process, interleaved on the device): every walker takes
    `nsteps` steps exactly as it would alone.  Returns the number of accepted moves over all walkers."""
    lib = walkers[0].lib
    arr = (C.c_void_p * len(walkers))(*[w.ptr for w in walkers])
    acc = lib.host_mc_steps_multi(arr, len(walkers), int(nsteps))
    if acc < 0:
        raise engine.EngineError(engine.load().mpmc_hip_last_error().decode())
    return acc


class HostSystem:
    """A system_t built from arrays, driven through the C host layer."""

    def __init__(self, system, flags, device=0, seed=None, move_factor=0.01, rot_factor=0.01, extra=None):
        self.lib = load()
        n = len(system["charge"])
        self.n = n
        arrs = [np.ascontiguousarray(system["pos"], dtype=np.float64).reshape(n, 3)]
        arrs += [np.ascontiguousarray(system[k], dtype=np.float64)
                 for k in ("charge", "alpha", "epsilon", "sigma", "mass")]
        arrs += [np.ascontiguousarray(system[k], dtype=np.int32) for k in ("molecule", "frozen")]
        arrs += [np.ascontiguousarray(system["basis"], dtype=np.float64).reshape(9)]
        self.ptr = C.c_void_p(self.lib.system_from_arrays(n, *[a.ctypes.data for a in arrs]))
        extra = dict({"move_factor": move_factor, "rot_factor": rot_factor}, **(extra or {}))
        if self.lib.host_apply_config(self.ptr, config_text(flags, extra).encode()) != 0:
            raise ValueError("host layer rejected the configuration")
        self.lib.host_set_device(self.ptr, device)
        if seed is not None:
            self.lib.host_seed(self.ptr, int(seed))

    def close(self):
        if self.ptr:
            self.lib.free_system(self.ptr)
            self.ptr = C.c_void_p()

    def energy(self):
        return self.lib.energy(self.ptr)

    def set_option(self, name, value):
        """Engine A/B knob (mpmc_hip_set_option); the device context exists after the first energy()."""
        if self.lib.host_set_option(self.ptr, name.encode(), int(value)) != 0:
            raise engine.EngineError(engine.load().mpmc_hip_last_error().decode())

    def observables(self):
        out = np.zeros(8)
        self.lib.host_get_observables(self.ptr, out.ctypes.data)
        return dict(energy=out[0], coulombic_energy=out[1], rd_energy=out[2], polarization_energy=out[3], N=out[4],
                    polar_iterations=out[5], accept=int(out[6]), reject=int(out[7]))

    def mc_steps(self, nsteps):
        r = self.lib.host_mc_steps(self.ptr, int(nsteps))
        if r < 0:
            raise engine.EngineError(engine.load().mpmc_hip_last_error().decode())
        return r

    # ---- walker pooling through the C ABI's RCCL entry (the reference's MPI_Gather, mc.c:417-432)
    def walkers_init(self, nranks, rank, unique_id):
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id)) if unique_id is not None else None
        if self.lib.walkers_init(self.ptr, int(nranks), int(rank), buf) != 0:
            raise engine.EngineError("walkers_init: " + engine.load().mpmc_hip_last_error().decode())

    def pool_begin(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        if self.lib.walkers_pool_begin(self.ptr, v.ctypes.data, len(v)) != 0:
            raise engine.EngineError("walkers_pool_begin: " + engine.load().mpmc_hip_last_error().decode())

    def pool_end(self, count):
        out = np.zeros(count)
        if self.lib.walkers_pool_end(self.ptr, out.ctypes.data, count) != 0:
            raise engine.EngineError("walkers_pool_end: " + engine.load().mpmc_hip_last_error().decode())
        return out

    def positions(self):
        pos = np.zeros((self.natoms(), 3))
        self.lib.host_get_positions(self.ptr, pos.ctypes.data)
        return pos

    def natoms(self):
        return int(self.lib.host_natoms(self.ptr))

    def system(self, basis):
        """Current configuration as a dict of arrays (N may differ from the initial one under uvt)."""
        n = self.natoms()
        f = {k: np.zeros(n) for k in ("charge", "alpha", "epsilon", "sigma", "mass")}
        pos = np.zeros((n, 3))
        mol = np.zeros(n, dtype=np.int32)
        frz = np.zeros(n, dtype=np.int32)
        self.lib.host_get_system(self.ptr, pos.ctypes.data, f["charge"].ctypes.data, f["alpha"].ctypes.data,
                                 f["epsilon"].ctypes.data, f["sigma"].ctypes.data, f["mass"].ctypes.data,
                                 mol.ctypes.data, frz.ctypes.data)
        return dict(pos=pos, molecule=mol, frozen=frz, basis=np.asarray(basis, dtype=np.float64), **f)

    def dipoles(self):
        n = self.natoms()  # differs from the initial count under uvt
        mu, es, ei = (np.zeros((n, 3)) for _ in range(3))
        if self.lib.host_get_dipoles(self.ptr, mu.ctypes.data, es.ctypes.data, ei.ctypes.data):
            raise engine.EngineError(engine.load().mpmc_hip_last_error().decode())
        return dict(mu=mu, ef_static=es, ef_induced=ei)

    def enable_timing(self, on=True):
        self.lib.host_enable_timing(self.ptr, int(on))

    def timings(self):
        t = engine.Timings()
        self.lib.host_get_timings(self.ptr, C.byref(t))
        return {f: getattr(t, f) for f, _ in engine.Timings._fields_}
