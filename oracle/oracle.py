"""ctypes binding of the CPU oracle (oracle/opal_oracle.c).

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  The product path (opal_amd/) never
does; it fails loudly when its HIP library is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libopal_oracle.so")

BDY_INTERNAL, BDY_LASER, BDY_ABSORBING, BDY_CONDUCTING = 0, 1, 2, 3
ELECTRON, ION, PHOTON = 0, 1, 2

# src/constants.rs:4-27
SPEED_OF_LIGHT = 2.997925e8
SPEED_OF_LIGHT_SQD = 89875517873681764.0
VACUUM_PERMITTIVITY = 8.854188e-12
VACUUM_PERMEABILITY = 1.256637e-6
ELECTRON_CHARGE = -1.602177e-19
ELEMENTARY_CHARGE = 1.602177e-19
ELECTRON_MASS = 9.109383e-31
PROTON_MASS = 1.672622e-27

FN1 = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)
FN2 = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_void_p)
FN3 = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p)


class Cell(C.Structure):
    _fields_ = [("x", C.c_double), ("rho", C.c_double), ("j", C.c_double * 3),
                ("E", C.c_double * 3), ("B", C.c_double * 3)]


class Electron(C.Structure):
    _fields_ = [("cell", C.c_long), ("prev_x", C.c_double), ("x", C.c_double), ("y", C.c_double),
                ("z", C.c_double), ("weight", C.c_double), ("gamma", C.c_double), ("u", C.c_double * 3),
                ("chi", C.c_double), ("tau", C.c_double), ("work", C.c_double), ("flag", C.c_int),
                ("id", C.c_ulong)]


class Ion(C.Structure):
    _fields_ = [("cell", C.c_long), ("prev_x", C.c_double), ("x", C.c_double), ("weight", C.c_double),
                ("Z_star", C.c_double), ("Z", C.c_double), ("A", C.c_double), ("gamma_m1", C.c_double),
                ("u", C.c_double * 3), ("work", C.c_double), ("flag", C.c_int)]


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


def build(force=False):
    """Compile the oracle with gcc (make).  Building the checker is not using it."""
    src = os.path.join(_HERE, "opal_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libopal_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        sig = {
            "oracle_weight": (C.c_double, [C.c_double]),
            "oracle_flux": (C.c_double, [C.c_double, C.c_double]),
            "oracle_rate": (C.c_double, [C.c_double, C.c_double]),
            "oracle_design_unbalanced": (C.c_int, [C.c_int, C.c_size_t, C.c_double, C.c_size_t, vp, vp]),
            "oracle_design_balanced": (C.c_int, [C.c_int, C.c_size_t, C.c_double, C.c_double, C.c_size_t, FN1, vp, vp, vp]),
            "oracle_grid_build": (vp, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp, vp]),
            "oracle_grid_free": (None, [vp]),
            "oracle_grid_size": (C.c_size_t, [vp]),
            "oracle_grid_len": (C.c_size_t, [vp]),
            "oracle_grid_left_bdy_size": (C.c_long, [vp]),
            "oracle_grid_right_bdy_size": (C.c_long, [vp]),
            "oracle_grid_dx": (C.c_double, [vp]),
            "oracle_grid_xmin": (C.c_double, [vp]),
            "oracle_grid_rank": (C.c_int, [vp]),
            "oracle_grid_to_left": (C.c_int, [vp]),
            "oracle_grid_to_right": (C.c_int, [vp]),
            "oracle_grid_get_raw": (None, [vp, C.c_int, vp]),
            "oracle_grid_set_raw": (None, [vp, C.c_int, vp]),
            "oracle_grid_clear": (None, [vp]),
            "oracle_grid_fields_at": (C.c_int, [vp, C.c_long, C.c_double, dp, dp]),
            "oracle_grid_send_left": (None, [vp, vp]),
            "oracle_grid_send_right": (None, [vp, vp]),
            "oracle_grid_synchronize_apply": (None, [vp, vp, vp, FN2, FN2, vp, C.c_double, C.c_double]),
            "oracle_world_synchronize": (None, [vp, C.c_int, FN2, FN2, vp, C.c_double, C.c_double]),
            "oracle_grid_advance": (None, [vp, C.c_double]),
            "oracle_grid_em_field_energy": (C.c_double, [vp]),
            "oracle_grid_interpolate": (None, [vp, vp]),
            "oracle_electron_create": (None, [C.c_long, C.c_double, dp, C.c_double, C.c_double, C.c_double, vp]),
            "oracle_electron_push": (C.c_int, [vp, dp, dp, C.c_double, C.c_double]),
            "oracle_ion_create": (None, [C.c_long, C.c_double, dp, C.c_double, C.c_double, C.c_double, vp]),
            "oracle_ion_push": (C.c_int, [vp, dp, dp, C.c_double, C.c_double]),
            "oracle_rng_seed": (None, [vp, C.c_uint64]),
            "oracle_rng_next": (C.c_uint64, [vp]),
            "oracle_rng_f64": (C.c_double, [vp]),
            "oracle_rng_normal": (C.c_double, [vp]),
            "oracle_rng_exp1": (C.c_double, [vp]),
            "oracle_pop_new_empty": (vp, [C.c_int]),
            "oracle_pop_free": (None, [vp]),
            "oracle_pop_len": (C.c_size_t, [vp]),
            "oracle_pop_new": (vp, [C.c_int, C.c_size_t, FN1, FN3, FN3, FN3, vp, vp, vp, C.c_double]),
            "oracle_pop_set_charge_state": (None, [vp, C.c_double, C.c_double, C.c_double]),
            "oracle_pop_get_soa": (None, [vp] + [vp] * 8),
            "oracle_pop_get_electron_extras": (None, [vp] + [vp] * 5),
            "oracle_pop_from_soa": (vp, [C.c_int, C.c_size_t, vp, vp, vp, vp, vp, vp, C.c_double, C.c_double]),
            "oracle_pop_set_prev_aux": (None, [vp, vp, vp]),
            "oracle_pop_total_kinetic_energy": (C.c_double, [vp]),
            "oracle_pop_push_and_sort": (C.c_size_t, [vp, vp, C.c_double, C.c_int]),
            "oracle_pop_gone_left": (C.c_size_t, [vp]),
            "oracle_pop_gone_right": (C.c_size_t, [vp]),
            "oracle_pop_send_left": (vp, [vp]),
            "oracle_pop_send_right": (vp, [vp]),
            "oracle_pop_particle_size": (C.c_size_t, [vp]),
            "oracle_pop_splice": (None, [vp, vp, vp, C.c_size_t, vp, C.c_size_t]),
            "oracle_world_pop_advance": (C.c_size_t, [vp, vp, C.c_int, C.c_double, C.c_int]),
            "oracle_grid_deposit": (C.c_size_t, [vp, vp, C.c_double, C.c_int]),
            "oracle_world_step": (C.c_size_t, [vp, vp, vp, C.c_int, C.c_int, FN2, FN2, vp, C.c_double, C.c_double, C.c_int]),
            "oracle_world_run_periodic": (C.c_size_t, [vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_size_t, C.c_int]),
            "oracle_max_threads": (C.c_int, []),
            "oracle_set_sort_mode": (None, [C.c_int]),
            "oracle_phase_times": (None, [vp, C.c_int]),
            "oracle_set_parallel_subdomains": (None, [C.c_int]),
            "oracle_pwmci_evaluate": (C.c_int, [C.c_double, vp, C.c_size_t, dp]),
            "oracle_pwmci_invert": (C.c_int, [C.c_double, vp, C.c_size_t, dp, C.POINTER(C.c_int)]),
            "oracle_qed_sample": (C.c_int, [C.c_double] * 5 + [vp]),
            "oracle_qed_classical_sample": (None, [C.c_double] * 5 + [vp]),
            "oracle_philox": (None, [vp, vp, vp]),
            "oracle_emission_randoms": (None, [C.c_uint64, C.c_uint64, C.c_uint32, vp]),
            "oracle_vec3_orthogonal": (None, [vp, vp]),
            "oracle_vec3_rotate_around": (None, [vp, vp, C.c_double, vp]),
            "oracle_emit_radiation": (C.c_long, [vp, vp, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double]),
            "oracle_pop_set_electron_state": (None, [vp, vp, vp]),
            "oracle_pop_get_electron_ids": (None, [vp, vp]),
            "oracle_pop_get_photon_extras": (None, [vp] + [vp] * 5),
            "oracle_pop_get_photon_depths": (None, [vp, vp, vp]),
            "oracle_pop_set_photon_state": (None, [vp] + [vp] * 7),
            "oracle_pop_set_electron_chi": (None, [vp, vp]),
            "oracle_pop_permute": (None, [vp, vp]),
            "oracle_airy_ai": (C.c_int, [C.c_double, dp]),
            "oracle_absorption_cross_section": (C.c_int, [vp, vp, C.c_double, C.c_double, dp]),
            "oracle_stimulated_cross_section": (C.c_int, [vp, vp, C.c_double, C.c_double, dp]),
            "oracle_birth_randoms": (None, [C.c_uint64, C.c_uint64, C.c_uint32, vp]),
            "oracle_absorption_randoms": (None, [C.c_uint64, C.c_uint64, C.c_double, C.c_uint32, vp]),
            "oracle_absorb": (C.c_size_t, [vp, vp, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, vp]),
            "oracle_histogram": (C.c_size_t, [vp, vp, C.c_int, vp, C.c_int, C.c_size_t, C.c_double, C.c_int, vp, C.c_size_t, vp, vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def weight(x):
    return lib().oracle_weight(x)


def flux(xi, xf):
    return lib().oracle_flux(xi, xf)


def rate(chi, gamma):
    return lib().oracle_rate(chi, gamma)


_ZERO2 = FN2(lambda t, x, u: 0.0)

RAW = {"x": 0, "rho": 1, "jx": 2, "jy": 3, "jz": 4, "Ex": 5, "Ey": 6, "Ez": 7, "Bx": 8, "By": 9, "Bz": 10}


def design_unbalanced(numtasks, size, dx, min_subsize=8):
    nx = np.zeros(numtasks, dtype=np.uint64)
    off = np.zeros(numtasks, dtype=np.float64)
    lib().oracle_design_unbalanced(numtasks, size, dx, min_subsize, _ptr(nx), _ptr(off))
    return nx, off


def design_balanced(numtasks, size, xmin, dx, ne, min_subsize=8):
    nx = np.zeros(numtasks, dtype=np.uint64)
    off = np.zeros(numtasks, dtype=np.float64)
    cb = FN1(lambda x, u: float(ne(x)))
    rc = lib().oracle_design_balanced(numtasks, size, xmin, dx, min_subsize, cb, None, _ptr(nx), _ptr(off))
    if rc:
        raise RuntimeError("balanced design failed (reference would panic, grid/mod.rs:179)")
    return nx, off


class Grid:
    """YeeGrid (yee.rs:255-269) for subdomain `rank` of `ngrids`."""

    def __init__(self, rank, ngrids, left, right, xmin, dx, nx, offset):
        nx = np.ascontiguousarray(nx, dtype=np.uint64)
        offset = np.ascontiguousarray(offset, dtype=np.float64)
        self.h = lib().oracle_grid_build(rank, ngrids, left, right, xmin, dx, _ptr(nx), _ptr(offset))
        self.size = lib().oracle_grid_size(self.h)
        self.len = lib().oracle_grid_len(self.h)
        self.left_bdy_size = lib().oracle_grid_left_bdy_size(self.h)
        self.right_bdy_size = lib().oracle_grid_right_bdy_size(self.h)
        self.dx = dx
        self.rank = rank
        self.ngrids = ngrids

    @classmethod
    def single(cls, nx, xmin, dx, left=BDY_INTERNAL, right=BDY_INTERNAL):
        n, off = design_unbalanced(1, nx, dx)
        return cls(0, 1, left, right, xmin, dx, n, off)

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_grid_free(self.h)
            self.h = None

    def xmin(self):
        return lib().oracle_grid_xmin(self.h)

    def to_left(self):
        return lib().oracle_grid_to_left(self.h)

    def to_right(self):
        return lib().oracle_grid_to_right(self.h)

    def raw(self, name):
        out = np.empty(self.len, dtype=np.float64)
        lib().oracle_grid_get_raw(self.h, RAW[name], _ptr(out))
        return out

    def set_raw(self, name, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self.len,)
        lib().oracle_grid_set_raw(self.h, RAW[name], _ptr(a))

    def clear(self):
        lib().oracle_grid_clear(self.h)

    def fields_at(self, c, xi):
        E = (C.c_double * 3)()
        B = (C.c_double * 3)()
        if lib().oracle_grid_fields_at(self.h, c, xi, E, B):
            raise IndexError("fields_at bounds (yee.rs:501-502)")
        return np.array(E), np.array(B)

    def send_left(self):
        buf = (Cell * 8)()
        lib().oracle_grid_send_left(self.h, buf)
        return buf

    def send_right(self):
        buf = (Cell * 8)()
        lib().oracle_grid_send_right(self.h, buf)
        return buf

    def synchronize_apply(self, recv_left, recv_right, laser_y=None, laser_z=None, t=0.0, dt=0.0):
        ly = FN2(lambda t_, x_, u: float(laser_y(t_, x_))) if laser_y else _ZERO2
        lz = FN2(lambda t_, x_, u: float(laser_z(t_, x_))) if laser_z else _ZERO2
        lib().oracle_grid_synchronize_apply(self.h, recv_left, recv_right, ly, lz, None, t, dt)

    def synchronize(self, laser_y=None, laser_z=None, t=0.0, dt=0.0):
        """Single-subdomain synchronize (self-copy when periodic)."""
        world_synchronize([self], laser_y, laser_z, t, dt)

    def advance(self, dt):
        lib().oracle_grid_advance(self.h, dt)

    def em_field_energy(self):
        return lib().oracle_grid_em_field_energy(self.h)

    def interpolate(self):
        buf = (Cell * self.size)()
        lib().oracle_grid_interpolate(self.h, buf)
        return np.frombuffer(buf, dtype=np.float64).reshape(self.size, 11).copy()

    def deposit(self, pop, dt, nthreads=1):
        return lib().oracle_grid_deposit(self.h, pop.h, dt, nthreads)


def _harr(objs):
    arr = (C.c_void_p * len(objs))(*[o.h for o in objs])
    return arr


def world_synchronize(grids, laser_y=None, laser_z=None, t=0.0, dt=0.0):
    ly = FN2(lambda t_, x_, u: float(laser_y(t_, x_))) if laser_y else _ZERO2
    lz = FN2(lambda t_, x_, u: float(laser_z(t_, x_))) if laser_z else _ZERO2
    lib().oracle_world_synchronize(_harr(grids), len(grids), ly, lz, None, t, dt)


def world_pop_advance(pops, grids, dt, nthreads=1):
    return lib().oracle_world_pop_advance(_harr(pops), _harr(grids), len(grids), dt, nthreads)


def world_step(grids, pops_e, pops_i, current_deposition, t, dt, laser_y=None, laser_z=None, nthreads=1):
    ly = FN2(lambda t_, x_, u: float(laser_y(t_, x_))) if laser_y else _ZERO2
    lz = FN2(lambda t_, x_, u: float(laser_z(t_, x_))) if laser_z else _ZERO2
    return lib().oracle_world_step(_harr(grids), _harr(pops_e) if pops_e else None,
                                   _harr(pops_i) if pops_i else None, len(grids),
                                   1 if current_deposition else 0, ly, lz, None, t, dt, nthreads)


def world_run_periodic(grids, pops_e, pops_i, current_deposition, t0, dt, n_steps, nthreads=1):
    return lib().oracle_world_run_periodic(_harr(grids), _harr(pops_e) if pops_e else None,
                                           _harr(pops_i) if pops_i else None, len(grids),
                                           1 if current_deposition else 0, t0, dt, n_steps, nthreads)


def max_threads():
    return lib().oracle_max_threads()


def set_parallel_subdomains(on):
    """S subdomains x T threads (the reference's MPI x Rayon layout) in one process: subdomain loops of the
    world_* functions run concurrently, each with its own nested `nthreads` team.  bench.py's CPU baseline."""
    lib().oracle_set_parallel_subdomains(1 if on else 0)


def phase_times(reset=True):
    """Seconds since the last reset spent in push, sort, migrate, deposit (scatter), deposit (serial reduce),
    synchronize + field solve; the four particle phases are summed over the subdomains of the world calls."""
    out = (C.c_double * 6)()
    lib().oracle_phase_times(out, 1 if reset else 0)
    return dict(zip(("push", "sort", "migrate", "deposit_scatter", "deposit_reduce", "fields"), [float(v) for v in out]))


def set_sort_mode(mode):
    """0: stable counting sort (default; parity tests).  1: parallel comparison sort of the AoS structs, the
    class of work the reference's par_sort_unstable_by_key does every step (bench.py's CPU baseline)."""
    lib().oracle_set_sort_mode(int(mode))


class RNG:
    def __init__(self, seed):
        self.s = Rng()
        lib().oracle_rng_seed(C.byref(self.s), seed)

    def next_u64(self):
        return lib().oracle_rng_next(C.byref(self.s))

    def f64(self):
        return lib().oracle_rng_f64(C.byref(self.s))

    def normal(self):
        return lib().oracle_rng_normal(C.byref(self.s))

    def exp1(self):
        return lib().oracle_rng_exp1(C.byref(self.s))


class Population:
    """Population<T> (particle/mod.rs:141-145), T = Electron or Ion."""

    def __init__(self, h, species):
        self.h = h
        self.species = species

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_pop_free(self.h)
            self.h = None

    @classmethod
    def empty(cls, species):
        return cls(lib().oracle_pop_new_empty(species), species)

    @classmethod
    def new(cls, species, npc, number_density, ux, uy, uz, grid, rng, dt):
        """Population::new (mod.rs:172-203) with Python callables for the expressions."""
        d = FN1(lambda x, u: float(number_density(x)))
        fx = FN3(lambda x, ur, nr, u: float(ux(x, ur, nr)))
        fy = FN3(lambda x, ur, nr, u: float(uy(x, ur, nr)))
        fz = FN3(lambda x, ur, nr, u: float(uz(x, ur, nr)))
        h = lib().oracle_pop_new(species, npc, d, fx, fy, fz, None, grid.h, C.byref(rng.s), dt)
        return cls(h, species)

    @classmethod
    def from_soa(cls, species, cell, x, ux, uy, uz, w, dx, dt, prev_x=None, aux=None):
        cell = np.ascontiguousarray(cell, dtype=np.int64)
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, ux, uy, uz, w)]
        n = len(cell)
        h = lib().oracle_pop_from_soa(species, n, _ptr(cell), *[_ptr(a) for a in arrs], dx, dt)
        p = cls(h, species)
        if prev_x is not None or aux is not None:
            px = np.ascontiguousarray(prev_x, dtype=np.float64) if prev_x is not None else None
            ax = np.ascontiguousarray(aux, dtype=np.float64) if aux is not None else None
            lib().oracle_pop_set_prev_aux(h, _ptr(px) if px is not None else None, _ptr(ax) if ax is not None else None)
        return p

    def __len__(self):
        return lib().oracle_pop_len(self.h)

    def set_charge_state(self, zstar, z, a):
        lib().oracle_pop_set_charge_state(self.h, zstar, z, a)

    def soa(self):
        n = len(self)
        cell = np.empty(n, dtype=np.int64)
        out = {k: np.empty(n, dtype=np.float64) for k in ("x", "prev_x", "ux", "uy", "uz", "w", "aux")}
        lib().oracle_pop_get_soa(self.h, _ptr(cell), *[_ptr(out[k]) for k in ("x", "prev_x", "ux", "uy", "uz", "w", "aux")])
        out["cell"] = cell
        return out

    def electron_extras(self):
        n = len(self)
        out = {k: np.empty(n, dtype=np.float64) for k in ("chi", "tau", "work", "y", "z")}
        lib().oracle_pop_get_electron_extras(self.h, *[_ptr(out[k]) for k in ("chi", "tau", "work", "y", "z")])
        return out

    def set_electron_state(self, tau=None, ids=None):
        """Overwrite the optical depths and/or the ids that key the emission random numbers."""
        t = np.ascontiguousarray(tau, dtype=np.float64) if tau is not None else None
        i = np.ascontiguousarray(ids, dtype=np.uint64) if ids is not None else None
        lib().oracle_pop_set_electron_state(self.h, _ptr(t) if t is not None else None, _ptr(i) if i is not None else None)

    def electron_ids(self):
        out = np.empty(len(self), dtype=np.uint64)
        lib().oracle_pop_get_electron_ids(self.h, _ptr(out))
        return out

    def photon_extras(self):
        n = len(self)
        out = {k: np.empty(n, dtype=np.float64) for k in ("chi", "y", "z", "birth_time")}
        ids = np.empty(n, dtype=np.uint64)
        lib().oracle_pop_get_photon_extras(self.h, *[_ptr(out[k]) for k in ("chi", "y", "z", "birth_time")], _ptr(ids))
        out["id"] = ids
        return out

    def photon_depths(self):
        n = len(self)
        a, b = np.empty(n), np.empty(n)
        lib().oracle_pop_get_photon_depths(self.h, _ptr(a), _ptr(b))
        return a, b

    def set_photon_state(self, chi=None, tau_abs=None, tau_st=None, birth_time=None, ids=None, y=None, z=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        arrs = [f(chi), f(tau_abs), f(tau_st), f(birth_time), None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64), f(y), f(z)]
        self._keep = arrs
        lib().oracle_pop_set_photon_state(self.h, *[None if a is None else _ptr(a) for a in arrs])

    def set_electron_chi(self, chi):
        chi = np.ascontiguousarray(chi, dtype=np.float64)
        lib().oracle_pop_set_electron_chi(self.h, _ptr(chi))

    def permute(self, order):
        """New slot i takes the particle that was in slot order[i]."""
        order = np.ascontiguousarray(order, dtype=np.uint64)
        assert len(order) == len(self)
        lib().oracle_pop_permute(self.h, _ptr(order))

    def total_kinetic_energy(self):
        return lib().oracle_pop_total_kinetic_energy(self.h)

    def advance(self, grid, dt, nthreads=1):
        """Single-subdomain Population::advance (mod.rs:247-376)."""
        return world_pop_advance([self], [grid], dt, nthreads)

    def push_and_sort(self, grid, dt, nthreads=1):
        return lib().oracle_pop_push_and_sort(self.h, grid.h, dt, nthreads)

    def leavers(self):
        """Raw AoS bytes of (send_left, send_right) after push_and_sort (mod.rs:296-297)."""
        sz = lib().oracle_pop_particle_size(self.h)
        nl, nr = lib().oracle_pop_gone_left(self.h), lib().oracle_pop_gone_right(self.h)
        left = C.string_at(lib().oracle_pop_send_left(self.h), nl * sz) if nl else b""
        right = C.string_at(lib().oracle_pop_send_right(self.h), nr * sz) if nr else b""
        return left, right

    def splice(self, grid, recv_left=b"", recv_right=b""):
        sz = lib().oracle_pop_particle_size(self.h)
        bl = C.create_string_buffer(recv_left, len(recv_left)) if recv_left else None
        br = C.create_string_buffer(recv_right, len(recv_right)) if recv_right else None
        lib().oracle_pop_splice(self.h, grid.h, bl, len(recv_left) // sz, br, len(recv_right) // sz)


def pwmci_invert(f, table):
    """pwmci::invert (pwmci.rs:113-228): (x, evaluations) or None."""
    t = np.ascontiguousarray(table, dtype=np.float64)
    x, n = C.c_double(), C.c_int()
    ok = lib().oracle_pwmci_invert(f, _ptr(t), len(t), C.byref(x), C.byref(n))
    return (x.value, n.value) if ok else None


def pwmci_evaluate(x, table):
    t = np.ascontiguousarray(table, dtype=np.float64)
    f = C.c_double()
    return f.value if lib().oracle_pwmci_evaluate(x, _ptr(t), len(t), C.byref(f)) else None


def qed_sample(chi, gamma, r1, r2, r3):
    """photon_emission::sample (photon_emission.rs:129-203): (omega/mc^2, theta, phi)."""
    out = np.empty(3)
    if lib().oracle_qed_sample(chi, gamma, r1, r2, r3, _ptr(out)):
        raise NotImplementedError("chi >= 100 (photon_emission.rs:146 unimplemented!())")
    return tuple(out)


def qed_classical_sample(chi, gamma, r1, r2, r3):
    out = np.empty(3)
    lib().oracle_qed_classical_sample(chi, gamma, r1, r2, r3, _ptr(out))
    return tuple(out)


def philox(ctr, key):
    c, k, o = np.asarray(ctr, dtype=np.uint32), np.asarray(key, dtype=np.uint32), np.empty(4, dtype=np.uint32)
    lib().oracle_philox(_ptr(c), _ptr(k), _ptr(o))
    return o


def emission_randoms(seed, pid, step):
    out = np.empty(4)
    lib().oracle_emission_randoms(seed, pid, step, _ptr(out))
    return out


def emit_radiation(electrons, photons, seed, step, t, min_energy=None, max_angle=None, max_formation_length=None):
    """emit_radiation (interactions.rs:45-107) for one subdomain; returns the number of photons appended."""
    f = lambda v: -1.0 if v is None else float(v)
    n = lib().oracle_emit_radiation(electrons.h, photons.h, seed, step, t, f(min_energy), f(max_angle), f(max_formation_length))
    if n < 0:
        raise NotImplementedError("chi >= 100 (photon_emission.rs:146 unimplemented!())")
    return n


def airy_ai(x):
    """airy_ai_for_positive (qed/special_functions/airy.rs:19-69); None where the reference returns None."""
    out = C.c_double()
    return out.value if lib().oracle_airy_ai(float(x), C.byref(out)) else None


def cross_sections(k, p, chi_gamma, chi_e):
    """(absorption, stimulated emission) scaled cross sections (photon_absorption.rs:16-34,
    stimulated_emission.rs:16-37); None where the reference returns None."""
    k = np.ascontiguousarray(k, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    a, b = C.c_double(), C.c_double()
    ra = lib().oracle_absorption_cross_section(_ptr(k), _ptr(p), chi_gamma, chi_e, C.byref(a))
    rb = lib().oracle_stimulated_cross_section(_ptr(k), _ptr(p), chi_gamma, chi_e, C.byref(b))
    return (a.value if ra else None), (b.value if rb else None)


def birth_randoms(seed, pid, step):
    out = np.empty(2)
    lib().oracle_birth_randoms(seed, pid, step, _ptr(out))
    return out


def absorption_randoms(seed, pid, birth_time, step):
    out = np.empty(3)
    lib().oracle_absorption_randoms(seed, pid, birth_time, step, _ptr(out))
    return out


def absorb(electrons, photons, seed, step, t, dt, dx, max_displacement=None, stop_time=None):
    """absorb (interactions.rs:145-340) for one subdomain; returns (absorbed, stimulated)."""
    f = lambda v: -1.0 if v is None else float(v)
    counts = np.zeros(2, dtype=np.uint64)
    bad = lib().oracle_absorb(electrons.h, photons.h, seed, step, t, dt, dx, f(max_displacement), f(stop_time), _ptr(counts))
    if bad:
        raise AssertionError("the reference's assert!(partial_prob >= 0.0) fires (photon.rs:325,336)")
    return int(counts[0]), int(counts[1])


ACCESSORS = {"x": 0, "r": 1, "energy": 2, "px": 3, "py": 4, "pz": 5, "p_perp": 6, "theta": 7, "phi": 8,
             "longitude": 9, "latitude": 10, "work": 11, "chi": 12}


def histogram(pop, grid, names, bspec="auto", weight="weight"):
    """Histogram::generate_1d / generate_2d (hgram.rs:168-392) with HeightSpec::Density for one subdomain.
    names: one or two accessor names; bspec: "auto", "log", an int (bin count) or a float (bin size)."""
    acc = np.array([ACCESSORS[n] for n in names], dtype=np.int32)
    bkind, fn, fdx = 0, 0, 0.0
    if bspec == "log":
        bkind = 1
    elif isinstance(bspec, int):
        bkind, fn = 2, bspec
    elif isinstance(bspec, float):
        bkind, fdx = 3, bspec
    cap = 1 << 22
    cts = np.zeros(cap)
    meta = np.zeros(8)
    nb = np.zeros(2, dtype=np.uint64)
    n = lib().oracle_histogram(pop.h, grid.h, len(names), _ptr(acc), bkind, fn, fdx, 1 if weight == "energy" else 0,
                               _ptr(cts), cap, _ptr(meta), _ptr(nb))
    return {"cts": cts[:n].copy(), "min": meta[0:2].copy(), "max": meta[2:4].copy(), "bin_sz": meta[4:6].copy(),
            "bin_vol": meta[6], "total": meta[7], "nbins": [int(nb[0]), int(nb[1])]}


def electron_create(cell, x, u, weight, dx, dt):
    e = Electron()
    uu = (C.c_double * 3)(*u)
    lib().oracle_electron_create(cell, x, uu, weight, dx, dt, C.byref(e))
    return e


def electron_push(e, E, B, dx, dt):
    return lib().oracle_electron_push(C.byref(e), (C.c_double * 3)(*E), (C.c_double * 3)(*B), dx, dt)


def ion_create(cell, x, u, weight, dx, dt):
    e = Ion()
    uu = (C.c_double * 3)(*u)
    lib().oracle_ion_create(cell, x, uu, weight, dx, dt, C.byref(e))
    return e


def ion_push(e, E, B, dx, dt):
    return lib().oracle_ion_push(C.byref(e), (C.c_double * 3)(*E), (C.c_double * 3)(*B), dx, dt)


def electron_kinetic_energy(e):
    f = lib().oracle_electron_kinetic_energy
    f.restype, f.argtypes = C.c_double, [C.c_void_p]
    return f(C.byref(e))


def ion_kinetic_energy(p):
    f = lib().oracle_ion_kinetic_energy
    f.restype, f.argtypes = C.c_double, [C.c_void_p]
    return f(C.byref(p))


def radiate_with_randoms(chi, gamma, u, r1, r2, r3):
    """electron.rs:213-240 with the three uniforms given: (k[3], u'[3], gamma', chi', formation length)."""
    f = lib().oracle_radiate_with_randoms
    f.restype = C.c_int
    f.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
    out = (C.c_double * 9)()
    if f(chi, gamma, (C.c_double * 3)(*u), r1, r2, r3, out):
        raise ValueError("chi >= 100: the reference's sampler panics (photon_emission.rs:143-144)")
    return np.array(out)

