"""opal_amd: MI355X-native implementation of opal's per-timestep PIC loop.

Python mirror of the reference's operator surface -- `Grid` (trait Grid,
src/grid/mod.rs:47-120; YeeGrid, src/grid/yee.rs) and `Population`
(src/particle/mod.rs:141-376) -- over the C ABI in include/opal_amd.h.  Method
names and argument meaning follow the reference; errors the reference raises
as ConfigError or panics surface as `OpalError`.  All compute is in the HIP
library; nothing here has a CPU path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FN1, FN2, FN3

BDY_INTERNAL, BDY_LASER, BDY_ABSORBING, BDY_CONDUCTING = 0, 1, 2, 3
ELECTRON, ION, PHOTON = 0, 1, 2
COMP = {"x": 0, "rho": 1, "jx": 2, "jy": 3, "jz": 4, "Ex": 5, "Ey": 6, "Ez": 7, "Bx": 8, "By": 9, "Bz": 10}
KERNELS = {"fused": 0, "advance": 1, "deposit": 2, "field": 3, "sort": 4, "photon": 5, "radiate": 6}

# src/constants.rs
SPEED_OF_LIGHT = 2.997925e8


class OpalError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[opal status {status}] {message}")
        self.status = status


def _check(rc):
    if rc != 0:
        raise OpalError(rc, _lib.load().opal_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    n = C.c_int(0)
    _check(_lib.load().opal_device_count(C.byref(n)))
    return n.value


def set_elision(on):
    """Measurement aid: False makes the pushes stream every attribute whatever the data hold (see opal_set_elision)."""
    _check(_lib.load().opal_set_elision(1 if on else 0))


def set_device(i):
    _check(_lib.load().opal_set_device(i))


def device_identity(i=-1):
    """(PCI bus id, device name) of HIP ordinal i (default: the current device)."""
    bus, name = C.create_string_buffer(64), C.create_string_buffer(256)
    _check(_lib.load().opal_device_identity(i, bus, 64, name, 256))
    return bus.value.decode(), name.value.decode()


def stream_create():
    """A HIP stream of the library's own runtime (an integer handle for every `stream=` argument here)."""
    h = C.c_void_p()
    _check(_lib.load().opal_stream_create(C.byref(h)))
    return h.value


def stream_destroy(stream):
    _check(_lib.load().opal_stream_destroy(C.c_void_p(stream)))


def design_unbalanced(numtasks, size, dx, min_subsize=8):
    """GridDesign::unbalanced (grid/mod.rs:126-151)."""
    nx = np.zeros(numtasks, dtype=np.uint64)
    off = np.zeros(numtasks, dtype=np.float64)
    _check(_lib.load().opal_design_unbalanced(numtasks, size, dx, min_subsize, _ptr(nx), _ptr(off)))
    return nx, off


def design_balanced(numtasks, size, xmin, dx, ne, min_subsize=8):
    """GridDesign::balanced (grid/mod.rs:157-206)."""
    nx = np.zeros(numtasks, dtype=np.uint64)
    off = np.zeros(numtasks, dtype=np.float64)
    cb = FN1(lambda x, u: float(ne(x)))
    _check(_lib.load().opal_design_balanced(numtasks, size, xmin, dx, min_subsize, cb, None, _ptr(nx), _ptr(off)))
    return nx, off


class Grid:
    """YeeGrid on the GPU."""

    def __init__(self, rank, ngrids, left, right, xmin, dx, nx, offset, stream=None, _handle=None):
        self._lib = _lib.load()
        self._owned = _handle is None
        if _handle is None:
            nx = np.ascontiguousarray(nx, dtype=np.uint64)
            offset = np.ascontiguousarray(offset, dtype=np.float64)
            h = C.c_void_p()
            _check(self._lib.opal_grid_build(C.byref(h), rank, ngrids, left, right, xmin, dx, _ptr(nx), _ptr(offset), stream))
            self.h = h
        else:
            self.h = C.c_void_p(_handle)
        self.left_bdy_size = self._lib.opal_grid_left_bdy_size(self.h)
        self.right_bdy_size = self._lib.opal_grid_right_bdy_size(self.h)
        self.dx = self._lib.opal_grid_dx(self.h)

    # (asked of the library each time: a Simulation's grid changes size when its cuts move, Simulation.set_recut)
    @property
    def size(self):
        return self._lib.opal_grid_size(self.h)

    @property
    def len(self):
        return self._lib.opal_grid_len(self.h)

    @classmethod
    def single(cls, nx, xmin, dx, left=BDY_INTERNAL, right=BDY_INTERNAL, stream=None):
        n, off = design_unbalanced(1, nx, dx)
        return cls(0, 1, left, right, xmin, dx, n, off, stream)

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "h", None):
            self._lib.opal_grid_free(self.h)
            self.h = None

    def rank(self):
        return self._lib.opal_grid_rank(self.h)

    def ngrids(self):
        return self._lib.opal_grid_ngrids(self.h)

    def to_left(self):
        return self._lib.opal_grid_to_left(self.h)

    def to_right(self):
        return self._lib.opal_grid_to_right(self.h)

    def xmin(self):
        return self._lib.opal_grid_xmin(self.h)

    def set_self_exchange(self, on=True):
        """A lone periodic subdomain exchanges with itself through the link instead of wrapping in place."""
        _check(self._lib.opal_grid_set_self_exchange(self.h, 1 if on else 0))

    def clear(self):
        _check(self._lib.opal_grid_clear(self.h))

    def deposit(self, pop, dt):
        _check(self._lib.opal_grid_deposit(self.h, pop.h, dt))

    def synchronize(self, laser_y=None, laser_z=None, t=0.0, dt=0.0, link=None):
        ly = FN2(lambda t_, x_, u: float(laser_y(t_, x_))) if laser_y else FN2()
        lz = FN2(lambda t_, x_, u: float(laser_z(t_, x_))) if laser_z else FN2()
        _check(self._lib.opal_grid_synchronize(self.h, link.h if link else None, ly, lz, None, t, dt))

    def advance(self, dt):
        _check(self._lib.opal_grid_advance(self.h, dt))

    def fields_at(self, c, xi):
        c = np.ascontiguousarray(np.atleast_1d(c), dtype=np.int32)
        xi = np.ascontiguousarray(np.atleast_1d(xi), dtype=np.float64)
        E = np.empty((len(c), 3))
        B = np.empty((len(c), 3))
        _check(self._lib.opal_grid_fields_at(self.h, len(c), _ptr(c), _ptr(xi), _ptr(E), _ptr(B)))
        return E, B

    def em_field_energy(self):
        e = C.c_double()
        _check(self._lib.opal_grid_em_field_energy(self.h, C.byref(e)))
        return e.value

    def interpolate(self):
        out = np.empty((self.size, 11))
        _check(self._lib.opal_grid_interpolate(self.h, _ptr(out)))
        return out

    def write_data(self, directory, index):
        _check(self._lib.opal_grid_write_data(self.h, directory.encode(), index))

    def raw(self, name):
        out = np.empty(self.len)
        _check(self._lib.opal_grid_get_raw(self.h, COMP[name], _ptr(out)))
        return out

    def set_raw(self, name, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self.len,)
        _check(self._lib.opal_grid_set_raw(self.h, COMP[name], _ptr(a)))

    def assume_electrostatic(self):
        """Withdraw the knowledge that B is non-zero somewhere (the pushes check; only speed depends on it)."""
        _check(self._lib.opal_grid_assume_electrostatic(self.h))


class Population:
    """Population<Electron> / Population<Ion> / Population<Photon> on the GPU (SoA in HBM)."""

    def __init__(self, species, capacity=0, stream=None, _handle=None):
        self._lib = _lib.load()
        self.species = species
        self._owned = _handle is None
        if _handle is None:
            h = C.c_void_p()
            _check(self._lib.opal_pop_create(C.byref(h), species, capacity, stream))
            self.h = h
        else:
            self.h = C.c_void_p(_handle)

    @classmethod
    def new(cls, species, npc, number_density, ux, uy, uz, grid, seed, dt, stream=None):
        """Population::new (mod.rs:172-203)."""
        lib = _lib.load()
        d = FN1(lambda x, u: float(number_density(x)))
        fx = FN3(lambda x, ur, nr, u: float(ux(x, ur, nr)))
        fy = FN3(lambda x, ur, nr, u: float(uy(x, ur, nr)))
        fz = FN3(lambda x, ur, nr, u: float(uz(x, ur, nr)))
        h = C.c_void_p()
        _check(lib.opal_pop_new(C.byref(h), species, npc, d, fx, fy, fz, None, grid.h, seed, dt, stream))
        p = cls(species, _handle=h.value)
        p._owned = True
        return p

    @classmethod
    def from_soa(cls, species, cell, x, ux, uy, uz, w, dx, dt, capacity=None, stream=None):
        n = len(cell)
        p = cls(species, capacity if capacity is not None else n + n // 8 + 1024, stream)
        p.upload(cell, x, ux, uy, uz, w, dx, dt)
        return p

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "h", None):
            self._lib.opal_pop_free(self.h)
            self.h = None

    def upload(self, cell, x, ux, uy, uz, w, dx, dt):
        cell = np.ascontiguousarray(cell, dtype=np.int32)
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, ux, uy, uz, w)]
        _check(self._lib.opal_pop_upload_soa(self.h, len(cell), _ptr(cell), *[_ptr(a) for a in arrs], dx, dt))

    def set_charge_state(self, z_star, z, a):
        _check(self._lib.opal_pop_set_charge_state(self.h, z_star, z, a))

    def __len__(self):
        n = C.c_size_t()
        _check(self._lib.opal_pop_len(self.h, C.byref(n)))
        return n.value

    def live(self):
        """Live particles from the device's counters (no download of the cell array, unlike len())."""
        n = C.c_size_t()
        _check(self._lib.opal_pop_live(self.h, C.byref(n)))
        return n.value

    def soa(self, prev_x=False, fields=("x", "ux", "uy", "uz", "w")):
        """Live particles in slot order; `fields` limits what is copied back (cell always is)."""
        n = len(self)
        cell = np.empty(n, dtype=np.int32)
        out = {k: np.empty(n) for k in fields}
        px = np.empty(n) if prev_x else None
        _check(self._lib.opal_pop_download_soa(self.h, _ptr(cell), _ptr(out.get("x")), _ptr(px), _ptr(out.get("ux")),
                                               _ptr(out.get("uy")), _ptr(out.get("uz")), _ptr(out.get("w"))))
        out["cell"] = cell
        if prev_x:
            out["prev_x"] = px
        return out

    def advance(self, grid, dt, link=None):
        _check(self._lib.opal_pop_advance(self.h, link.h if link else None, grid.h, dt))

    def advance_deposit(self, grid, dt, link=None):
        _check(self._lib.opal_pop_advance_deposit(self.h, link.h if link else None, grid.h, dt))

    def total_kinetic_energy(self):
        e = C.c_double()
        _check(self._lib.opal_pop_total_kinetic_energy(self.h, C.byref(e)))
        return e.value

    def sort(self, grid):
        _check(self._lib.opal_pop_sort(self.h, grid.h))

    def set_sort_interval(self, interval):
        _check(self._lib.opal_pop_set_sort_interval(self.h, interval))

    def counters(self):
        out = (C.c_uint32 * 8)()
        _check(self._lib.opal_pop_counters(self.h, out))
        keys = ("slots", "deleted", "tile_misses", "steps_since_sort", "ppt", "live_at_sort", "nonempty_cells", "slots_upper")
        return dict(zip(keys, [int(v) for v in out]))

    def histogram(self, grid, spec):
        """One `output:` spec -> dict(cts, nbins, min, bin_sz, total) for this subdomain
        (Histogram::generate_1d/2d with HeightSpec::Density, hgram.rs:168-392)."""
        dim = C.c_int()
        mn = (C.c_double * 2)()
        mx = (C.c_double * 2)()
        n = C.c_size_t()
        _check(self._lib.opal_pop_hist_range(self.h, grid.h, spec.encode(), C.byref(dim), mn, mx, C.byref(n)))
        if n.value == 0:
            return None
        h = C.c_void_p()
        _check(self._lib.opal_pop_hist_fill(self.h, grid.h, spec.encode(), mn, mx, n.value, C.byref(h)))
        try:
            _check(self._lib.opal_hist_finish(h))
            nb = (C.c_size_t * 2)(1, 1)
            hmin = (C.c_double * 2)()
            bsz = (C.c_double * 2)()
            _check(self._lib.opal_hist_dims(h, C.byref(dim), nb, hmin, bsz))
            ln = self._lib.opal_hist_len(h)
            cts = np.ctypeslib.as_array(self._lib.opal_hist_counts(h), shape=(ln,)).copy()
            return {"cts": cts, "nbins": [int(nb[0]), int(nb[1])], "min": np.array(hmin), "max": np.array(mx),
                    "bin_sz": np.array(bsz), "total": self._lib.opal_hist_total(h), "dim": dim.value}
        finally:
            self._lib.opal_hist_free(h)

    def write_data(self, grid, directory, index, name, specs):
        """Population::write_data (mod.rs:383-568) for one subdomain."""
        arr = (C.c_char_p * len(specs))(*[s.encode() for s in specs])
        _check(self._lib.opal_pop_write_data(self.h, grid.h, directory.encode(), index, name.encode(), arr, len(specs)))

    def track_extras(self, on=True):
        """on: False/0 off, True/1 chi work y z (photons: chi y z parent id), 2 also the optical depth and id
        photon emission needs (electrons)."""
        _check(self._lib.opal_pop_track_extras(self.h, int(on)))

    def emission_state(self):
        """(tau, id) of every electron in slot order; for photons (None, parent id)."""
        n = len(self)
        tau = np.empty(n) if self.species == ELECTRON else None
        ids = np.empty(n, dtype=np.uint64)
        _check(self._lib.opal_pop_download_emission_state(self.h, _ptr(tau) if tau is not None else None, _ptr(ids)))
        return tau, ids

    def ids(self):
        """The ids of the live particles in slot order (electrons: own id; photons: the parent electron's)."""
        out = np.empty(len(self), dtype=np.uint64)
        _check(self._lib.opal_pop_download_emission_state(self.h, None, _ptr(out)))
        return out

    def set_emission_state(self, tau=None, ids=None):
        """Overwrite the optical depths / ids of the particles just uploaded (same order)."""
        t = np.ascontiguousarray(tau, dtype=np.float64) if tau is not None else None
        i = np.ascontiguousarray(ids, dtype=np.uint64) if ids is not None else None
        n = len(t) if t is not None else len(i)
        _check(self._lib.opal_pop_upload_emission_state(self.h, n, _ptr(t) if t is not None else None,
                                                        _ptr(i) if i is not None else None))

    def track_fields(self, fields):
        """Bit mask of TRACK_* flags: carry only what is named (opal_pop_track_fields)."""
        _check(self._lib.opal_pop_track_fields(self.h, int(fields)))

    def absorption_state(self):
        """(tau_abs, tau_st, birth_time) of every photon in slot order (photon.rs:28-30)."""
        n = len(self)
        out = [np.empty(n) for _ in range(3)]
        _check(self._lib.opal_pop_download_absorption_state(self.h, *[_ptr(a) for a in out]))
        return tuple(out)

    def set_absorption_state(self, tau_abs=None, tau_st=None, birth_time=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (tau_abs, tau_st, birth_time)]
        n = max(len(a) for a in arrs if a is not None)
        _check(self._lib.opal_pop_upload_absorption_state(self.h, n, *[_ptr(a) for a in arrs]))

    def set_chi(self, chi):
        chi = np.ascontiguousarray(chi, dtype=np.float64)
        _check(self._lib.opal_pop_upload_chi(self.h, len(chi), _ptr(chi)))

    def set_yz(self, y, z):
        y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (y, z))
        _check(self._lib.opal_pop_upload_yz(self.h, len(y), _ptr(y), _ptr(z)))

    def extras(self):
        n = len(self)
        out = {k: np.empty(n) for k in ("chi", "work", "y", "z")}
        _check(self._lib.opal_pop_download_extras(self.h, *[_ptr(out[k]) for k in ("chi", "work", "y", "z")]))
        return out


def emit_radiation(electrons, photons, seed, step, t, min_energy=None, max_angle=None, max_formation_length=None,
                   count=True):
    """emit_radiation (particle/interactions.rs:45-107); returns the number of electrons that emitted, or with
    count=False nothing: the library then reads nothing back from the device (the driver's mode)."""
    f = lambda v: -1.0 if v is None else float(v)
    n = C.c_size_t()
    _check(_lib.load().opal_emit_radiation(electrons.h, photons.h, seed, step, t, f(min_energy), f(max_angle),
                                           f(max_formation_length), C.byref(n) if count else None))
    return n.value if count else None


TRACK_CHI, TRACK_WORK, TRACK_YZ, TRACK_EMISSION, TRACK_ID, TRACK_ABSORPTION = 1, 2, 4, 8, 16, 32


def absorb(electrons, photons, grid, seed, step, t, dt, max_displacement=None, stop_time=None, count=True):
    """absorb (particle/interactions.rs:145-340); returns (photons absorbed, emissions stimulated) -- or, with
    count=False, nothing: the form the time loop uses, in which no count is read back from the device."""
    f = lambda v: -1.0 if v is None else float(v)
    if not count:
        _check(_lib.load().opal_absorb(electrons.h, photons.h, grid.h, seed, step, t, dt, f(max_displacement), f(stop_time),
                                       None, None))
        return None
    na, ns = C.c_size_t(), C.c_size_t()
    _check(_lib.load().opal_absorb(electrons.h, photons.h, grid.h, seed, step, t, dt, f(max_displacement), f(stop_time),
                                   C.byref(na), C.byref(ns)))
    return na.value, ns.value


def qed_airy(x):
    """airy_ai_for_positive on the device; NaN where the reference returns None."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(len(x))
    _check(_lib.load().opal_qed_airy_batch(len(x), _ptr(x), _ptr(out)))
    return out


def qed_cross_sections(k, p, chi_gamma, chi_e):
    """(absorption, stimulated emission) scaled cross sections on the device for n pairs; k, p are (n, 4)."""
    k = np.ascontiguousarray(k, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    cg = np.ascontiguousarray(chi_gamma, dtype=np.float64)
    ce = np.ascontiguousarray(chi_e, dtype=np.float64)
    a, b = np.empty(len(cg)), np.empty(len(cg))
    _check(_lib.load().opal_qed_cross_sections_batch(len(cg), _ptr(k), _ptr(p), _ptr(cg), _ptr(ce), _ptr(a), _ptr(b)))
    return a, b


def qed_sample(chi, gamma, r1, r2, r3, classical=False):
    """photon_emission::sample / classical_sample evaluated on the device; returns an (n, 3) array."""
    a = [np.ascontiguousarray(np.broadcast_to(v, np.shape(chi)), dtype=np.float64) for v in (chi, gamma, r1, r2, r3)]
    out = np.empty((len(a[0]), 3))
    _check(_lib.load().opal_qed_sample_batch(len(a[0]), *[_ptr(v) for v in a], 1 if classical else 0, _ptr(out)))
    return out


def qed_radiate(chi, u, r1, r2, r3):
    """Electron::radiate's event on the device with the three uniforms given: (n, 9) array of the photon's k[3], the
    electron's u[3] after the recoil, its gamma and chi, the formation length (electron.rs:213-240)."""
    chi = np.ascontiguousarray(chi, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    a = [chi] + [np.ascontiguousarray(u[:, k]) for k in range(3)] + [np.ascontiguousarray(r, dtype=np.float64) for r in (r1, r2, r3)]
    out = np.empty((len(chi), 9))
    _check(_lib.load().opal_qed_radiate_batch(len(chi), *[_ptr(v) for v in a], _ptr(out)))
    return out


def qed_randoms(seed, ids, step):
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    out = np.empty((len(ids), 4))
    _check(_lib.load().opal_qed_randoms_batch(len(ids), seed, _ptr(ids), step, _ptr(out)))
    return out


def qed_rate(chi, gamma):
    chi = np.ascontiguousarray(chi, dtype=np.float64)
    gamma = np.ascontiguousarray(np.broadcast_to(gamma, chi.shape), dtype=np.float64)
    out = np.empty(len(chi))
    _check(_lib.load().opal_qed_rate_batch(len(chi), _ptr(chi), _ptr(gamma), _ptr(out)))
    return out


class Simulation:
    """main() of the reference (src/main.rs) for one subdomain."""

    def __init__(self, yaml_path=None, yaml_text=None, output_dir=".", rank=0, numtasks=1, link=None, stream=None,
                 collectives=None):
        """collectives: an exchange.TorchCollectives for run()'s output-time reductions; built from `link`
        (same process group and device) when omitted."""
        self._lib = _lib.load()
        h = C.c_void_p()
        lk = link.h if link else None
        if yaml_path is not None:
            _check(self._lib.opal_sim_create(C.byref(h), str(yaml_path).encode(), rank, numtasks, lk, stream))
        else:
            _check(self._lib.opal_sim_create_from_string(C.byref(h), yaml_text.encode(), output_dir.encode(), rank,
                                                         numtasks, lk, stream))
        self.h = h
        self._link = link
        if numtasks > 1:
            if collectives is None and link is not None and hasattr(link, "transport"):
                from .exchange import TorchCollectives
                collectives = TorchCollectives(link.device, link.transport.group)
            if collectives is not None:
                _check(self._lib.opal_sim_set_collectives(h, collectives.allreduce_cb, collectives.gather_cb, None))
        self._collectives = collectives
        self.grid = Grid(0, 0, 0, 0, 0, 0, None, None, _handle=self._lib.opal_sim_grid(h))
        e = self._lib.opal_sim_electrons(h)
        i = self._lib.opal_sim_ions(h)
        self.electrons = Population(ELECTRON, _handle=e) if e else None
        self.ions = Population(ION, _handle=i) if i else None
        ph = self._lib.opal_sim_photons(h) if hasattr(self._lib, "opal_sim_photons") else None
        self.photons = Population(PHOTON, _handle=ph) if ph else None  # only with qed.photon_emission

    def __del__(self):
        if getattr(self, "h", None):
            self._lib.opal_sim_free(self.h)
            self.h = None

    def step(self, n=1):
        _check(self._lib.opal_sim_step(self.h, n))

    def sync(self):
        _check(self._lib.opal_sim_sync(self.h))

    def run(self):
        _check(self._lib.opal_sim_run(self.h))

    @property
    def overlapped_steps(self):
        return self._lib.opal_sim_overlapped_steps(self.h)

    def set_overlap(self, on):
        """Overlap the neighbour exchange with the bulk of the push in decomposed fused runs (default off)."""
        _check(self._lib.opal_sim_set_overlap(self.h, 1 if on else 0))

    def set_fused(self, fused):
        _check(self._lib.opal_sim_set_fused(self.h, 1 if fused else 0))

    def set_recut(self, every, threshold=1.1, max_shift=64):
        """Cuts that follow the plasma (opal_sim_set_recut): every `every` steps the subdomains compare their particle
        loads and, above `threshold` (max / mean), move their cuts by up to max_shift cells.  every = 0: the reference's
        partition, made once at t = 0 (grid/mod.rs:157-206)."""
        _check(self._lib.opal_sim_set_recut(self.h, int(every), float(threshold), int(max_shift)))

    def recut_now(self):
        """One check + re-cut at once, whatever the threshold (collective)."""
        _check(self._lib.opal_sim_recut_now(self.h))

    def recut_stats(self, numtasks):
        """dict: checks, recuts, imbalance (max / mean at the last check; 0.0: never checked), imbalance_worst (since the
        previous call), sizes (interior cells of every subdomain now), seconds (host time spent in checks and re-cuts so far)."""
        a, b, c, d, t = C.c_size_t(), C.c_size_t(), C.c_double(), C.c_double(), C.c_double()
        nx = (C.c_size_t * numtasks)()
        _check(self._lib.opal_sim_recut_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), nx, C.byref(t)))
        v = (C.c_double * (numtasks + 1))()
        ds = C.c_size_t()
        _check(self._lib.opal_sim_drift_stats(self.h, v, C.byref(ds)))
        return {"checks": a.value, "recuts": b.value, "imbalance": c.value, "imbalance_worst": d.value, "sizes": list(nx),
                "seconds": t.value, "drift_cells_per_step": list(v)[1:numtasks], "drift_steps": ds.value}

    @property
    def t(self):
        return self._lib.opal_sim_time(self.h)

    @property
    def dt(self):
        return self._lib.opal_sim_dt(self.h)

    @property
    def total_steps(self):
        return self._lib.opal_sim_total_steps(self.h)

    @property
    def steps_between_outputs(self):
        return self._lib.opal_sim_steps_between_outputs(self.h)

    @property
    def n_outputs(self):
        return self._lib.opal_sim_n_outputs(self.h)

    def energies(self):
        f, e, i = C.c_double(), C.c_double(), C.c_double()
        _check(self._lib.opal_sim_energies(self.h, C.byref(f), C.byref(e), C.byref(i)))
        return f.value, e.value, i.value


def recut_plan(cuts, cost, most=None, min_size=8, max_shift=64, edge_room=None):
    """opal_recut_plan: (new cuts, imbalance before, imbalance after); host arithmetic only (no GPU)."""
    cuts = np.ascontiguousarray(cuts, dtype=np.uint64)
    n = len(cuts) - 1
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    assert len(cost) == int(cuts[-1])
    most = None if most is None else np.ascontiguousarray(most, dtype=np.float64)
    room = None if edge_room is None else np.ascontiguousarray(edge_room, dtype=np.float64)
    out = np.zeros(n + 1, dtype=np.uint64)
    a, b = C.c_double(), C.c_double()
    _check(_lib.load().opal_recut_plan(n, _ptr(cuts), _ptr(cost), _ptr(most) if most is not None else None, min_size, max_shift,
                                       _ptr(room) if room is not None else None, _ptr(out), C.byref(a), C.byref(b)))
    return out.astype(np.int64), a.value, b.value


class Config:
    """Config (src/setup.rs:84-285)."""

    def __init__(self, path=None, text=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        if path is not None:
            _check(self._lib.opal_config_from_file(C.byref(h), str(path).encode()))
        else:
            _check(self._lib.opal_config_from_string(C.byref(h), text.encode()))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self._lib.opal_config_free(self.h)
            self.h = None

    def contains(self, section):
        return bool(self._lib.opal_config_contains(self.h, section.encode()))

    def read_f64(self, section, field):
        v = C.c_double()
        _check(self._lib.opal_config_read_f64(self.h, section.encode(), field.encode(), C.byref(v)))
        return v.value

    def read_usize(self, section, field):
        v = C.c_size_t()
        _check(self._lib.opal_config_read_usize(self.h, section.encode(), field.encode(), C.byref(v)))
        return v.value

    def read_bool(self, section, field):
        v = C.c_int()
        _check(self._lib.opal_config_read_bool(self.h, section.encode(), field.encode(), C.byref(v)))
        return bool(v.value)

    def func(self, section, field, args):
        """Config::func/func2/func3: returns a Python callable of len(args) floats."""
        names = (C.c_char_p * len(args))(*[a.encode() for a in args])

        def f(*vals):
            v = (C.c_double * len(args))(*vals)
            out = C.c_double()
            _check(self._lib.opal_config_eval(self.h, section.encode(), field.encode(), len(args), names, v, C.byref(out)))
            return out.value

        return f

    def eval(self, expr):
        out = C.c_double()
        _check(self._lib.opal_config_eval_expr(self.h, expr.encode(), C.byref(out)))
        return out.value


def prof_enable(stride=1):
    """0/False = off; n = time every n-th launch of each tracked kernel with HIP events."""
    _check(_lib.load().opal_prof_enable(int(stride)))


def prof_reset():
    _check(_lib.load().opal_prof_reset())


def prof_seen(kernel):
    n = C.c_uint64()
    _check(_lib.load().opal_prof_seen(KERNELS[kernel], C.byref(n)))
    return n.value


def prof_forms():
    """{form: launches} of the fused push since prof_reset; form bits: 1 electrostatic, 2 rho deposited, 4 / 8 the
    electrons' / ions' weight from the launch arguments, 16 single-species launch (opal_prof_forms)."""
    a = (C.c_uint64 * 32)()
    _check(_lib.load().opal_prof_forms(a))
    return {i: int(a[i]) for i in range(32) if a[i]}


def prof_get(kernel):
    n = C.c_uint64()
    ms = C.c_double()
    _check(_lib.load().opal_prof_get(KERNELS[kernel], C.byref(n), C.byref(ms)))
    return n.value, ms.value
