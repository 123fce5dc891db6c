"""N subdomains of a decomposed HIP run inside ONE process, one host thread and one HIP stream each, exchanging through
host buffers: how the 8-way partitions of BASELINE configs[2]-[3] are executed on a box with a single GPU (which
admits at most 6 processes on its card, so 8 ranks cannot be 8 processes there).

The link is the library's ordinary callback transport (opal_link_create_layout with host staging, the path gloo ranks
use); the callback is a rendezvous between the threads: every subdomain posts its two send buffers, waits at a
barrier, copies what its neighbours posted, waits again.  ctypes releases the GIL around the C calls, so the
subdomains' steps really interleave as separate ranks' would.
"""
import ctypes as C
import sys
import threading
import traceback

import numpy as np

import opal_amd as oa
from opal_amd import _lib


class World:
    def __init__(self, n, timeout=120.0):
        self.n = n
        self.barrier = threading.Barrier(n, timeout=timeout)
        self.slots = [None] * n
        self.errors = []


class InProcLink:
    """opal_link_t whose transport is a rendezvous of the threads of `world` (host-staged messages)."""

    def __init__(self, world, rank, particle_capacity=1024, stream=None, photon_emission=False, periodic=True, growth=False):
        self._lib = _lib.load()
        self.world, self.rank = world, rank
        self.periodic = periodic
        self._cb = _lib.EXCHANGE_FN(self._exchange)
        self._cb2 = _lib.EXCHANGE2_FN(self._exchange2)
        slots, fields = (3, 12) if photon_emission else (2, 6)
        h = C.c_void_p()
        if self._lib.opal_link_create_layout(C.byref(h), rank, world.n, 0, particle_capacity, slots, fields, self._cb, None, stream):
            raise RuntimeError(self._lib.opal_last_error().decode())
        self.h = h
        if self._lib.opal_link_set_exchange2(h, self._cb2):
            raise RuntimeError(self._lib.opal_last_error().decode())
        if growth and self._lib.opal_link_set_growth(h, 1):  # (off by default for transports that are told a size)
            raise RuntimeError(self._lib.opal_last_error().decode())
        self.calls = 0
        self.bytes = 0

    def growth(self):
        a, b, n = C.c_size_t(), C.c_size_t(), C.c_uint()
        self._lib.opal_link_growth(self.h, C.byref(a), C.byref(b), C.byref(n))
        return a.value, b.value, n.value

    def __del__(self):
        if getattr(self, "h", None):
            self._lib.opal_link_free(self.h)
            self.h = None

    def _exchange(self, user, sl, sr, rl, rr, n):
        return self._exchange2(user, sl, sr, rl, rr, n, n)

    def _exchange2(self, user, sl, sr, rl, rr, n_left, n_right):
        try:
            w, r = self.world, self.rank
            w.slots[r] = (sl, sr, n_left, n_right)
            w.barrier.wait()
            if rl:  # what my left neighbour sent to the right
                lsl, lsr, lnl, lnr = w.slots[(r - 1) % w.n]
                assert lsr and lnr == n_left, (r, lnr, n_left)   # both ends of an edge agree on its message size
                C.memmove(rl, lsr, n_left)
                self.bytes += n_left
            if rr:
                rsl, rsr, rnl, rnr = w.slots[(r + 1) % w.n]
                assert rsl and rnl == n_right, (r, rnl, n_right)
                C.memmove(rr, rsl, n_right)
                self.bytes += n_right
            w.barrier.wait()  # nobody refills a send buffer before its neighbours have copied it
            self.calls += 1
            return 0
        except Exception:  # never let an exception cross the C boundary
            traceback.print_exc(file=sys.stderr)
            self.world.barrier.abort()
            return 1


class InProcCollectives:
    """opal_sim_set_collectives for the threads of `world`: all-reduce and gather-to-rank-0 as rendezvous (what
    exchange.TorchCollectives does between processes).  Pass as Simulation(collectives=...)."""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        if not hasattr(world, "coll"):
            world.coll = [None] * world.n
        self.allreduce_cb = _lib.ALLREDUCE_FN(self._allreduce)
        self.gather_cb = _lib.GATHER_FN(self._gather)

    def _allreduce(self, user, values, n, op):
        try:
            w = self.world
            mine = np.ctypeslib.as_array(values, shape=(n,))
            w.coll[self.rank] = mine.copy()
            w.barrier.wait()
            parts = np.stack(w.coll)  # rank order: every subdomain reduces in the same order and gets the same bits
            out = parts.sum(axis=0) if op == 0 else parts.min(axis=0) if op == 1 else parts.max(axis=0)
            w.barrier.wait()
            mine[:] = out
            return 0
        except Exception:
            traceback.print_exc(file=sys.stderr)
            self.world.barrier.abort()
            return 1

    def _gather(self, user, local, n_local, out, counts, numtasks):
        try:
            w = self.world
            w.coll[self.rank] = np.ctypeslib.as_array(local, shape=(n_local,)).copy() if n_local else np.zeros(0)
            w.barrier.wait()
            if self.rank == 0:
                cnt = [counts[r] for r in range(numtasks)]
                dst = np.ctypeslib.as_array(out, shape=(sum(cnt),))
                at = 0
                for r in range(numtasks):
                    dst[at:at + cnt[r]] = w.coll[r][:cnt[r]]
                    at += cnt[r]
            w.barrier.wait()
            return 0
        except Exception:
            traceback.print_exc(file=sys.stderr)
            self.world.barrier.abort()
            return 1


def run_world(n, body, timeout=600.0):
    """Runs body(rank, world) -> result on n threads; returns the results in rank order, or raises the first error."""
    world = World(n)
    results = [None] * n

    def target(r):
        try:
            oa.set_device(0)
            results[r] = body(r, world)
        except BaseException as e:  # noqa: BLE001
            world.errors.append((r, e, traceback.format_exc()))
            world.barrier.abort()

    threads = [threading.Thread(target=target, args=(r,), daemon=True) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
        if t.is_alive():
            world.barrier.abort()
            raise TimeoutError("a subdomain thread did not finish")
    if world.errors:
        r, e, tb = world.errors[0]
        raise RuntimeError(f"subdomain {r} failed:\n{tb}")
    return results


def soa_dump(sim, prefix=""):
    out = {}
    for name, p in (("e", sim.electrons), ("i", sim.ions)):
        if p is None:
            continue
        for k, v in p.soa().items():
            out[f"{prefix}{name}_{k}"] = v
    return out


def grid_dump(g):
    return {k: g.raw(k) for k in ("Ex", "Ey", "Ez", "By", "Bz", "jx", "jy", "jz", "rho")}


def by_position(d, name):
    """A population's arrays ordered by (cell, x): a canonical order for comparing two runs particle by particle."""
    order = np.lexsort((d[f"{name}_x"], d[f"{name}_cell"]))
    return {k: d[f"{name}_{k}"][order] for k in ("cell", "x", "ux", "uy", "uz", "w")}
