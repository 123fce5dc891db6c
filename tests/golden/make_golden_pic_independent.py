#!/usr/bin/env python3
"""Generates tests/golden/pic_step_vectors_independent.json — the hot path, restated a second time.

What this is.  The Rust reference cannot be built here (no cargo/rustc), so no vector can be captured
from it, and tests/golden/pic_step_vectors.json is written *by* oracle/ (a regression freeze of the
oracle, nothing more).  This file is the other leg: a restatement of the per-timestep loop written
straight from the reference's source text, in plain Python floats (IEEE binary64, no fused
multiply-add, `math` only, one Python operation per Rust operation, in the Rust's evaluation order) —
the way tests/golden/make_golden_qed_sample.py does it for the photon sampler.

It shares NOTHING with oracle/ or opal_amd/csrc/: it imports neither, was written without consulting
either, and keeps the reference's own data layout (an array of `Cell`s, particle structs in a list).
Its outputs are committed as hex floats.  tests/test_golden_independent.py then requires
  * the C oracle to reproduce every vector BIT FOR BIT (CPU, -m "not gpu"), and
  * the HIP operators, through the C ABI, to 1e-12 of each quantity's scale (-m gpu).
A mis-reading of the source would have to be made identically twice, in two languages and two data
layouts, to pass.

The restatement is itself held to every answer the reference's own tests hold for this path
(`self_checks` in the output): laser_bc / conducting_bc (1e-3), periodic_bc (1e-6) — yee.rs:903-1079 —
and static_magnetic_field / static_electric_field (1e-6) — electron.rs:401-436.

Runs in the build container only (nothing here reads /root/reference; the citations are for the
reader).  Usage: python tests/golden/make_golden_pic_independent.py [--skip-self-checks]

Citations are file:line under the reference's src/.
"""
import json
import math
import os
import random
import sys

# ---------------------------------------------------------------------------------------------------
# constants.rs:4-27 (deliberately truncated literals; SPEED_OF_LIGHT_SQD is NOT SPEED_OF_LIGHT squared)
# ---------------------------------------------------------------------------------------------------
SPEED_OF_LIGHT = 2.997925e8
SPEED_OF_LIGHT_SQD = 89875517873681764.0
VACUUM_PERMITTIVITY = 8.854188e-12
VACUUM_PERMEABILITY = 1.256637e-6
ELECTRON_CHARGE = -1.602177e-19
ELEMENTARY_CHARGE = -ELECTRON_CHARGE
ELECTRON_MASS = 9.109383e-31
PROTON_MASS = 1.672622e-27
ELECTRON_MASS_MEV = 0.510999
CRITICAL_FIELD = 1.323285e18

INTERNAL, LASER, ABSORBING, CONDUCTING = "internal", "laser", "absorbing", "conducting"  # grid/mod.rs:22-28

# yee.rs:239-242
GHOST_SIZE = 4
LASER_BDY_SIZE = 4
ABSORBING_BDY_SIZE = 200
CONDUCTING_BDY_SIZE = 4


def powi2(x):
    """f64::powi(2): one multiplication."""
    return x * x


# ---------------------------------------------------------------------------------------------------
# particle/vec3.rs:39-112 — tuples of three floats
# ---------------------------------------------------------------------------------------------------
def vadd(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def vdot(a, b):  # vec3.rs:56-61: x*x + y*y + z*z, left to right
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def smul(s, v):  # vec3.rs:72-77 (f64 * Vec3) and :64-69 (Vec3 * f64): the same products
    return (s * v[0], s * v[1], s * v[2])


def vdiv(v, s):  # vec3.rs:88-93
    return (v[0] / s, v[1] / s, v[2] / s)


def vcross(a, b):  # vec3.rs:105-111
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def norm_sqr(a):  # vec3.rs:113-115
    return vdot(a, a)


# ---------------------------------------------------------------------------------------------------
# grid/yee.rs: Cell, weight, flux
# ---------------------------------------------------------------------------------------------------
class Cell:  # yee.rs:84-92
    __slots__ = ("x", "rho", "j", "E", "B")

    def __init__(self, x):
        self.x = x
        self.rho = 0.0
        self.j = [0.0, 0.0, 0.0]
        self.E = [0.0, 0.0, 0.0]
        self.B = [0.0, 0.0, 0.0]

    def copy(self):
        c = Cell(self.x)
        c.rho = self.rho
        c.j = list(self.j)
        c.E = list(self.E)
        c.B = list(self.B)
        return c

    def overlay_ghost(self, other):  # yee.rs:97-104
        self.rho += other.rho
        self.j[0] += other.j[0]
        self.j[1] += other.j[1]
        self.j[2] += other.j[2]
        self.E = list(other.E)
        self.B = list(other.B)

    def overlay(self, other):  # yee.rs:108-113
        self.rho += other.rho
        self.j[0] += other.j[0]
        self.j[1] += other.j[1]
        self.j[2] += other.j[2]


BRANCHES = {}


def hit(name):
    BRANCHES[name] = BRANCHES.get(name, 0) + 1


def weight(xi):  # yee.rs:140-149
    xhat = abs(xi)
    if xhat > 1.5:
        hit("weight: xhat > 1.5")
        return 0.0
    elif xhat < 0.5:
        hit("weight: xhat < 0.5")
        return 0.75 - powi2(xhat)
    else:
        hit("weight: 0.5 <= xhat <= 1.5")
        return 1.125 - 1.5 * xhat + 0.5 * powi2(xhat)


def flux(x_i, x_f):  # yee.rs:185-204
    if abs(x_i) < 1.0:
        if abs(x_f) >= 1.0:
            hit("flux: |x_i| < 1, |x_f| >= 1")
            v = 0.5 * powi2(1.0 - abs(x_i))
            return math.copysign(v, -x_i)
        elif x_i * x_f >= 0.0:
            hit("flux: both inside, same sign")
            v = 0.5 * powi2(1.0 - abs(x_f)) - 0.5 * powi2(1.0 - abs(x_i))
            return math.copysign(v, x_i - x_f)
        else:
            hit("flux: both inside, different signs")
            v = abs(x_i) * (1.0 - 0.5 * abs(x_i)) + abs(x_f) * (1.0 - 0.5 * abs(x_f))
            return math.copysign(v, x_i)
    elif abs(x_f) < 1.0:
        hit("flux: |x_i| >= 1, |x_f| < 1")
        v = 0.5 * powi2(1.0 - abs(x_f))
        return math.copysign(v, x_f)
    else:
        hit("flux: both outside")
        return 0.0


# ---------------------------------------------------------------------------------------------------
# grid/mod.rs:126-206 — GridDesign
# ---------------------------------------------------------------------------------------------------
class GridDesign:  # grid/mod.rs:33-42
    def __init__(self, id_, numtasks, left, right, xmin, dx, nx, offset):
        self.id, self.numtasks, self.left, self.right = id_, numtasks, left, right
        self.xmin, self.dx, self.nx, self.offset = xmin, dx, nx, offset


def _offsets(nx, dx):  # the scan of grid/mod.rs:131-138 / 186-193
    out, rt = [], 0
    for n in nx:
        out.append(float(rt) * dx)
        rt = rt + n
    return out


def design_unbalanced(id_, numtasks, size, xmin, dx, min_subsize, left, right):  # grid/mod.rs:126-151
    subsize = max(size // numtasks, min_subsize)
    nx = [subsize] * numtasks
    return GridDesign(id_, numtasks, left, right, xmin, dx, nx, _offsets(nx, dx))


def design_balanced(id_, numtasks, size, xmin, dx, min_subsize, left, right, ne):  # grid/mod.rs:157-206
    if numtasks == 0:
        return design_unbalanced(id_, numtasks, size, xmin, dx, min_subsize, left, right)
    ppc = []
    for n in range(size - min_subsize):
        x = xmin + float(n) * dx
        ppc.append(dx * ne(x))
    cumsum, total = [], 0.0
    for np_ in ppc:
        total += np_
        cumsum.append(total)
    target = cumsum[-1] / float(numtasks)
    ncells, start = [], 0
    for p in range(1, numtasks):
        i = None
        for k, cs in enumerate(cumsum[start + min_subsize:]):
            if cs >= target * float(p):
                i = k
                break
        if i is None:
            raise RuntimeError("position(..).unwrap() on None: the reference panics (grid/mod.rs:179)")
        ncells.append(i + min_subsize)
        start += i + min_subsize
    ndone = sum(ncells)
    ncells.append(size - ndone)
    return GridDesign(id_, numtasks, left, right, xmin, dx, ncells, _offsets(ncells, dx))


# ---------------------------------------------------------------------------------------------------
# grid/yee.rs:255-867 — YeeGrid
# ---------------------------------------------------------------------------------------------------
class YeeGrid:
    def __init__(self, g):  # build, yee.rs:272-319
        id_, numtasks = g.id, g.numtasks
        ncells, offset = g.nx[id_], g.offset[id_]
        if id_ == 0 and g.left == LASER:
            self.left_bdy, self.left_bdy_size = LASER, LASER_BDY_SIZE
        else:
            self.left_bdy, self.left_bdy_size = INTERNAL, GHOST_SIZE
        if id_ == numtasks - 1:
            if g.right == ABSORBING:
                self.right_bdy, self.right_bdy_size = ABSORBING, ABSORBING_BDY_SIZE
            elif g.right == CONDUCTING:
                self.right_bdy, self.right_bdy_size = CONDUCTING, CONDUCTING_BDY_SIZE
            else:
                self.right_bdy, self.right_bdy_size = INTERNAL, GHOST_SIZE
        else:
            self.right_bdy, self.right_bdy_size = INTERNAL, GHOST_SIZE
        self.id, self.ngrids = id_, numtasks
        self.rank_to_left = (id_ - 1 + numtasks) % numtasks
        self.rank_to_right = (id_ + 1) % numtasks
        self.size, self.dx = ncells, g.dx
        n = ncells + self.left_bdy_size + self.right_bdy_size
        self.cell = [Cell(g.xmin + offset + (float(c) - float(self.left_bdy_size)) * g.dx) for c in range(n)]

    def to_left(self):  # yee.rs:329-335
        return self.rank_to_left if self.left_bdy == INTERNAL else None

    def to_right(self):  # yee.rs:337-343
        return self.rank_to_right if self.right_bdy == INTERNAL else None

    def xmin(self):  # yee.rs:531-533
        return self.cell[self.left_bdy_size].x

    def advance(self, dt):  # yee.rs:345-349
        self.advance_B(0.5 * dt)
        self.advance_E(dt)
        self.advance_B(0.5 * dt)

    def send_buffers(self):  # yee.rs:353-354
        return ([c.copy() for c in self.cell[0:2 * GHOST_SIZE]], [c.copy() for c in self.cell[-2 * GHOST_SIZE:]])

    def apply_synchronize(self, recv_left, recv_right, laser_y, laser_z, t, dt):  # yee.rs:411-496
        G = GHOST_SIZE
        n = len(self.cell)
        if self.left_bdy == INTERNAL:
            assert len(recv_left) == 2 * G
            for k in range(G):  # cell[0..4] <- recv_left[0..4], overlay_ghost
                self.cell[k].overlay_ghost(recv_left[k])
            for k in range(G):  # cell[4..8] <- recv_left[4..8], overlay
                self.cell[G + k].overlay(recv_left[G + k])
        if self.right_bdy == INTERNAL:
            assert len(recv_right) == 2 * G
            for k in range(G):  # cell[-8..-4] <- recv_right[0..4], overlay
                self.cell[n - 2 * G + k].overlay(recv_right[k])
            for k in range(G):  # cell[-4..] <- recv_right[4..8], overlay_ghost
                self.cell[n - G + k].overlay_ghost(recv_right[G + k])
        if self.left_bdy == LASER:  # yee.rs:456-462
            hit("synchronize: laser")
            start = self.left_bdy_size - 2
            x = self.cell[start].x
            r = SPEED_OF_LIGHT * dt / self.dx
            self.cell[start].E[1] += 2.0 * r * laser_y(t, x)
            self.cell[start].E[2] += 2.0 * r * laser_z(t, x)
        if self.right_bdy == ABSORBING:  # yee.rs:464-479
            hit("synchronize: absorbing")
            xmin = self.cell[self.size + self.left_bdy_size].x
            xmax = self.cell[self.size + self.left_bdy_size + self.right_bdy_size - 1].x
            sigma_max = 10.0 / float(self.right_bdy_size)
            for c in self.cell[n - self.right_bdy_size + 1:]:
                sigma = sigma_max * (c.x - xmin) / (xmax - xmin)
                c.E[0] = (1.0 - sigma) * c.E[0]
                c.E[1] = (1.0 - sigma) * c.E[1]
                c.E[2] = (1.0 - sigma) * c.E[2]
                c.B[0] = (1.0 - sigma) * c.B[0]
                c.B[1] = (1.0 - sigma) * c.B[1]
                c.B[2] = (1.0 - sigma) * c.B[2]
            for c in self.cell[n - 2:]:
                c.E = [0.0, 0.0, 0.0]
                c.B = [0.0, 0.0, 0.0]
        elif self.right_bdy == CONDUCTING:  # yee.rs:480-495
            hit("synchronize: conducting")
            start = n - self.right_bdy_size
            self.cell[start].E[0] = 0.0
            self.cell[start].B[1] = 0.0
            self.cell[start].B[2] = 0.0
            for i in range(1, self.right_bdy_size):
                self.cell[start + i].E[0] = -self.cell[start - i].E[0]
                self.cell[start + i].E[1] = self.cell[start + 1 - i].E[1]
                self.cell[start + i].E[2] = self.cell[start + 1 - i].E[2]
                self.cell[start + i].B[0] = self.cell[start + 1 - i].B[0]
                self.cell[start + i].B[1] = -self.cell[start - i].B[1]
                self.cell[start + i].B[2] = -self.cell[start - i].B[2]

    def fields_at(self, c, xi):  # yee.rs:499-529
        j = c + self.left_bdy_size
        assert j >= self.left_bdy_size
        assert j < len(self.cell) - self.right_bdy_size
        cl = self.cell
        E = [
            cl[j - 1].E[0] * weight(0.5 + xi) + cl[j].E[0] * weight(0.5 - xi) + cl[j + 1].E[0] * weight(1.5 - xi),
            cl[j - 1].E[1] * weight(1.0 + xi) + cl[j].E[1] * weight(xi) + cl[j + 1].E[1] * weight(1.0 - xi)
            + cl[j + 2].E[1] * weight(2.0 - xi),
            cl[j - 1].E[2] * weight(1.0 + xi) + cl[j].E[2] * weight(xi) + cl[j + 1].E[2] * weight(1.0 - xi)
            + cl[j + 2].E[2] * weight(2.0 - xi),
        ]
        B = [
            cl[j].B[0],
            cl[j - 1].B[1] * weight(0.5 + xi) + cl[j].B[1] * weight(0.5 - xi) + cl[j + 1].B[1] * weight(1.5 - xi),
            cl[j - 1].B[2] * weight(0.5 + xi) + cl[j].B[2] * weight(0.5 - xi) + cl[j + 1].B[2] * weight(1.5 - xi),
        ]
        return E, B

    def clear(self):  # yee.rs:543-549
        for c in self.cell:
            c.rho = 0.0
            c.j = [0.0, 0.0, 0.0]

    def deposit(self, pt, dt):  # yee.rs:551-641, one rayon thread: one chunk holding every particle
        dx = self.dx
        size = len(self.cell)
        if not pt:
            return
        j = [[0.0, 0.0, 0.0, 0.0] for _ in range(size)]
        for p in pt:
            c, x, prev_x = p.location()
            macrocharge = p.weight * p.charge()
            velocity = p.velocity()
            index = c + self.left_bdy_size
            assert index >= 2
            assert index < len(j) - 2
            j[index][0] += macrocharge * flux(0.5 - prev_x, 0.5 - x) / dt
            j[index - 1][0] += macrocharge * flux(-0.5 - prev_x, -0.5 - x) / dt
            j[index - 2][0] += macrocharge * flux(-1.5 - prev_x, -1.5 - x) / dt
            j[index + 1][0] += macrocharge * flux(1.5 - prev_x, 1.5 - x) / dt
            j[index + 2][0] += macrocharge * flux(2.5 - prev_x, 2.5 - x) / dt

            j[index - 1][1] += macrocharge * velocity[1] * weight(1.0 + x) / dx
            j[index][1] += macrocharge * velocity[1] * weight(x) / dx
            j[index + 1][1] += macrocharge * velocity[1] * weight(1.0 - x) / dx
            j[index + 2][1] += macrocharge * velocity[1] * weight(2.0 + x) / dx  # sic, yee.rs:597: identically zero

            j[index - 1][2] += macrocharge * velocity[2] * weight(1.0 + x) / dx
            j[index][2] += macrocharge * velocity[2] * weight(x) / dx
            j[index + 1][2] += macrocharge * velocity[2] * weight(1.0 - x) / dx
            j[index + 2][2] += macrocharge * velocity[2] * weight(2.0 + x) / dx  # sic, yee.rs:602

            j[index - 1][3] += macrocharge * weight(1.0 + x) / dx
            j[index][3] += macrocharge * weight(x) / dx
            j[index + 1][3] += macrocharge * weight(1.0 - x) / dx
            j[index - 2][3] += macrocharge * weight(2.0 - x) / dx  # sic, yee.rs:609: index - 2
        total = [[0.0, 0.0, 0.0, 0.0] for _ in range(size)]
        for i in range(size):  # yee.rs:621-628 with j.len() == 1
            total[i][0] += j[i][0]
            total[i][1] += j[i][1]
            total[i][2] += j[i][2]
            total[i][3] += j[i][3]
        for c, jj in zip(self.cell, total):  # yee.rs:635-640
            c.j[0] += jj[0]
            c.j[1] += jj[1]
            c.j[2] += jj[2]
            c.rho += jj[3]

    def em_field_energy_local(self):  # yee.rs:787-798 (the reduce over ranks is the caller's)
        e2, b2 = 0.0, 0.0
        for c in self.cell[self.left_bdy_size:len(self.cell) - self.right_bdy_size]:
            e2 = e2 + powi2(c.E[0]) + powi2(c.E[1]) + powi2(c.E[2])
            b2 = b2 + powi2(c.B[0]) + powi2(c.B[1]) + powi2(c.B[2])
        return 0.5 * (VACUUM_PERMITTIVITY * e2 + b2 / VACUUM_PERMEABILITY) * self.dx

    def interpolate(self):  # yee.rs:815-835
        v = self.cell[self.left_bdy_size - 1:len(self.cell) - self.right_bdy_size]
        out = []
        for a, b in zip(v[:-1], v[1:]):
            c = Cell(b.x)
            c.rho = b.rho
            c.j = [0.5 * (a.j[0] + b.j[0]), b.j[1], b.j[2]]
            c.E = [0.5 * (a.E[0] + b.E[0]), b.E[1], b.E[2]]
            c.B = [b.B[0], 0.5 * (a.B[1] + b.B[1]), 0.5 * (a.B[2] + b.B[2])]
            out.append(c)
        return out

    def advance_B(self, dt):  # yee.rs:839-848
        cl = self.cell
        for i in range(0, len(cl) - 1):
            cl[i].B[1] += dt * (cl[i + 1].E[2] - cl[i].E[2]) / self.dx
            cl[i].B[2] += dt * (cl[i].E[1] - cl[i + 1].E[1]) / self.dx

    def advance_E(self, dt):  # yee.rs:852-867
        cl = self.cell
        kappa = 2.0 * SPEED_OF_LIGHT * dt / (SPEED_OF_LIGHT * dt + self.dx)
        sigma = 1.0 - kappa
        cl[0].E[0] = 0.0
        cl[0].E[1] = sigma * cl[0].E[1] - SPEED_OF_LIGHT * kappa * cl[0].B[2]
        cl[0].E[2] = sigma * cl[0].E[2] + SPEED_OF_LIGHT * kappa * cl[0].B[1]
        for i in range(1, len(cl)):
            cl[i].E[0] += -dt * cl[i].j[0] / VACUUM_PERMITTIVITY
            cl[i].E[1] += dt * SPEED_OF_LIGHT_SQD * (cl[i - 1].B[2] - cl[i].B[2]) / self.dx - dt * cl[i].j[1] / VACUUM_PERMITTIVITY
            cl[i].E[2] += dt * SPEED_OF_LIGHT_SQD * (cl[i].B[1] - cl[i - 1].B[1]) / self.dx - dt * cl[i].j[2] / VACUUM_PERMITTIVITY


def world_synchronize(grids, laser_y, laser_z, t, dt):
    """yee.rs:351-409 for every rank of a world at once: each rank's send buffers are copies taken on entry; what a rank
    receives "from the right" is its right neighbour's send_left (its own when it is its own neighbour) and vice versa.
    The even/odd ordering only sequences the blocking calls."""
    sends = [g.send_buffers() for g in grids]
    for g in grids:
        recv_left = sends[g.rank_to_left][1] if g.left_bdy == INTERNAL else []
        recv_right = sends[g.rank_to_right][0] if g.right_bdy == INTERNAL else []
        g.apply_synchronize(recv_left, recv_right, laser_y, laser_z, t, dt)


# ---------------------------------------------------------------------------------------------------
# particle/electron.rs, particle/ion.rs
# ---------------------------------------------------------------------------------------------------
def step_cell(cell, x, prev_x):
    """electron.rs:318-329 == ion.rs:204-214"""
    floor = float(math.floor(x))
    if floor < 0.0:
        hit("push: floor < 0 (cell - 1)")
        cell = cell - 1
    elif floor > 0.0:
        hit("push: floor > 0 (cell + 1)")
        cell = cell + 1
    else:
        hit("push: floor == 0 (same cell)")
    prev_x -= floor
    x -= floor
    return cell, x, prev_x


class Electron:  # electron.rs:14-27
    species = "electron"

    def __init__(self, cell, x, u, weight_, dx, dt):  # create, electron.rs:85-103
        u = (u[0], u[1], u[2])
        gamma = math.sqrt(1.0 + vdot(u, u))
        self.cell = cell
        self.prev_x = x - SPEED_OF_LIGHT * u[0] * dt / (gamma * dx)
        self.x = x
        self.y = 0.0
        self.z = 0.0
        self.weight = weight_
        self.gamma = gamma
        self.u = u
        self.chi = 0.0
        self.work = 0.0

    def location(self):  # electron.rs:105-107
        return self.cell, self.x, self.prev_x

    def charge(self):  # electron.rs:140-142
        return ELECTRON_CHARGE

    def velocity(self):  # electron.rs:148-151
        return vdiv(smul(SPEED_OF_LIGHT, self.u), self.gamma)

    def total_kinetic_energy(self):  # electron.rs:122-126
        to_joules = 1.0e6 * ELECTRON_MASS_MEV * ELEMENTARY_CHARGE
        return self.weight * norm_sqr(self.u) * to_joules / (self.gamma + 1.0)

    def shift_cell(self, delta):
        self.cell += delta

    def push(self, E, B, dx, dt):  # vay_push, electron.rs:268-330 (tau is not carried: unobservable with emission off)
        E = (E[0], E[1], E[2])
        B = (B[0], B[1], B[2])
        v = vdiv(smul(SPEED_OF_LIGHT, self.u), self.gamma)
        alpha = ELECTRON_CHARGE * dt / (2.0 * ELECTRON_MASS * SPEED_OF_LIGHT)
        u_half = vadd(self.u, smul(alpha, vadd(E, vcross(v, B))))
        self.work += ELECTRON_CHARGE * SPEED_OF_LIGHT * vdot(u_half, E) * dt / math.sqrt(1.0 + vdot(u_half, u_half))
        gamma_half = math.sqrt(1.0 + vdot(u_half, u_half))
        self.chi = math.sqrt(norm_sqr(vadd(smul(gamma_half, E), smul(SPEED_OF_LIGHT, vcross(u_half, B))))
                             - powi2(vdot(E, u_half))) / CRITICAL_FIELD
        u_prime = vadd(u_half, smul(alpha, E))
        gamma_prime_sqd = 1.0 + vdot(u_prime, u_prime)
        tau = smul(alpha * SPEED_OF_LIGHT, B)
        u_star = vdot(u_prime, tau)
        sigma = gamma_prime_sqd - vdot(tau, tau)
        self.gamma = math.sqrt(0.5 * sigma + math.sqrt(0.25 * powi2(sigma) + vdot(tau, tau) + powi2(u_star)))
        t = vdiv(tau, self.gamma)
        s = 1.0 / (1.0 + vdot(t, t))
        self.u = smul(s, vadd(vadd(u_prime, smul(vdot(u_prime, t), t)), vcross(u_prime, t)))
        self.prev_x = self.x
        dxi = SPEED_OF_LIGHT * self.u[0] * dt / (dx * self.gamma)
        assert dxi < 1.0
        self.x = self.x + dxi
        self.y = self.y + v[1] * dt
        self.z = self.z + v[2] * dt
        self.cell, self.x, self.prev_x = step_cell(self.cell, self.x, self.prev_x)


class Ion:  # ion.rs:14-26
    species = "ion"

    def __init__(self, cell, x, u, weight_, dx, dt):  # create, ion.rs:81-101
        u = (u[0], u[1], u[2])
        gamma_m1 = norm_sqr(u) / (1.0 + math.sqrt(1.0 + norm_sqr(u)))
        v = vdiv(smul(SPEED_OF_LIGHT, u), 1.0 + gamma_m1)
        self.cell = cell
        self.prev_x = x - v[0] * dt / dx
        self.x = x
        self.weight = weight_
        self.Z_star, self.Z, self.A = 1.0, 1.0, 1.0
        self.gamma_m1 = gamma_m1
        self.u = u

    def with_charge_state(self, charge_state, atomic_number, mass_number):  # ion.rs:236-241
        self.Z_star, self.Z, self.A = charge_state, atomic_number, mass_number
        return self

    def location(self):
        return self.cell, self.x, self.prev_x

    def charge(self):  # ion.rs:111-113
        return self.Z_star * ELEMENTARY_CHARGE

    def velocity(self):  # ion.rs:119-122
        return vdiv(smul(SPEED_OF_LIGHT, self.u), 1.0 + self.gamma_m1)

    def energy(self):  # ion.rs:124-126
        return self.gamma_m1 * (self.A * PROTON_MASS / ELECTRON_MASS) * ELECTRON_MASS_MEV

    def total_kinetic_energy(self):  # ion.rs:128-130
        return self.weight * self.energy() * 1.0e6 * ELEMENTARY_CHARGE

    def shift_cell(self, delta):  # ion.rs:107-109
        self.cell += delta

    def push(self, E, B, dx, dt):  # ion.rs:168-214
        E = (E[0], E[1], E[2])
        cB = smul(SPEED_OF_LIGHT, (B[0], B[1], B[2]))
        q = self.Z_star * ELEMENTARY_CHARGE
        M = self.A * PROTON_MASS
        alpha = q * dt / (2.0 * M * SPEED_OF_LIGHT)
        u_minus = vadd(self.u, smul(alpha, E))
        gamma = 1.0 + norm_sqr(u_minus) / (1.0 + math.sqrt(1.0 + norm_sqr(u_minus)))
        t = alpha / gamma
        u_prime = vadd(u_minus, smul(t, vcross(u_minus, cB)))
        t_prime = 2.0 * t / (1.0 + powi2(t) * norm_sqr(cB))
        u_plus = vadd(u_minus, smul(t_prime, vcross(u_prime, cB)))
        self.u = vadd(u_plus, smul(alpha, E))
        self.gamma_m1 = norm_sqr(self.u) / (1.0 + math.sqrt(1.0 + norm_sqr(self.u)))
        self.prev_x = self.x
        dxi = SPEED_OF_LIGHT * self.u[0] * dt / (dx * (1.0 + self.gamma_m1))
        assert dxi < 1.0
        self.x = self.x + dxi
        self.cell, self.x, self.prev_x = step_cell(self.cell, self.x, self.prev_x)


class Photon:  # photon.rs:16-32 (polarisation members and optical depths: not on this path)
    species = "photon"

    def __init__(self, cell, x, u, weight_, dx, dt):  # create, photon.rs:95-119
        k = (u[0], u[1], u[2])
        k0 = math.sqrt(norm_sqr(k))
        self.cell = cell
        self.prev_x = x - SPEED_OF_LIGHT * k[0] * dt / (k0 * dx)
        self.x = x
        self.y = 0.0
        self.z = 0.0
        self.weight = weight_
        self.k = k
        self.u = k  # (what the dumps call the momentum)
        self.chi = 0.0

    def location(self):  # photon.rs:185-187
        return self.cell, self.x, self.prev_x

    def charge(self):  # photon.rs:197-199
        return 0.0

    def velocity(self):  # photon.rs:213-217
        k0 = math.sqrt(norm_sqr(self.k))
        return vdiv(smul(SPEED_OF_LIGHT, self.k), k0)

    def energy(self):  # photon.rs:219-221
        return math.sqrt(norm_sqr(self.k)) * ELECTRON_MASS_MEV

    def total_kinetic_energy(self):  # photon.rs:223-225
        return self.weight * self.energy() * 1.0e6 * ELEMENTARY_CHARGE

    def shift_cell(self, delta):  # photon.rs:193-195
        self.cell = self.cell + delta

    def push(self, E, B, dx, dt):  # photon.rs:150-183
        E = (E[0], E[1], E[2])
        B = (B[0], B[1], B[2])
        k0 = math.sqrt(norm_sqr(self.k))
        v = vdiv(smul(SPEED_OF_LIGHT, self.k), k0)
        self.chi = math.sqrt(norm_sqr(vadd(smul(k0, E), smul(SPEED_OF_LIGHT, vcross(self.k, B))))
                             - powi2(vdot(E, self.k))) / CRITICAL_FIELD
        self.prev_x = self.x
        dxi = v[0] * dt / dx
        assert dxi < 1.0
        self.x = self.x + dxi
        self.y = self.y + v[1] * dt
        self.z = self.z + v[2] * dt
        self.cell, self.x, self.prev_x = step_cell(self.cell, self.x, self.prev_x)


# ---------------------------------------------------------------------------------------------------
# particle/mod.rs:247-376 — Population::advance over a whole world
# ---------------------------------------------------------------------------------------------------
def world_pop_advance(stores, grids, dt):
    """Every rank pushes (mod.rs:263-282), sorts by cell (mod.rs:288: unstable in the reference — the order inside a
    cell is not defined there; a STABLE sort here, stated in the file), re-bases its right-leavers by its own size
    (mod.rs:291-293), trades with its neighbours, re-bases the arrivals from the right by ITS size (mod.rs:366-368),
    splices them over the leavers (mod.rs:372-373).  A missing neighbour (to_left/to_right None) deletes."""
    sends = []
    for store, grid in zip(stores, grids):
        mx = grid.size
        gone_left = gone_right = 0
        for pt in store:
            c, x, _ = pt.location()
            E, B = grid.fields_at(c, x)
            pt.push(E, B, grid.dx, dt)
            c = pt.location()[0]
            if c < 0:
                gone_left += 1
            elif c >= mx:
                gone_right += 1
        store.sort(key=lambda p: p.cell)
        num = len(store)
        for pt in store[num - gone_right:num]:
            pt.shift_cell(-mx)
        sends.append((list(store[0:gone_left]), list(store[num - gone_right:num]), gone_left, gone_right))
    import copy
    for r, (store, grid) in enumerate(zip(stores, grids)):
        mx = grid.size
        send_left, send_right, gone_left, gone_right = sends[r]
        recv_left, recv_right = [], []
        rr = grid.to_right()
        if rr is not None:
            hit("migration: to the right neighbour")
            src = sends[rr][0] if rr != grid.id else send_left
            recv_right = [copy.copy(p) for p in src]
        else:
            if gone_right:
                hit("migration: right-leavers deleted at an open edge")
        ll = grid.to_left()
        if ll is not None:
            hit("migration: to the left neighbour")
            src = sends[ll][1] if ll != grid.id else send_right
            recv_left = [copy.copy(p) for p in src]
        else:
            if gone_left:
                hit("migration: left-leavers deleted at an open edge")
        for pt in recv_right:
            pt.shift_cell(mx)
        num = len(store)
        store[num - gone_right:num] = recv_right
        store[0:gone_left] = recv_left


def world_step(grids, electrons, ions, current_deposition, laser_y, laser_z, t, dt):
    """main.rs:238-267 (photons, absorb, emit_radiation: off)"""
    world_pop_advance(electrons, grids, dt)
    world_pop_advance(ions, grids, dt)
    if current_deposition:
        for g, e, i in zip(grids, electrons, ions):
            g.clear()
            g.deposit(e, dt)
            g.deposit(i, dt)
    world_synchronize(grids, laser_y, laser_z, t, dt)
    for g in grids:
        g.advance(dt)
    return t + dt


# ---------------------------------------------------------------------------------------------------
# serialisation: every float as a C99 hex literal
# ---------------------------------------------------------------------------------------------------
def hx(v):
    if isinstance(v, (list, tuple)):
        return [hx(a) for a in v]
    return float(v).hex()


COMPS = ("rho", "jx", "jy", "jz", "Ex", "Ey", "Ez", "Bx", "By", "Bz")


def get_comp(cell, name):
    if name == "rho":
        return cell.rho
    k = "xyz".index(name[1])
    return {"j": cell.j, "E": cell.E, "B": cell.B}[name[0]][k]


def set_comp(cell, name, v):
    if name == "rho":
        cell.rho = v
    else:
        k = "xyz".index(name[1])
        {"j": cell.j, "E": cell.E, "B": cell.B}[name[0]][k] = v


def grid_dump(g, comps=COMPS):
    return {c: hx([get_comp(cl, c) for cl in g.cell]) for c in comps}


def design_dump(d):
    return {"numtasks": d.numtasks, "left": d.left, "right": d.right, "xmin": hx(d.xmin), "dx": hx(d.dx),
            "nx": list(d.nx), "offset": hx(d.offset)}


def randomise(g, rng, comps=COMPS, amp=None):
    amp = amp or {"E": 1e3, "B": 1e-5, "j": 1e2, "r": 1e-9}
    for cl in g.cell:
        for c in comps:
            set_comp(cl, c, amp[c[0]] * rng.gauss(0.0, 1.0))


def pt_dump(p):
    d = {"cell": p.cell, "x": hx(p.x), "prev_x": hx(p.prev_x), "u": hx(p.u), "w": hx(p.weight)}
    if isinstance(p, Electron):
        d.update(gamma=hx(p.gamma), chi=hx(p.chi), work=hx(p.work), y=hx(p.y), z=hx(p.z))
    else:
        d.update(gamma_m1=hx(p.gamma_m1))
    d["velocity"] = hx(p.velocity())
    d["kinetic_energy"] = hx(p.total_kinetic_energy())
    return d


# ---------------------------------------------------------------------------------------------------
# the vectors
# ---------------------------------------------------------------------------------------------------
def vectors_weight_flux():
    ulp = 2.0 ** -52
    args = [-2.5, -1.5 - ulp, -1.5, -1.5 + ulp, -1.2, -1.0, -0.75, -0.5 - ulp, -0.5, -0.5 + ulp / 2, -0.25, -0.0, 0.0,
            1e-300, 0.1, 0.25, 0.5 - ulp / 2, 0.5, 0.5 + ulp, 0.6, 0.99, 1.0, 1.3, 1.5 - ulp, 1.5, 1.5 + 2 * ulp, 1.6, 2.0,
            2.999, 1e6]
    w = [[hx(a), hx(weight(a))] for a in args]
    fl = []
    xs = [-1.7, -1.3, -1.0, -1.0 + ulp / 2, -0.9, -0.4, -0.1, -0.0, 0.0, 0.1, 0.45, 0.9, 1.0 - ulp / 2, 1.0, 1.2, 2.5]
    ds = [-0.95, -0.9, -0.5, -0.05, -1e-17, 0.0, 1e-17, 0.05, 0.5, 0.9, 0.95]
    for xi in xs:
        for d in ds:
            xf = xi - d
            fl.append([hx(xi), hx(xf), hx(flux(xi, xf))])
    # the arguments deposit() really forms: k + 1/2 - prev_x, k + 1/2 - x for a particle that moved
    rng = random.Random(11)
    for _ in range(60):
        prev_x = rng.random()
        x = prev_x + rng.uniform(-0.95, 0.95)
        fl_ = float(math.floor(x))
        x, prev_x = x - fl_, prev_x - fl_
        for k in (0.5, -0.5, -1.5, 1.5, 2.5):
            fl.append([hx(k - prev_x), hx(k - x), hx(flux(k - prev_x, k - x))])
    return w, fl


def vectors_fields_at(rng):
    out = []
    dx = 1.0e-8
    for name, d in (
        ("periodic", design_unbalanced(0, 1, 24, -1.0e-7, dx, 8, INTERNAL, INTERNAL)),
        ("laser_absorbing", design_unbalanced(0, 1, 24, 2.0e-7, dx, 8, LASER, ABSORBING)),
        ("middle_of_three", design_unbalanced(1, 3, 48, 0.0, dx, 8, LASER, CONDUCTING)),
    ):
        g = YeeGrid(d)
        randomise(g, rng, comps=("Ex", "Ey", "Ez", "Bx", "By", "Bz"))
        q = []
        ulp = 2.0 ** -53
        for c in (0, 1, g.size // 2, g.size - 2, g.size - 1):
            for xi in (0.0, ulp, 0.25, 0.5 - ulp, 0.5, 0.5 + 2 * ulp, 0.75, 1.0 - ulp, rng.random(), rng.random()):
                E, B = g.fields_at(c, xi)
                q.append({"c": c, "xi": hx(xi), "E": hx(E), "B": hx(B)})
        out.append({"name": name, "id": d.id, "design": design_dump(d),
                    "fields": grid_dump(g, ("Ex", "Ey", "Ez", "Bx", "By", "Bz")), "queries": q})
    return out


def vectors_pushes(rng):
    dx = 1.0e-8
    dt = 0.95 * dx / SPEED_OF_LIGHT
    ep, ip = [], []

    def fields(kind):
        E = [1e12 * rng.gauss(0, 1) for _ in range(3)]
        B = [3e3 * rng.gauss(0, 1) for _ in range(3)]
        if kind == "E only":
            B = [0.0, 0.0, 0.0]
        if kind == "B only":
            E = [0.0, 0.0, 0.0]
        if kind == "none":
            E, B = [0.0] * 3, [0.0] * 3
        if kind == "weak":
            E, B = [1e3 * e / 1e12 for e in E], [1e-5 * b / 3e3 for b in B]
        return E, B

    cases = []
    for kind in ("both", "E only", "B only", "none", "weak"):
        for scale in (0.0, 1e-3, 0.3, 3.0, 80.0):
            cases.append((kind, scale))
    for kind, scale in cases:
        u = [scale * rng.gauss(0, 1) for _ in range(3)]
        x = rng.random() if scale else rng.choice([0.0, 0.5, 1.0 - 2.0 ** -53])
        E, B = fields(kind)
        e = Electron(5, x, u, 2.5e3, dx, dt)
        created = pt_dump(e)
        steps = []
        for _ in range(3):  # three pushes in the same field: gamma, work, y, z carry over
            e.push(E, B, dx, dt)
            steps.append(pt_dump(e))
        ep.append({"kind": kind, "cell": 5, "x": hx(x), "u": hx(u), "w": hx(2.5e3), "dx": hx(dx), "dt": hx(dt),
                   "E": hx(E), "B": hx(B), "created": created, "pushes": steps})
        for zs, z, a in ((1.0, 1.0, 1.0), (6.0, 6.0, 12.0), (1.0, 1.0, ELECTRON_MASS / PROTON_MASS), (3.0, 13.0, 26.98)):
            light = a < 1e-2
            uu = [c * (1.0 if light else 1e-2) for c in u]
            p = Ion(5, x, uu, 4.0e2, dx, dt).with_charge_state(zs, z, a)
            created = pt_dump(p)
            steps = []
            for _ in range(3):
                p.push(E, B, dx, dt)
                steps.append(pt_dump(p))
            ip.append({"kind": kind, "cell": 5, "x": hx(x), "u": hx(uu), "w": hx(4.0e2), "dx": hx(dx), "dt": hx(dt),
                       "Z_star": hx(zs), "Z": hx(z), "A": hx(a), "E": hx(E), "B": hx(B), "created": created,
                       "pushes": steps})
    return ep, ip


def make_particles(rng, grid, n, cls, dt, u_scale, w, cells=None, charge_state=None):
    out = []
    for _ in range(n):
        c = rng.randrange(grid.size) if cells is None else rng.choice(cells)
        x = rng.random()
        u = [u_scale * rng.gauss(0, 1) for _ in range(3)]
        p = cls(c, x, u, w * (0.5 + rng.random()), grid.dx, dt)
        if charge_state:
            p.with_charge_state(*charge_state)
        out.append(p)
    return out


def created_dump(pts):
    """What Particle::create was called with, per particle (cell, x, u, weight)."""
    return {"cell": [p.cell for p in pts], "x": hx([p.x for p in pts]), "ux": hx([p.u[0] for p in pts]),
            "uy": hx([p.u[1] for p in pts]), "uz": hx([p.u[2] for p in pts]), "w": hx([p.weight for p in pts])}


def state_dump(pts):
    d = {"cell": [p.cell for p in pts], "x": hx([p.x for p in pts]), "prev_x": hx([p.prev_x for p in pts]),
         "ux": hx([p.u[0] for p in pts]), "uy": hx([p.u[1] for p in pts]), "uz": hx([p.u[2] for p in pts]),
         "w": hx([p.weight for p in pts])}
    if pts and isinstance(pts[0], Electron):
        d.update(chi=hx([p.chi for p in pts]), work=hx([p.work for p in pts]), y=hx([p.y for p in pts]),
                 z=hx([p.z for p in pts]), gamma=hx([p.gamma for p in pts]))
    if pts and isinstance(pts[0], Photon):
        d.update(chi=hx([p.chi for p in pts]), y=hx([p.y for p in pts]), z=hx([p.z for p in pts]))
    return d


def vectors_photons(rng):
    """Photon::create + push (photon.rs:95-119, 150-183) through Population::advance on one periodic subdomain of random
    fields: straight lines at c (a cell crossing almost every step), chi from the gathered fields, y and z, the kinetic
    energy (photon.rs:219-225); three steps, the state after each."""
    out = []
    dx = 1.0e-8
    dt = 0.95 * dx / SPEED_OF_LIGHT
    for label, kscale in (("MeV photons", 5.0), ("soft photons, all directions", 1.0e-3)):
        d = design_unbalanced(0, 1, 20, 0.0, dx, 8, INTERNAL, INTERNAL)
        g = YeeGrid(d)
        randomise(g, rng, comps=("Ex", "Ey", "Ez", "Bx", "By", "Bz"), amp={"E": 3e12, "B": 1e4})
        world_synchronize([g], lambda a, b: 0.0, lambda a, b: 0.0, 0.0, dt)
        ph = []
        for _ in range(40):
            k = [kscale * rng.gauss(0, 1) for _ in range(3)]
            ph.append(Photon(rng.randrange(g.size), rng.random(), k, 7.0e5 * (0.5 + rng.random()), dx, dt))
        ph.append(Photon(3, 0.5, [kscale, 0.0, 0.0], 7.0e5, dx, dt))     # along +x: |dxi| = 0.95
        ph.append(Photon(3, 0.5, [-kscale, 0.0, 0.0], 7.0e5, dx, dt))
        ph.append(Photon(3, 0.5, [0.0, kscale, 0.0], 7.0e5, dx, dt))     # never leaves its cell
        rec = {"name": label, "design": design_dump(d), "dt": hx(dt),
               "fields": grid_dump(g, ("Ex", "Ey", "Ez", "Bx", "By", "Bz")), "created": created_dump(ph),
               "created_prev_x": hx([p.prev_x for p in ph]), "steps": []}
        for _ in range(3):
            world_pop_advance([ph], [g], dt)
            st = state_dump(ph)
            st["kinetic_energy_sequential_sum"] = hx(sum_seq([p.total_kinetic_energy() for p in ph]))
            rec["steps"].append(st)
        out.append(rec)
    return out


def vectors_deposit(rng):
    """clear + deposit(electrons) + deposit(ions) on one periodic subdomain: (A) straight after create — the first
    deposit of a run sees create()'s prev_x (electron.rs:88, ion.rs:87); (B) after one advance in random fields, the
    edge cells included, so that wrapped particles and the ghost-cell taps are in."""
    out = []
    dx = 1.0e-8
    dt = 0.95 * dx / SPEED_OF_LIGHT
    for label, nx, ne_, ni_, uscale in (("sparse", 16, 7, 5, 0.9), ("edges", 12, 40, 30, 2.0), ("cold", 10, 12, 0, 0.0)):
        d = design_unbalanced(0, 1, nx, 0.0, dx, 8, INTERNAL, INTERNAL)
        g = YeeGrid(d)
        randomise(g, rng, comps=("Ex", "Ey", "Ez", "Bx", "By", "Bz"), amp={"E": 3e11, "B": 1e3})
        # a state the loop can reach: the push always follows a synchronize, which makes the ghost cells copies of
        # the cells they stand for (an implementation is free to read either image)
        world_synchronize([g], lambda a, b: 0.0, lambda a, b: 0.0, 0.0, dt)
        cells = None if label != "edges" else [0, 0, 1, nx - 2, nx - 1, nx - 1, nx // 2]
        el = make_particles(rng, g, ne_, Electron, dt, uscale, 1.0e6, cells)
        io = make_particles(rng, g, ni_, Ion, dt, uscale * 1e-2, 3.0e5, cells, (6.0, 6.0, 12.0))
        rec = {"name": label, "design": design_dump(d), "dt": hx(dt), "charge_state": hx([6.0, 6.0, 12.0]),
               "fields": grid_dump(g, ("Ex", "Ey", "Ez", "Bx", "By", "Bz")),
               "electrons_created": created_dump(el), "ions_created": created_dump(io)}
        g.clear()
        g.deposit(el, dt)
        g.deposit(io, dt)
        rec["after_create"] = grid_dump(g, ("rho", "jx", "jy", "jz"))
        world_pop_advance([el], [g], dt)
        world_pop_advance([io], [g], dt)
        rec["electrons_advanced"] = state_dump(el)
        rec["ions_advanced"] = state_dump(io)
        g.clear()
        g.deposit(el, dt)
        g.deposit(io, dt)
        rec["after_advance"] = grid_dump(g, ("rho", "jx", "jy", "jz"))
        rec["kinetic_energy_sequential_sum"] = {
            "electrons": hx(sum_seq([p.total_kinetic_energy() for p in el])),
            "ions": hx(sum_seq([p.total_kinetic_energy() for p in io]))}
        out.append(rec)
    return out


def sum_seq(v):
    s = 0.0
    for a in v:
        s += a
    return s


def vectors_sync_advance(rng):
    """synchronize (yee.rs:351-496) then advance (yee.rs:345-349, 839-867) on worlds of random cells, every boundary
    kind, equal and unequal subdomains; the state after each of the two operators, the energy and the interpolated
    rows (yee.rs:787-835)."""
    out = []
    dx = 1.0e-8
    dt = 0.95 * dx / SPEED_OF_LIGHT
    worlds = [
        ("one periodic subdomain", 1, 24, INTERNAL, INTERNAL, None),
        ("three periodic subdomains", 3, 30, INTERNAL, INTERNAL, None),
        ("laser + absorbing, one subdomain", 1, 16, LASER, ABSORBING, None),
        ("laser + absorbing, two unequal subdomains", 2, 40, LASER, ABSORBING, [27, 13]),
        ("laser + conducting, one subdomain", 1, 20, LASER, CONDUCTING, None),
        ("laser + conducting, two subdomains", 2, 32, LASER, CONDUCTING, None),
        ("smallest subdomains (8 cells), four periodic", 4, 32, INTERNAL, INTERNAL, None),
    ]
    for name, ntasks, size, left, right, nx in worlds:
        designs = []
        for r in range(ntasks):
            d = design_unbalanced(r, ntasks, size, -3.0e-8, dx, 8, left, right)
            if nx:
                d.nx = list(nx)
                d.offset = _offsets(nx, dx)
            designs.append(d)
        grids = [YeeGrid(d) for d in designs]
        for g in grids:
            randomise(g, rng)
        t = 1.25e-15
        ly_v, lz_v = 4.0e9 * rng.gauss(0, 1), 4.0e9 * rng.gauss(0, 1)
        calls = []

        def ly(t_, x_):
            calls.append(["y", hx(t_), hx(x_)])
            return ly_v

        def lz(t_, x_):
            calls.append(["z", hx(t_), hx(x_)])
            return lz_v
        rec = {"name": name, "design": design_dump(designs[0]), "t": hx(t), "dt": hx(dt),
               "laser_y_value": hx(ly_v), "laser_z_value": hx(lz_v),
               "before": [grid_dump(g) for g in grids]}
        world_synchronize(grids, ly, lz, t, dt)
        rec["laser_calls"] = calls
        rec["after_synchronize"] = [grid_dump(g) for g in grids]
        for g in grids:
            g.advance(dt)
        rec["after_advance"] = [grid_dump(g) for g in grids]
        rec["em_field_energy_local"] = [hx(g.em_field_energy_local()) for g in grids]
        rec["interpolate"] = [[hx([c.x, c.rho] + c.j + c.E + c.B) for c in g.interpolate()] for g in grids]
        rec["cell_x"] = [hx([c.x for c in g.cell]) for g in grids]
        out.append(rec)
    return out


DENSITIES = {
    # name -> ne(x); simple closed forms (no transcendental functions) so that the test can rebuild them exactly
    "uniform": lambda p: (lambda x: p[0]),
    "slab": lambda p: (lambda x: p[0] if (x >= p[1] and x < p[2]) else 0.0),
    "ramp": lambda p: (lambda x: p[0] * (x - p[1]) / (p[2] - p[1]) if (x >= p[1] and x < p[2]) else 0.0),
    "two_slabs": lambda p: (lambda x: p[0] if (x >= p[1] and x < p[2]) else (p[3] if (x >= p[4] and x < p[5]) else 0.0)),
}


def vectors_design():
    un, ba = [], []
    for numtasks, size, dx, mins in ((1, 1000, 500.0, 8), (8, 32768, 1.0e-8, 8), (3, 100, 0.25, 8), (7, 1000, 1e-9, 8),
                                     (8, 40, 2.0, 8), (5, 33, 1.5e-7, 8), (2, 17, 0.1, 8)):
        d = design_unbalanced(0, numtasks, size, 0.0, dx, mins, INTERNAL, INTERNAL)
        if size // numtasks < mins:
            hit("design: unbalanced clamps to min_subsize")
        if (size // numtasks) * numtasks != size:
            hit("design: unbalanced drops the remainder")
        un.append({"numtasks": numtasks, "size": size, "dx": hx(dx), "min_subsize": mins, "nx": list(d.nx),
                   "offset": hx(d.offset)})
    for numtasks, size, xmin, dx, name, p in (
        (1, 200, 0.0, 1e-9, "uniform", [1e27]),
        (4, 4096, 0.0, 500.0, "uniform", [20.0]),
        (8, 20000, -5.0e-6, 1.0e-9, "slab", [1.0e29, 5.0e-6, 10.0e-6]),   # hole_boring's shape: vacuum, then plasma
        (4, 20000, -5.0e-6, 1.0e-9, "slab", [1.0e29, 5.0e-6, 10.0e-6]),
        (2, 20000, -5.0e-6, 1.0e-9, "slab", [1.0e29, 5.0e-6, 10.0e-6]),
        (8, 1600, 0.0, 1.0e-8, "slab", [3.0e28, 4.0e-6, 1.5e-5]),
        (5, 900, 1.0e-6, 2.0e-8, "ramp", [4.0e27, 3.0e-6, 1.7e-5]),
        (3, 300, 0.0, 1.0, "two_slabs", [2.0, 10.0, 60.0, 5.0, 200.0, 260.0]),
        (6, 64, 0.0, 1.0, "slab", [1.0, 3.0, 9.0]),                        # every cut forced by the 8-cell minimum
    ):
        d = design_balanced(0, numtasks, size, xmin, dx, 8, LASER, ABSORBING, DENSITIES[name](p))
        ba.append({"numtasks": numtasks, "size": size, "xmin": hx(xmin), "dx": hx(dx), "min_subsize": 8,
                   "density": name, "params": hx(p), "nx": list(d.nx), "offset": hx(d.offset)})
        hit("design: balanced")
    return un, ba


def vectors_world_runs(rng):
    """The loop body of main.rs:238-267 — advance (gather, push, sort, migration), clear, deposit, synchronize, advance of
    the fields — for whole small worlds, several steps, from Particle::create state."""
    out = []
    dx = 1.0e-8
    dt = 0.95 * dx / SPEED_OF_LIGHT
    cases = [
        # name, ntasks, size, left, right, nx, steps, n_e, n_i per subdomain, u scale, deposition
        ("periodic, one subdomain", 1, 16, INTERNAL, INTERNAL, None, 6, 60, 40, 1.5, True),
        ("periodic, three unequal subdomains", 3, 40, INTERNAL, INTERNAL, [9, 19, 12], 8, 40, 30, 2.5, True),
        ("laser + absorbing, two unequal subdomains", 2, 36, LASER, ABSORBING, [23, 13], 8, 50, 40, 2.5, True),
        ("periodic, two subdomains, no deposition", 2, 24, INTERNAL, INTERNAL, None, 5, 30, 0, 1.0, False),
    ]
    for name, ntasks, size, left, right, nx, steps, n_e, n_i, us, dep in cases:
        designs = []
        for r in range(ntasks):
            d = design_unbalanced(r, ntasks, size, 0.0, dx, 8, left, right)
            if nx:
                d.nx = list(nx)
                d.offset = _offsets(nx, dx)
            designs.append(d)
        grids = [YeeGrid(d) for d in designs]
        for g in grids:
            randomise(g, rng, comps=("Ex", "Ey", "Ez", "Bx", "By", "Bz"), amp={"E": 2e11, "B": 6e2})
        world_synchronize(grids, lambda a, b: 0.0, lambda a, b: 0.0, 0.0, dt)  # ghosts = copies, as before any push
        el = [make_particles(rng, g, n_e, Electron, dt, us, 2.0e6) for g in grids]
        io = [make_particles(rng, g, n_i, Ion, dt, us * 0.05, 5.0e5, None, (1.0, 1.0, ELECTRON_MASS / PROTON_MASS)
                             if "three" in name else (6.0, 6.0, 12.0)) for g in grids]
        # a bunch at the edges so that every cut sees traffic both ways and the open ends delete something
        for g, e in zip(grids, el):
            for c, ux in ((0, -3.0), (0, -0.5), (g.size - 1, 3.0), (g.size - 1, 0.7)):
                for _ in range(3):
                    e.append(Electron(c, rng.random(), [ux, 0.2 * rng.gauss(0, 1), 0.1], 2.0e6, dx, dt))
        cs = (1.0, 1.0, ELECTRON_MASS / PROTON_MASS) if "three" in name else (6.0, 6.0, 12.0)
        rec = {"name": name, "design": design_dump(designs[0]), "dt": hx(dt), "steps": steps,
               "current_deposition": dep, "charge_state": hx(list(cs)),
               "fields": [grid_dump(g, ("Ex", "Ey", "Ez", "Bx", "By", "Bz")) for g in grids],
               "electrons_created": [created_dump(e) for e in el], "ions_created": [created_dump(i) for i in io]}
        t = -2.0e-15
        rec["t0"] = hx(t)
        laser = []

        def mk(comp):
            def f(t_, x_):
                v = 6.0e11 * math.sin(2.0e15 * t_ + (0.0 if comp == "y" else 1.0))
                laser.append([comp, hx(t_), hx(x_), hx(v)])
                return v
            return f
        ly, lz = mk("y"), mk("z")
        n0 = [(len(e), len(i)) for e, i in zip(el, io)]
        for _ in range(steps):
            t = world_step(grids, el, io, dep, ly, lz, t, dt)
        rec["laser_calls"] = laser
        rec["t_end"] = hx(t)
        rec["grids"] = [grid_dump(g) for g in grids]
        rec["electrons"] = [state_dump(e) for e in el]
        rec["ions"] = [state_dump(i) for i in io]
        rec["counts_before"] = n0
        rec["counts_after"] = [(len(e), len(i)) for e, i in zip(el, io)]
        rec["em_field_energy_local"] = [hx(g.em_field_energy_local()) for g in grids]
        out.append(rec)
    return out


# ---------------------------------------------------------------------------------------------------
# the reference's own tests, run on THIS restatement
# ---------------------------------------------------------------------------------------------------
def self_checks():
    res = {}
    pi = math.pi
    # electron.rs:401-418
    b0, u0 = 1.0, 1.0
    r_c = ELECTRON_MASS * SPEED_OF_LIGHT * u0 / (abs(ELECTRON_CHARGE) * b0)
    omega_c = SPEED_OF_LIGHT * u0 / (math.sqrt(1.0 + u0 * u0) * r_c)
    dt = 0.01 * 2.0 * pi / omega_c
    e = Electron(0, 0.0, [u0, 0.0, 0.0], 1.0, 1.0, dt)
    for _ in range(100):
        e.push([0.0, 0.0, 0.0], [0.0, 0.0, b0], 1.0, dt)
    target = Electron(0, 0.0, [1.0, 0.0, 0.0], 1.0, 1.0, dt)
    err = (e.gamma - target.gamma) / target.gamma
    assert err < 1.0e-6
    res["static_magnetic_field (electron.rs:401-418)"] = {"error": err, "bound": 1e-6}
    # electron.rs:424-436
    e0, dt = 1.0e3, 1.0e-9
    e = Electron(0, 0.0, [0.0, 0.0, 0.0], 1.0, 1.0, dt)
    for _ in range(100):
        e.push([e0, 0.0, 0.0], [0.0, 0.0, 0.0], 1.0, dt)
    target = math.sqrt(1.0 + powi2(ELECTRON_CHARGE * e0 * 1.0e-7 / (ELECTRON_MASS * SPEED_OF_LIGHT)))
    err = (e.gamma - target) / target
    assert err < 1.0e-6
    res["static_electric_field (electron.rs:424-436)"] = {"error": err, "bound": 1e-6}

    xmin, dx = -10.0e-6, 1.0e-6 / 100.0
    dt = 0.95 * dx / SPEED_OF_LIGHT
    omega = 2.0 * pi * SPEED_OF_LIGHT / 1.0e-6
    emax = 1.0e10
    # yee.rs:903-953
    t = -15.0e-6 / SPEED_OF_LIGHT
    nsteps = int(abs(t) / dt)

    def laser_y(t_, z_):
        phi = omega * (t_ - z_ / SPEED_OF_LIGHT)
        return emax * math.sin(phi) * powi2(math.cos(phi / 16.0)) if abs(phi) < 8.0 * pi else 0.0
    grid = YeeGrid(design_unbalanced(0, 1, 2000, xmin, dx, 8, LASER, ABSORBING))
    for _ in range(nsteps + 1):
        world_synchronize([grid], laser_y, lambda a, b: 0.0, t, dt)
        grid.advance(dt)
        t += dt
    energy = grid.em_field_energy_local()
    target = VACUUM_PERMITTIVITY * powi2(emax) * (3.0 * pi * SPEED_OF_LIGHT / omega)
    err = abs(energy - target) / target
    assert err < 1.0e-3
    ey1 = grid.fields_at(1025, 0.0)[0][1] / emax
    ey2 = grid.fields_at(1125, 0.0)[0][1] / emax
    res["laser_bc (yee.rs:903-953)"] = {"error": err, "bound": 1e-3, "em_energy": hx(energy),
                                        "ey_quarter_wave": ey1, "expected": -powi2(math.cos(pi / 32.0)),
                                        "ey_five_quarter": ey2, "expected5": -powi2(math.cos(5.0 * pi / 32.0))}
    # yee.rs:956-1011
    t = -15.0e-6 / SPEED_OF_LIGHT
    nsteps = int(35.0e-6 / (SPEED_OF_LIGHT * dt))

    def cy(t_, z_):
        phi = omega * (t_ - z_ / SPEED_OF_LIGHT)
        if abs(phi) < 8.0 * pi:
            return -emax * (math.sin(phi) + (2.0 / 16.0) * math.cos(phi) * math.tan(phi / 16.0)) * powi2(math.cos(phi / 16.0))
        return 0.0

    def cz(t_, z_):
        phi = omega * (t_ - z_ / SPEED_OF_LIGHT)
        if abs(phi) < 8.0 * pi:
            return emax * (math.cos(phi) + (2.0 / 16.0) * math.sin(phi) * math.tan(phi / 16.0)) * powi2(math.cos(phi / 16.0))
        return 0.0
    grid = YeeGrid(design_unbalanced(0, 1, 2000, xmin, dx, 8, LASER, CONDUCTING))
    for _ in range(nsteps + 1):
        world_synchronize([grid], cy, cz, t, dt)
        grid.advance(dt)
        t += dt
    energy = grid.em_field_energy_local()
    target = VACUUM_PERMITTIVITY * powi2(emax) * (193.0 / 32.0) * (pi * SPEED_OF_LIGHT / omega)
    err = abs(energy - target) / target
    assert err < 1.0e-3
    res["conducting_bc (yee.rs:956-1011)"] = {"error": err, "bound": 1e-3, "em_energy": hx(energy)}
    # yee.rs:1014-1079
    t = -5.0e-6 / SPEED_OF_LIGHT
    nsteps = int(10.0e-6 / (SPEED_OF_LIGHT * dt))
    jmax = omega * VACUUM_PERMITTIVITY * 1.0e5
    grid = YeeGrid(design_unbalanced(0, 1, 2000, xmin, dx, 8, INTERNAL, INTERNAL))
    zero = lambda a, b: 0.0  # noqa: E731
    energies = []
    for _half in range(2):
        for _ in range(nsteps):
            phi = omega * t
            if abs(phi) < 4.0 * pi:
                grid.cell[1000].j[2] = jmax * (math.sin(phi) + (2.0 / 8.0) * math.cos(phi) * math.tan(phi / 8.0)) * powi2(math.cos(phi / 8.0))
            else:
                grid.cell[1000].j[2] = 0.0
            world_synchronize([grid], zero, zero, t, dt)
            grid.advance(dt)
            t += dt
        energies.append(grid.em_field_energy_local())
    err = abs(energies[0] - energies[1]) / energies[0]
    assert err < 1.0e-6
    res["periodic_bc (yee.rs:1014-1079)"] = {"error": err, "bound": 1e-6, "em_energy": hx(energies)}
    return res


def main():
    rng = random.Random(20261004)
    out = {
        "_generator": "tests/golden/make_golden_pic_independent.py — a pure-Python restatement written from the reference's "
                      "source, sharing nothing with oracle/ or opal_amd/csrc/; every float is a C99 hex literal",
        "_conventions": {
            "sort": "Population::advance sorts with an UNSTABLE sort (mod.rs:288): the order inside a cell is not defined by "
                    "the reference.  Vectors use a stable sort by cell; comparisons of populations are therefore exact only "
                    "for an implementation with the same tie order, and otherwise hold per cell as multisets.",
            "deposit": "one rayon thread (one chunk): yee.rs:561-567 with nthreads = 1",
            "kinetic_energy": "per-particle values are the reference's; population totals are summed in slot order (rayon's "
                              "reduction tree is not specified)",
            "tau": "the optical depth of electron.rs:284-288 is not carried (emission off: unobservable)",
        },
    }
    w, fl = vectors_weight_flux()
    out["weight"], out["flux"] = w, fl
    out["fields_at"] = vectors_fields_at(rng)
    out["electron_push"], out["ion_push"] = vectors_pushes(rng)
    out["deposit"] = vectors_deposit(rng)
    out["photon_push"] = vectors_photons(random.Random(404))
    out["synchronize_advance"] = vectors_sync_advance(rng)
    out["design_unbalanced"], out["design_balanced"] = vectors_design()
    out["world_runs"] = vectors_world_runs(rng)
    out["branches_covered"] = dict(sorted(BRANCHES.items()))
    if "--skip-self-checks" not in sys.argv:
        out["self_checks"] = self_checks()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pic_step_vectors_independent.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
    for k, v in out["branches_covered"].items():
        print(f"  {v:7d}  {k}")
    for k, v in out.get("self_checks", {}).items():
        print(f"  self-check {k}: error {v['error']:.3e} (bound {v['bound']:g})")


if __name__ == "__main__":
    main()
