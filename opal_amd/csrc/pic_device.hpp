// Device-side arithmetic of the PIC step for gfx950: shape functions, field
// gather, the Vay (electron) and Boris (ion) pushers, and the deposit taps.
// Every function cites the reference lines whose arithmetic it reproduces;
// the operation order inside each expression follows the reference so that the
// only differences from the CPU oracle are FMA contraction and summation order.
#pragma once
#include <hip/hip_runtime.h>

namespace opal {

// src/constants.rs:4-27 (deliberately truncated literals; C2 is NOT C*C)
constexpr double SPEED_OF_LIGHT = 2.997925e8;
constexpr double SPEED_OF_LIGHT_SQD = 89875517873681764.0;
constexpr double VACUUM_PERMITTIVITY = 8.854188e-12;
constexpr double VACUUM_PERMEABILITY = 1.256637e-6;
constexpr double ELECTRON_CHARGE = -1.602177e-19;
constexpr double ELEMENTARY_CHARGE = 1.602177e-19;
constexpr double ELECTRON_MASS = 9.109383e-31;
constexpr double PROTON_MASS = 1.672622e-27;
constexpr double ELECTRON_MASS_MEV = 0.510999;
constexpr double CRITICAL_FIELD = 1.323285e18;

// yee.rs:239-242
constexpr int GHOST_SIZE = 4;
constexpr int LASER_BDY_SIZE = 4;
constexpr int ABSORBING_BDY_SIZE = 200;
constexpr int CONDUCTING_BDY_SIZE = 4;

constexpr int DEAD_CELL = -(1 << 30);  // slot holds no particle

struct V3 {
    double x, y, z;
};
__host__ __device__ inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__host__ __device__ inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline V3 cross(V3 a, V3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Cell::weight, yee.rs:140-149 (branch-free: both polynomials, then select)
__host__ __device__ inline double weight(double xi)
{
    double xhat = fabs(xi);
    double inner = 0.75 - xhat * xhat;
    double outer = 1.125 - 1.5 * xhat + 0.5 * (xhat * xhat);
    double w = xhat < 0.5 ? inner : outer;
    return xhat > 1.5 ? 0.0 : w;
}

// Cell::flux, yee.rs:185-204, without divergent branches.  With pa := 0 when
// |x_i| >= 1 and pb := 0 when |x_f| >= 1 the first, second, fourth and fifth arms
// of the reference all read |pb - pa| carrying the sign of (x_i - x_f): for a
// boundary that moves by less than one cell (asserted by the pushers) the signs
// the reference takes from -x_i or x_f coincide with that of x_i - x_f.
__host__ __device__ inline double flux(double x_i, double x_f)
{
    double a = fabs(x_i), b = fabs(x_f);
    bool ina = a < 1.0, inb = b < 1.0;
    double ca = 1.0 - a, cb = 1.0 - b;
    double pa = ina ? 0.5 * (ca * ca) : 0.0;
    double pb = inb ? 0.5 * (cb * cb) : 0.0;
    double same = copysign(pb - pa, x_i - x_f);
    double diff = copysign(a * (1.0 - 0.5 * a) + b * (1.0 - 0.5 * b), x_i);
    bool opposite = ina && inb && !(x_i * x_f >= 0.0);
    return opposite ? diff : same;
}

// 2 x Cell::flux in closed form.  The reference's five arms (yee.rs:185-204) are the pieces of ONE expression: the
// flux through a boundary is the part of the particle's triangular shape (half-width one cell) that crossed it,
// G(x_i) - G(x_f) with G the cumulative triangle and x = boundary - particle; for a move of less than a cell
// (asserted by the pushers) every arm is this difference with the constants cancelled by hand.  With
//     h2(x) = 2 G(x) - 1 = sign(x) (1 - max(1 - |x|, 0)^2)
// flux2 = h2(x_i) - h2(x_f): four instructions per end point, no compare, no select, where the arm-by-arm form
// (flux() above, what the oracle evaluates) takes about 25 per boundary.  A particle that has not moved still
// deposits exactly zero (equal arguments give equal h2).  The two agree to an ulp of the unit flux (2e-16 of the
// charge, whatever the displacement): the oracle comparisons hold 1e-12 of the deposit scale.
__host__ __device__ inline double tri_cdf2(double x)
{
    double c = fmax(1.0 - fabs(x), 0.0);
    return copysign(__builtin_fma(-c, c, 1.0), x);
}
__host__ __device__ inline double flux2(double x_i, double x_f) { return tri_cdf2(x_i) - tri_cdf2(x_f); }
__host__ __device__ inline double flux2_outer(double x_i, double displacement)
{
    double ca = fmax(1.0 - fabs(x_i), 0.0);
    return copysign(ca * ca, displacement);
}

// Cell::flux for the two outermost taps of the deposit (boundaries at -1.5 and 2.5 cells from the
// left edge of the particle's cell): with the final offset x in [0, 1] the boundary is at least 1.5
// cells from the particle's final position, so |x_f| >= 1 always and only the first and last arms of
// the reference can fire: the result is 0.5 (1 - |x_i|)^2 for |x_i| < 1, else 0, with the sign of the
// displacement (see flux()).
__host__ __device__ inline double flux_outer(double x_i, double displacement)
{
    double a = fabs(x_i);
    double ca = 1.0 - a;
    double pa = a < 1.0 ? 0.5 * (ca * ca) : 0.0;
    return copysign(pa, displacement);
}

// The two polynomial pieces of Cell::weight (yee.rs:145,147) for callers that know the range of
// their argument, which spares the compares and selects of the general form.  The outer piece
// 1.125 - 1.5 xhat + 0.5 xhat^2 is evaluated as 0.5 (1.5 - xhat)^2: the same polynomial (differs from the
// reference's evaluation order by an ulp of 1), three instructions instead of five.
__host__ __device__ inline double w_inner(double xhat) { return 0.75 - xhat * xhat; }                          // xhat < 0.5
__host__ __device__ inline double w_outer(double xhat) { double d = 1.5 - xhat; return 0.5 * (d * d); }        // 0.5 <= xhat <= 1.5

// weight(1+x), weight(x), weight(1-x), weight(2-x) for an offset x in [0, 1] (the stored offsets:
// electron.rs:319-329 / ion.rs:203-213 leave x - floor(x)), without compares or selects.  With h = x - 0.5,
// p = max(h, 0) and m = max(-h, 0):
//   weight(1+x) = 0.5 (0.5 - x)^2 for x <= 0.5, else 0           = 0.5 m^2
//   weight(2-x) = 0.5 (x - 0.5)^2 for x >= 0.5, else 0           = 0.5 p^2
//   weight(x)   = 0.75 - x^2 for x < 0.5, else 0.5 (1.5 - x)^2   = 0.75 - x^2 + 1.5 p^2
//   weight(1-x) = the same in 1 - x                              = 0.75 - (1-x)^2 + 1.5 m^2
// (0.5 (1.5 - x)^2 - (0.75 - x^2) = 1.5 (x - 0.5)^2).  Each agrees with the reference's piecewise evaluation
// to an ulp of 1; at x = 0.5 the pieces meet, so the choice there is immaterial.
struct EdgeW {
    double e0, e1, e2, e3;
};
__host__ __device__ inline EdgeW edge_weights(double x)
{
    EdgeW w;
    const double h = x - 0.5;
    const double p = fmax(h, 0.0), m = fmax(-h, 0.0);
    const double p2 = p * p, m2 = m * m;
    const double a2 = 1.0 - x;
    w.e0 = 0.5 * m2;
    w.e3 = 0.5 * p2;
    w.e1 = (0.75 - x * x) + 1.5 * p2;
    w.e2 = (0.75 - a2 * a2) + 1.5 * m2;
    return w;
}

// The seven gather weights of fields_at, yee.rs:504-526, for an offset xi in [0, 1]
struct GatherW {
    double c0, c1, c2;      // centred quantities (Ex, By, Bz): cells j-1, j, j+1
    double e0, e1, e2, e3;  // edge quantities (Ey, Ez): cells j-1 .. j+2
};
__host__ __device__ inline GatherW gather_weights(double xi)
{
    GatherW g;
    const double a = 1.0 - xi, h = 0.5 - xi;
    g.c0 = 0.5 * (a * a);    // weight(0.5 + xi), 0.5+xi in [0.5, 1.5]: 0.5 (1.5 - (0.5 + xi))^2
    g.c1 = 0.75 - h * h;     // weight(0.5 - xi), |0.5-xi| <= 0.5
    g.c2 = 0.5 * (xi * xi);  // weight(1.5 - xi), 1.5-xi in [0.5, 1.5]
    EdgeW e = edge_weights(xi);
    g.e0 = e.e0;
    g.e1 = e.e1;
    g.e2 = e.e2;
    g.e3 = e.e3;
    return g;
}

// fields_at, yee.rs:499-529.  F is any accessor F(comp, cell_index) over the six
// field components 0..5 = Ex Ey Ez Bx By Bz; j is the array index of the cell.
// ES: the caller knows Ey, Ez and B to be zero in every cell this gather can touch (see push_body) and reads Ex only.
template <bool ES = false, class F>
__host__ __device__ inline void gather(const F &f, int j, double xi, V3 &E, V3 &B)
{
    GatherW g = gather_weights(xi);
    E.x = f(0, j - 1) * g.c0 + f(0, j) * g.c1 + f(0, j + 1) * g.c2;
    if (ES) {
        E.y = E.z = 0.0;
        B = {0.0, 0.0, 0.0};
        return;
    }
    E.y = f(1, j - 1) * g.e0 + f(1, j) * g.e1 + f(1, j + 1) * g.e2 + f(1, j + 2) * g.e3;
    E.z = f(2, j - 1) * g.e0 + f(2, j) * g.e1 + f(2, j + 1) * g.e2 + f(2, j + 2) * g.e3;
    B.x = f(3, j);
    B.y = f(4, j - 1) * g.c0 + f(4, j) * g.c1 + f(4, j + 1) * g.c2;
    B.z = f(5, j - 1) * g.c0 + f(5, j) * g.c1 + f(5, j + 1) * g.c2;
}

// Result of one push: new momentum, Lorentz factor, displacement in cells.
struct Pushed {
    V3 u;
    double gamma;      // electrons: Vay gamma; ions: 1 + gamma_m1
    double inv_gamma;  // 1 / gamma, shared by the displacement and the deposit's velocity
    double dxi;
    // electron diagnostics (only meaningful when requested)
    double chi, dwork, gamma_half;
    V3 v_old;
};

// 1/sqrt(a): hardware estimate (v_rsq_f64, ~2^-26 relative) plus OPAL_NR_STEPS Newton steps.  Used where the
// reference takes a square root only to divide by it.  One step squares the error (1.5 e^2 ~ 3e-16, a few ulp):
// the single-operator parity tests hold 1e-12 with it, and a second step (measured: +2.5 % kernel time) buys
// nothing the tolerances can see.
#ifndef OPAL_NR_STEPS
#define OPAL_NR_STEPS 1
#endif
__device__ inline double rsqrt_nr(double a)
{
    double y = __builtin_amdgcn_rsq(a);
    double h = 0.5 * a;
#pragma unroll
    for (int i = 0; i < OPAL_NR_STEPS; i++) y = y * (1.5 - h * y * y);
    return y;
}

// 1/a: hardware estimate (v_rcp_f64) plus OPAL_NR_STEPS Newton steps, three instructions where the IEEE-exact
// division sequence (div_scale, rcp, fma chain, div_fmas, div_fixup) takes a dozen.
__device__ inline double rcp_nr(double a)
{
    double y = __builtin_amdgcn_rcp(a);
#pragma unroll
    for (int i = 0; i < OPAL_NR_STEPS; i++) y = y * (2.0 - a * y);
    return y;
}

// Electron::vay_push, electron.rs:268-330 (momentum + displacement part).
// The reference carries gamma in the struct and divides by it; here 1/gamma_old
// is recomputed as 1/sqrt(1 + u.u) (equal up to rounding) so that gamma never
// travels through HBM.
//
// B0 = true (the electrostatic launch, particles.hip) is the same arithmetic with B == 0 put in: every term dropped below is an exact zero or an exact
// identity of the general form (v x B = 0; tau = 0, so u* = tau^2 = 0, sigma = gamma'^2, t = 0, s = 1/(1 + 0) = 1 --
// the Newton reciprocal of 1.0 is 1.0 -- and u_new = u'), so the two forms give the same numbers (up to the sign of
// a zero) and a particle's result does not depend on which of them its wave took.
template <bool EXTRAS, bool B0 = false>
__device__ inline Pushed vay_push(V3 u, V3 E, V3 B, double alpha, double dt_over_dx, double dt)
{
    Pushed r;
    if (B0 && !EXTRAS) {
        V3 u_prime = (u + alpha * E) + alpha * E;
        double sigma = 1.0 + dot(u_prime, u_prime);
        double rad = 0.25 * (sigma * sigma);
        double g2 = 0.5 * sigma + rad * rsqrt_nr(rad);
        double inv_gn = rsqrt_nr(g2);
        r.u = u_prime;
        r.gamma = g2 * inv_gn;
        r.inv_gamma = inv_gn;
        r.dxi = SPEED_OF_LIGHT * r.u.x * dt_over_dx * inv_gn;
        return r;
    }
    double inv_g = rsqrt_nr(1.0 + dot(u, u));
    V3 v = (SPEED_OF_LIGHT * inv_g) * u;
    V3 u_half = u + alpha * (E + cross(v, B));
    if (EXTRAS) {
        double gh2 = 1.0 + dot(u_half, u_half);
        double gamma_half = sqrt(gh2);
        r.dwork = ELECTRON_CHARGE * SPEED_OF_LIGHT * dot(u_half, E) * dt / gamma_half;
        V3 w = gamma_half * E + SPEED_OF_LIGHT * cross(u_half, B);
        double eu = dot(E, u_half);
        r.chi = sqrt(dot(w, w) - eu * eu) / CRITICAL_FIELD;
        r.gamma_half = gamma_half;
        r.v_old = v;
    }
    V3 u_prime = u_half + alpha * E;
    double gamma_prime_sqd = 1.0 + dot(u_prime, u_prime);
    V3 tau = (alpha * SPEED_OF_LIGHT) * B;
    double u_star = dot(u_prime, tau);
    double tau2 = dot(tau, tau);
    double sigma = gamma_prime_sqd - tau2;
    double rad = 0.25 * (sigma * sigma) + tau2 + u_star * u_star;  // > 0: tau2 and sigma cannot both vanish
    double g2 = 0.5 * sigma + rad * rsqrt_nr(rad);
    double inv_gn = rsqrt_nr(g2);
    double gamma = g2 * inv_gn;
    V3 t = inv_gn * tau;
    double s = rcp_nr(1.0 + dot(t, t));  // in [0.5, 1]: no range concerns
    r.u = s * (u_prime + dot(u_prime, t) * t + cross(u_prime, t));
    r.gamma = gamma;
    r.inv_gamma = inv_gn;
    r.dxi = SPEED_OF_LIGHT * r.u.x * dt_over_dx * inv_gn;
    return r;
}

// Photon::push, photon.rs:150-183: ballistic, with the quantum parameter of the local field.
template <bool EXTRAS>
__device__ inline Pushed photon_push(V3 k, V3 E, V3 B, double dt_over_dx)
{
    Pushed r;
    double k0 = sqrt(dot(k, k));
    V3 v = (SPEED_OF_LIGHT / k0) * k;
    if (EXTRAS) {
        V3 w = k0 * E + SPEED_OF_LIGHT * cross(k, B);
        double ek = dot(E, k);
        r.chi = sqrt(dot(w, w) - ek * ek) / CRITICAL_FIELD;
        r.v_old = v;
    }
    r.u = k;
    r.gamma = k0;
    r.inv_gamma = 1.0 / k0;
    r.dxi = v.x * dt_over_dx;
    return r;
}

// Ion::push, ion.rs:168-214 (Boris).  alpha = q dt / (2 M c) with q = Z* e, M = A m_p.
// B0 = true: the same with B == 0 put in (both rotations add exact zeros: u_plus = u_minus), see vay_push.
template <bool B0 = false>
__device__ inline Pushed boris_push(V3 u, V3 E, V3 B, double alpha, double dt_over_dx)
{
    Pushed r;
    if (B0) {
        r.u = (u + alpha * E) + alpha * E;
        double g2 = 1.0 + dot(r.u, r.u);
        r.inv_gamma = rsqrt_nr(g2);
        r.gamma = g2 * r.inv_gamma;
        r.dxi = SPEED_OF_LIGHT * r.u.x * dt_over_dx * r.inv_gamma;
        return r;
    }
    V3 cB = SPEED_OF_LIGHT * B;
    V3 u_minus = u + alpha * E;
    double um2 = dot(u_minus, u_minus);
    // ion.rs:183 writes gamma as 1 + u^2 / (1 + sqrt(1 + u^2)), which is sqrt(1 + u^2); t = alpha / gamma
    double t = alpha * rsqrt_nr(1.0 + um2);
    V3 u_prime = u_minus + t * cross(u_minus, cB);
    double t_prime = (2.0 * t) * rcp_nr(1.0 + (t * t) * dot(cB, cB));  // the denominator is >= 1
    V3 u_plus = u_minus + t_prime * cross(u_prime, cB);
    r.u = u_plus + alpha * E;
    double g2 = 1.0 + dot(r.u, r.u);
    // ion.rs:194,198: 1 + gamma_m1 with gamma_m1 = u^2 / (1 + sqrt(1 + u^2)), i.e. sqrt(1 + u^2) again
    r.inv_gamma = rsqrt_nr(g2);
    r.gamma = g2 * r.inv_gamma;
    r.dxi = SPEED_OF_LIGHT * r.u.x * dt_over_dx * r.inv_gamma;
    return r;
}

// The deposit stencil of one particle, yee.rs:584-609, relative to its (new)
// cell index: jx[k] -> index + k - 2 (k = 0..4); jy, jz[k] -> index + k - 1
// (k = 0..2; the reference's fourth tap weight(2.0 + x) is identically zero);
// rho[k] -> index + k - 2 (k = 0..3; the reference adds W(2 - x) at index - 2).
struct Taps {
    double jx[5];
    double e[3];    // weight(1+x), weight(x), weight(1-x): the shape shared by jy, jz and three of the rho taps
    double qy, qz;  // Q v_y / dx, Q v_z / dx: jy[k] = qy e[k], jz[k] = qz e[k]
    double rho[4];
    __device__ double jy(int k) const { return qy * e[k]; }
    __device__ double jz(int k) const { return qz * e[k]; }
};

template <bool RHO>
__device__ inline Taps deposit_taps(double x, double prev_x, double macrocharge, double vy, double vz,
                                    double inv_dt, double inv_dx)
{
    Taps t;
    // jx = Q flux / dt (yee.rs:584-588) with flux = 0.5 * flux2: one constant carries Q, 1/dt and the 0.5
    const double hq = 0.5 * (macrocharge * inv_dt);
    const double d = x - prev_x;
    t.jx[2] = hq * flux2(0.5 - prev_x, 0.5 - x);
    t.jx[1] = hq * flux2(-0.5 - prev_x, -0.5 - x);
    t.jx[0] = hq * flux2_outer(-1.5 - prev_x, d);
    t.jx[3] = hq * flux2(1.5 - prev_x, 1.5 - x);
    t.jx[4] = hq * flux2_outer(2.5 - prev_x, d);
    EdgeW ew = edge_weights(x);
    // j_perp = Q v W / dx (yee.rs:594-602), rho = Q W / dx (:606-609).  The products with the weights are
    // left to the scatter, which skips a transverse component whose velocity is zero in every lane.
    const double qdx = macrocharge * inv_dx;
    t.qy = qdx * vy;
    t.qz = qdx * vz;
    t.e[0] = ew.e0;
    t.e[1] = ew.e1;
    t.e[2] = ew.e2;
    if (RHO) {
        t.rho[0] = qdx * ew.e3;
        t.rho[1] = qdx * ew.e0;
        t.rho[2] = qdx * ew.e1;
        t.rho[3] = qdx * ew.e2;
    }
    return t;
}

}  // namespace opal
