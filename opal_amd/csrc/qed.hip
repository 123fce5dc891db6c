// Photon emission on the GPU: emit_radiation (interactions.rs:45-107) = Electron::radiate
// (electron.rs:208-251) for every electron whose optical depth has run out, sampling the angularly
// resolved quantum synchrotron spectrum (photon_emission.rs:87-292) by inverting tabulated cdfs with
// piecewise monotone cubic interpolation (pwmci.rs), and appending the photons that pass the filters
// to the photon population.
//
// GPU shape: a streaming pass reads every electron's optical depth (12 B per slot) and compacts the few
// that emit (a fraction rate * dt of the population per step) into a list; a second, dense pass over the
// list draws from a counter-based generator keyed by (seed, particle id, step) -- so the result does not
// depend on how particles are laid out or scheduled, unlike the reference's per-thread jumped Xoshiro
// streams (interactions.rs:58-67) -- samples, recoils, and takes a photon slot with one atomic.  The sampling tables (37 KB: the
// reference's numbers, transcribed as constant data by tools/transcribe_qed_tables.py) sit in device memory and
// stay cache-resident.
#include "opal_internal.hpp"
#include "pic_device.hpp"
#include "qed_device.hpp"

#include <cmath>
#include <cstring>
#include <vector>

using namespace opal;

namespace opal {
int pop_check_err(opal_pop_t *p);
int pop_reserve(opal_pop_t *p, size_t extra);
int pop_tighten_upper(opal_pop_t *p);
int pop_snapshot_upper(opal_pop_t *p, size_t just_added);
}

namespace {

struct QedCdf {
    double coeff, power;
    double table[31][2];
};
#define QED_TABLE_QUAL __device__ const
#include "qed_tables.inc"

// The sampler below is compiled WITHOUT floating-point contraction: Brent's iteration in pwmci_invert
// stops on a 1e-6 bracket and branches on comparisons of residuals, so a fused multiply-add can return a
// different iterate than the reference's (Rust never fuses).  Unfused, the device differs from the CPU
// restatement only by the last-ulp differences of log/exp/pow/acos, and tests/test_gpu_qed.py holds the
// golden vectors of tests/golden/qed_sample_reference_tables.json to 1e-12.  k_radiate is ~70 us of a colliding-beams step
// at its peak (1e5-3e5 emitters).  In the steps it is long in, a launch is about one wave per SIMD, every wave resident from
// the start: it lasts as long as one wave takes -- ~0.025 ms for the emitters' attributes gathered by slot, the scattered
// stores and the photons' slots (the kernel with the sampler taken out), ~0.04 ms for the sampler's ~1e4 instructions one
// after the other.  Brent's loop diverges (3.5 passes per inversion on average, 8.7 for the slowest of 64 lanes, 28 of 64
// lanes active per VALU instruction: tests/test_qed_oracle.py::test_brent_iterations_have_a_long_tail,
// profiles/r05_radiate_pmc_sq.json), but the launch does not wait for it: more resident waves, fewer instructions, the two
// inversions of a stage side by side, a pool of inversions shared among a wave's lanes, the tables in LDS, shared instead
// of inlined functions, even the loop cut off at 8 passes each move the kernel by -6 to +11 %
// (profiles/r05_ab/ab_radiate_*.txt; patches under scripts/archive/).  The reference's iterates are the result, so the
// loop stays as it is; what is left to gain is the launch BESIDE the photons' push (the GPU is nearly idle meanwhile).
#pragma clang fp contract(off)

constexpr int RECURSION_LIMIT = 32;      // pwmci.rs:3
constexpr double PRECISION_LIMIT = 1.0e-6;  // pwmci.rs:4

typedef const double (*Table)[2];

// Cubic Hermite segment between rows i-1 and i with monotonicity-preserving tangents, pwmci.rs:14-78
struct Fit {
    double x0, x1, f0, f1, m0, m1;
    __device__ Fit(int i, Table t, int len)
    {
        x0 = t[i - 1][0]; x1 = t[i][0];
        f0 = t[i - 1][1]; f1 = t[i][1];
        double mid = (f1 - f0) / (x1 - x0);
        // the secant before / after; at the table's ends the missing one repeats the middle one (:18-37)
        double before = i == 1 ? mid : (f0 - t[i - 2][1]) / (x0 - t[i - 2][0]);
        double after = i == len - 1 ? mid : (t[i + 1][1] - f1) / (t[i + 1][0] - x1);
        m0 = before * mid > 0.0 ? 0.5 * (before + mid) : 0.0;
        m1 = mid * after > 0.0 ? 0.5 * (mid + after) : 0.0;
        if (mid != 0.0 && m0 / mid > 3.0) m0 = 3.0 * mid;
        if (after != 0.0 && m1 / after > 3.0) m1 = 3.0 * after;
    }
    __device__ double operator()(double x) const
    {
        double t = (x - x0) / (x1 - x0);
        double omt2 = (1.0 - t) * (1.0 - t), t2 = t * t;
        double h00 = (1.0 + 2.0 * t) * omt2, h10 = t * omt2, h01 = t2 * (3.0 - 2.0 * t), h11 = t2 * (t - 1.0);
        return f0 * h00 + f1 * h01 + (x1 - x0) * (m0 * h10 + m1 * h11);
    }
};

// first row whose column `col` is >= v, pwmci.rs:99-102,116-119.  The reference scans from the first row; the same row is
// found here by bisection over the rows [0, len - 4), which are non-decreasing in both columns in every table
// (tests/test_qed_oracle.py::test_table_prefixes_are_monotone), and a scan of the last four, where the tabulated ln cdf is
// rounding noise about zero: 5 dependent loads instead of up to 31-41 (round 4: k_radiate is a chain of these look-ups
// per emitter and was 21 % of the GPU time of the colliding-beams step).
__device__ inline int first_row_not_below(double v, Table t, int len, int col)
{
    int lo = 0, hi = len - 4;  // the answer is in [lo, hi] if it is below len - 4
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v > t[mid][col]) lo = mid + 1;
        else hi = mid;
    }
    int i = lo;
    while (i < len && v > t[i][col]) i++;
    return i;
}

// pwmci::evaluate, pwmci.rs:96-111
__device__ bool pwmci_evaluate(double x, Table t, int len, double &out)
{
    int i = first_row_not_below(x, t, len, 0);
    if (i >= len - 1 && x > t[len - 1][0]) return false;
    if (i == 0) i = 1;  // x == t[0][0]: the reference would index row -1
    out = Fit(i, t, len)(x);
    return true;
}

__device__ inline bool is_between(double x, double lo, double hi)
{
    if (lo < hi) return x > lo && x < hi;
    if (hi < lo) return x > hi && x < lo;
    return true;
}

// pwmci::invert, pwmci.rs:113-228: Brent's method (inverse quadratic interpolation, secant, bisection)
// on the Hermite segment that brackets f
__device__ bool pwmci_invert(double f, Table t, int len, double &out)
{
    int i = first_row_not_below(f, t, len, 1);
    if (i >= len - 1 && f > t[len - 1][1]) return false;
    if (i == 0) i = 1;
    const Fit fit(i, t, len);
    struct Root { double x, f; };
    Root r0 = {t[i - 1][0], t[i - 1][1] - f}, r1 = {t[i][0], t[i][1] - f}, r2, r3 = {0.0, 0.0};
    if (fabs(r1.f) < fabs(r0.f)) { Root tmp = r0; r0 = r1; r1 = tmp; }
    r2 = r1;
    bool prev_bisect = false;
    for (int it = 0; it < RECURSION_LIMIT; it++) {
        double s;
        if (r0.f != r2.f && r1.f != r2.f) {
            double r = r0.f / r2.f, ss = r0.f / r1.f, tt = r1.f / r2.f;
            double p = ss * (tt * (r - tt) * (r2.x - r0.x) - (1.0 - r) * (r0.x - r1.x));
            double q = (tt - 1.0) * (r - 1.0) * (ss - 1.0);
            s = r0.x + p / q;
        } else {
            s = r0.x - (r0.f * (r0.x - r1.x) / (r0.f - r1.f));
        }
        if (!is_between(s, 0.25 * (3.0 * r1.x + r0.x), r0.x) || (prev_bisect && fabs(s - r0.x) >= 0.5 * fabs(r0.x - r2.x))
            || (!prev_bisect && fabs(s - r0.x) >= 0.5 * fabs(r2.x - r3.x))) {
            prev_bisect = true;
            s = 0.5 * (r0.x + r1.x);
        } else {
            prev_bisect = false;
        }
        r3 = r2;
        r2 = r0;
        r0.x = s;
        r0.f = fit(s) - f;
        if (r1.f * r0.f >= 0.0) {
            r1 = r0;
            r0 = r2;
        }
        if (fabs(r1.f) < fabs(r0.f)) { Root tmp = r0; r0 = r1; r1 = tmp; }
        double error = fabs(r1.x - r0.x) / fabs(r0.x);
        if (error < PRECISION_LIMIT || r0.f == 0.0) break;
    }
    out = r0.x;
    return true;
}

// photon_emission.rs:87-121: draw y > local_zero from a cdf tabulated on (global_zero, inf)
__device__ double from_linear_cdf_table(double global_zero, double local_zero, double rnd, const QedCdf &cdf)
{
    double r_zero;
    if (local_zero < cdf.table[0][0]) r_zero = cdf.coeff * pow(local_zero - global_zero, cdf.power);
    else if (!pwmci_evaluate(local_zero, cdf.table, 31, r_zero)) return local_zero;
    double r = r_zero + (1.0 - r_zero) * rnd;
    double y;
    if (r <= cdf.table[0][1]) y = exp((log(r) - log(cdf.coeff)) / cdf.power) + global_zero;
    else if (!pwmci_invert(r, cdf.table, 31, y)) y = local_zero;
    return fmax(y, local_zero);  // the reference asserts y >= local_zero (:119)
}

struct Sample {
    double omega_mc2, theta, phi;
};

__device__ inline double polar_angle(double gamma, double z)
{
    // photon_emission.rs:197-198, 289-290
    double cos_theta = (gamma - pow(z, 2.0 / 3.0) / (2.0 * gamma)) / sqrt(gamma * gamma - 1.0);
    return acos(fmax(fmin(cos_theta, 1.0), -1.0));
}

// photon_emission.rs:264-292
__device__ Sample classical_sample(double chi, double gamma, double rand1, double rand2, double rand3)
{
    // within ~3e-8 of rand2 = 1 the argument can round to just above 1, where the reference's acos gives
    // NaN (photon_emission.rs:268): clamped here, which is the limit the formula tends to
    double delta = acos(fmin((-9.0 + 50.0 * rand2 - 25.0 * (rand2 * rand2)) / 16.0, 1.0));
    double zc = (2.0 + 4.0 * cos(delta / 3.0)) / (5.0 * (1.0 - rand2));
    double z = zc * zc * zc;
    double ln_rand = log(rand1), x;
    if (ln_rand < QED_CLASSICAL_SPECTRUM[0][1]) {
        // cdf -> A x^(5/3) below the table; the reference carries A^(-3/5) as a 10-digit literal (:275)
        x = QED_CLASSICAL_SMALL_X_COEFF * pow(rand1, 0.6);
    } else {
        double ln_x;
        if (!pwmci_invert(ln_rand, QED_CLASSICAL_SPECTRUM, 41, ln_x)) ln_x = QED_CLASSICAL_SPECTRUM[40][0];
        x = exp(ln_x);
    }
    double u = 3.0 * chi * x / (2.0 * z);
    return {u * gamma, polar_angle(gamma, z), 2.0 * M_PI * rand3};
}

__device__ inline double quantum_ln_u(const QedCdf &c, double ln_rand)
{
    double r;
    if (ln_rand <= c.table[0][1]) return (ln_rand - log(c.coeff)) / c.power;
    if (pwmci_invert(ln_rand, c.table, 31, r)) return r;
    return c.table[30][0];
}

// photon_emission.rs:129-203; false where the reference stops at unimplemented!() (chi >= 100)
__device__ bool quantum_sample(double chi, double gamma, double rand1, double rand2, double rand3, Sample &out)
{
    double ln_chi = log(chi);
    if (ln_chi <= QED_LN_CHI_MIN) {
        out = classical_sample(chi, gamma, rand1, rand2, rand3);
        out.omega_mc2 = out.omega_mc2 * gamma / (gamma + out.omega_mc2);
        return true;
    }
    double index = (ln_chi - QED_LN_CHI_MIN) / QED_LN_CHI_STEP;
    double fl = floor(index);
    double weight = index - fl;
    if (!(fl < 40.0)) return false;
    int i = (int)fl;
    double ln_rand = log(rand1);
    double u = exp((1.0 - weight) * quantum_ln_u(QED_QUANTUM_CDF[i], ln_rand) + weight * quantum_ln_u(QED_QUANTUM_CDF[i + 1], ln_rand));

    double beta = 2.0 * u / (3.0 * chi);
    double delta = (1.0 + (1.0 + u) * (1.0 + u)) * pow(beta, -2.0 / 3.0) / (1.0 + u);
    index = (log(delta) - QED_LN_DELTA_MIN) / QED_LN_DELTA_STEP;
    fl = floor(index);
    weight = index - fl;
    double y;
    if (!(fl < 30.0)) {  // delta > 100 (or not a number): the delta -> infinity table
        y = from_linear_cdf_table(0.0, beta, rand2, QED_Y_INFINITE_DELTA_CDF[0]);
    } else {
        int j = max((int)fl, 0);  // delta >= 0.1 for every u when chi > 0.01 (:170-171)
        double y0 = pow(delta, -1.5);
        double y_lower = from_linear_cdf_table(y0, beta, rand2, QED_Y_CDF[j]);
        double y_upper = from_linear_cdf_table(y0, beta, rand2, QED_Y_CDF[j + 1]);
        y = (1.0 - weight) * y_lower + weight * y_upper;
    }
    double z = fmax(y / beta, 1.0);
    out = {gamma * u / (1.0 + u), polar_angle(gamma, z), 2.0 * M_PI * rand3};
    return true;
}

// vec3.rs:115-138
__device__ inline V3 normalize(V3 a) { return (1.0 / sqrt(dot(a, a))) * a; }
__device__ inline V3 orthogonal(V3 a)
{
    V3 perp = fabs(a.x) > fabs(a.z) ? V3{-a.y, a.x, 0.0} : V3{0.0, -a.z, a.y};
    return normalize(perp);
}
__device__ inline V3 rotate_around(V3 v, V3 ax, double theta)
{
    double s, c;
    sincos(theta, &s, &c);
    double omc = 1.0 - c;
    return V3{(c + ax.x * ax.x * omc) * v.x + (ax.x * ax.y * omc - ax.z * s) * v.y + (ax.x * ax.z * omc + ax.y * s) * v.z,
              (ax.y * ax.x * omc + ax.z * s) * v.x + (c + ax.y * ax.y * omc) * v.y + (ax.y * ax.z * omc - ax.x * s) * v.z,
              (ax.z * ax.x * omc - ax.y * s) * v.x + (ax.z * ax.y * omc + ax.x * s) * v.y + (c + ax.z * ax.z * omc) * v.z};
}

struct RadiateArgs {
    // electrons
    const int *cell;
    const double *x, *w;
    double *ux, *uy, *uz, *chi, *tau;
    const double *id;
    PopCounters *ectr;
    const unsigned *list;  // slots of the emitters (their number is ectr->n_emit)
    // photons
    int *pcell;
    double *px, *pkx, *pky, *pkz, *pw;
    double *pchi, *py, *pz, *pid;  // photon extras, each may be null
    double *ptau_abs, *ptau_st, *pbirth;  // with photon absorption on: the new photon's optical depths and birth time
    double t;
    PopCounters *pctr;
    unsigned pcap;
    unsigned long long seed;
    unsigned step;
    double min_energy, max_angle, max_formation_length;  // < 0: no limit
};

// Pass 1 -- normally done by the electron push itself (particles.hip, push_body: the emitters are compacted
// where tau is computed); this kernel is the fallback for callers that changed optical depths or moved particles
// between the push and the emission.  The slots whose optical depth has run out (electron.rs:209), compacted into
// a list.  Only a
// fraction rate * dt of the electrons emits in a step; sampling straight from the particle arrays would
// leave 63 of 64 lanes idle through the long table-inversion code, so the emitters are gathered first and
// pass 2 runs with full waves.  A workgroup scans LIST_ITER x 256 consecutive slots, keeps its hits in a
// register bit mask, and reserves its stretch of the list with ONE atomic (a per-wave atomic on the single
// counter serialises badly in the steps where most waves hold an emitter: measured 1 ms instead of 20 us).
// Within a workgroup the list keeps slot order; across workgroups the order is arbitrary, which only
// permutes the photons.
constexpr int LIST_ITER = 8;
__global__ __launch_bounds__(256) void k_list_emitters(const int *__restrict__ cell, const double *__restrict__ tau,
                                                       PopCounters *ctr, unsigned *__restrict__ list)
{
    __shared__ unsigned s_wave[4];
    __shared__ unsigned s_base;
    const unsigned n = ctr->n_slots;
    const size_t chunk0 = (size_t)blockIdx.x * 256 * LIST_ITER;
    if (chunk0 >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned hits = 0;
#pragma unroll
    for (int it = 0; it < LIST_ITER; it++) {
        size_t i = chunk0 + (size_t)it * 256 + threadIdx.x;
        if (i < n && tau[i] < 0.0 && cell[i] != DEAD_CELL) hits |= 1u << it;
    }
    // exclusive prefix of the per-thread counts over the workgroup
    unsigned mine = __popc(hits), incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
    }
    if (total == 0) return;  // block-uniform
    if (threadIdx.x == 0) s_base = atomicAdd(&ctr->n_emit, total);
    __syncthreads();
    unsigned at = s_base + before + incl - mine;
#pragma unroll
    for (int it = 0; it < LIST_ITER; it++)
        if (hits & (1u << it)) list[at++] = (unsigned)(chunk0 + (size_t)it * 256 + threadIdx.x);
}

__device__ void radiate_one(const RadiateArgs &a, size_t i);

// Pass 2: Electron::radiate + the filters of emit_radiation, one listed emitter per thread.  One-wave workgroups: the kernel
// holds 2 waves per SIMD (201 VGPRs) and a launch is 1-5 waves per SIMD of work, dispatched wave by wave instead of four
// at a time (-3 % on the kernel; more resident waves bought with scratch are slower, and nothing else that was tried on it
// moved it by more: the notes at the top of this file, profiles/r05_ab/ab_radiate_*.txt).
constexpr int RADIATE_BLOCK = 64;
__global__ __launch_bounds__(RADIATE_BLOCK) void k_radiate(RadiateArgs a)
{
    // the list length is read from the device: the host may not know it (asynchronous mode), so the grid is
    // sized for the most it can be and strides over the list
    const unsigned n_list = a.ectr->n_emit;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n_list; e += gridDim.x * blockDim.x) radiate_one(a, a.list[e]);
}

// electron.rs:219-240: the photon's momentum at polar angle theta around the electron's direction from the sampled
// (omega / mc^2, theta, phi), the formation length (:231), the recoil (:234-239: u, chi)
__device__ inline V3 radiate_core(V3 &u, double chi, double gamma, const Sample &smp, double &formation_length, double &chi_new)
{
    V3 parallel = normalize(u);
    V3 perp = rotate_around(orthogonal(parallel), parallel, smp.phi);
    double st, ct;
    sincos(smp.theta, &st, &ct);
    V3 k = smp.omega_mc2 * (ct * parallel + st * perp);
    formation_length = 2.0 * (gamma * gamma) * smp.theta * SPEED_OF_LIGHT * COMPTON_TIME / chi;
    u = u - k;
    chi_new = chi * sqrt(1.0 + dot(u, u)) / gamma;
    return k;
}

__device__ void radiate_one(const RadiateArgs &a, size_t i)
{

    unsigned long long id;
    memcpy(&id, &a.id[i], sizeof id);
    double rnd[4];
    emission_randoms(a.seed, id, a.step, rnd);
    a.tau[i] = -log(1.0 - rnd[0]);  // Exp1, electron.rs:211

    V3 u = {a.ux[i], a.uy[i], a.uz[i]};
    const double chi = a.chi[i];
    const double gamma = sqrt(1.0 + dot(u, u));  // the reference keeps gamma in the struct (same value up to rounding)
    Sample smp;
    if (!quantum_sample(chi, gamma, rnd[1], rnd[2], rnd[3], smp)) {
        atomicOr(&a.ectr->err, 32);  // photon_emission.rs:143-144: chi >= 100 is unimplemented!() there
        return;
    }
    double formation_length, chi_new;
    V3 k = radiate_core(u, chi, gamma, smp, formation_length, chi_new);
    a.ux[i] = u.x; a.uy[i] = u.y; a.uz[i] = u.z;
    a.chi[i] = chi_new;

    // interactions.rs:73-98
    double k0 = sqrt(dot(k, k));
    bool keep = true;
    if (a.min_energy >= 0.0) keep = keep && k0 * ELECTRON_MASS_MEV >= a.min_energy;
    if (a.max_angle >= 0.0) {
        V3 p = ELECTRON_MASS_MEV * k;
        keep = keep && acos(-p.x / sqrt(dot(p, p))) <= a.max_angle;
    }
    if (a.max_formation_length >= 0.0) keep = keep && formation_length < a.max_formation_length;
    if (!keep) return;
    // One atomic per wave for the photons' slots (tens of thousands of photons a step all bump ONE counter: lane by lane that
    // was most of this kernel's time): the lanes that keep their photon are counted, the first of them reserves the run.
    const unsigned long long keepers = __ballot(1);
    const int lane = (int)(threadIdx.x & 63u), first = __ffsll((long long)keepers) - 1;
    unsigned base = 0;
    if (lane == first) base = atomicAdd(&a.pctr->n_slots, (unsigned)__popcll(keepers));
    base = __shfl(base, first);
    unsigned slot = base + (unsigned)__popcll(keepers & ((1ull << lane) - 1ull));
    if (slot >= a.pcap) {  // cannot happen: the host sized the photon arrays for every emitter
        atomicSub(&a.pctr->n_slots, 1u);
        atomicAdd(&a.pctr->n_lost, 1u);
        return;
    }
    // Photon::create at the electron's position (electron.rs:243), weight and all
    a.pcell[slot] = a.cell[i];
    a.px[slot] = a.x[i];
    a.pkx[slot] = k.x; a.pky[slot] = k.y; a.pkz[slot] = k.z;
    a.pw[slot] = a.w[i];
    if (a.pchi) a.pchi[slot] = 0.0;
    if (a.py) a.py[slot] = 0.0;
    if (a.pz) a.pz[slot] = 0.0;
    if (a.pid) memcpy(&a.pid[slot], &id, sizeof id);
    if (a.ptau_abs) {  // .with_optical_depths(rng).at_time(t), electron.rs:244-246
        double od[2];
        birth_randoms(a.seed, id, a.step, od);
        a.ptau_abs[slot] = -log(1.0 - od[0]);
        a.ptau_st[slot] = -log(1.0 - od[1]);
        a.pbirth[slot] = a.t;
    }
}

__global__ void k_sample_batch(int n, const double *chi, const double *gamma, const double *r1, const double *r2,
                               const double *r3, int classical, double *out, int *bad)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sample s = {0.0, 0.0, 0.0};
    if (classical) s = classical_sample(chi[i], gamma[i], r1[i], r2[i], r3[i]);
    else if (!quantum_sample(chi[i], gamma[i], r1[i], r2[i], r3[i], s)) atomicAdd(bad, 1);
    out[3 * i] = s.omega_mc2; out[3 * i + 1] = s.theta; out[3 * i + 2] = s.phi;
}

// the emission event with the three uniforms given (tests/test_gpu_qed.py against tests/golden/qed_radiate_independent.json)
__global__ void k_radiate_batch(int n, const double *chi, const double *ux, const double *uy, const double *uz, const double *r1,
                                const double *r2, const double *r3, double *out, int *bad)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 u = {ux[i], uy[i], uz[i]};
    const double gamma = sqrt(1.0 + dot(u, u));
    Sample smp = {0.0, 0.0, 0.0};
    if (!quantum_sample(chi[i], gamma, r1[i], r2[i], r3[i], smp)) {
        atomicAdd(bad, 1);
        return;
    }
    double fl, chi_new;
    V3 k = radiate_core(u, chi[i], gamma, smp, fl, chi_new);
    double *o = out + 9 * (size_t)i;
    o[0] = k.x; o[1] = k.y; o[2] = k.z; o[3] = u.x; o[4] = u.y; o[5] = u.z; o[6] = sqrt(1.0 + dot(u, u)); o[7] = chi_new; o[8] = fl;
}

__global__ void k_randoms_batch(int n, unsigned long long seed, const unsigned long long *id, unsigned step, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r[4];
    emission_randoms(seed, id[i], step, r);
    for (int k = 0; k < 4; k++) out[4 * i + k] = r[k];
}

__global__ void k_rate_batch(int n, const double *chi, const double *gamma, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = qed_rate(chi[i], gamma[i]);
}

}  // namespace

extern "C" {

int opal_emit_radiation(opal_pop_t *e, opal_pop_t *ph, uint64_t seed, uint32_t step, double t, double min_energy,
                        double max_angle, double max_formation_length, size_t *n_emitted)
{
    if (!e || !ph || e->species != OPAL_ELECTRON || ph->species != OPAL_PHOTON) {
        set_error("opal_emit_radiation: needs an electron and a photon population");
        return OPAL_ERR_ARG;
    }
    e->cold_yz = false;  // this pass rewrites the electrons' momenta (see opal_pop::cold_yz)
    if (e->extras != 2 || !e->d_ex[EX_TAU] || !e->d_ex[EX_ID] || !e->d_ex[EX_CHI]) {
        set_error("opal_emit_radiation: the electrons do not carry optical depths (opal_pop_track_extras level 2)");
        return OPAL_ERR_ARG;
    }
    if (e->stream != ph->stream) {
        set_error("opal_emit_radiation: both populations must live on the same stream");
        return OPAL_ERR_ARG;
    }
    if (n_emitted) *n_emitted = 0;
    if (e->n_upper == 0) return OPAL_OK;
    // How many electrons emit decides how much room the photons need.  When the caller wants the number
    // (n_emitted), count, read it back (one host synchronisation) and reserve exactly.  Otherwise nothing
    // is read back: room is reserved for the most that can emit (every electron; HBM is plentiful), the
    // second pass sizes itself on the device, and the host's bound on the photon slots in use is tightened
    // from counter snapshots that arrive a step or two later.
    int k = e->cur;
    if (!e->emit_list_valid) {
        // No list from the push (the caller moved particles or changed optical depths since, or asks a second
        // time in one step): scan for the emitters.  The time loop never comes here: its pushes leave the list.
        if (e->emit_list_cap < e->cap || !e->d_emit_list) {
            if (e->d_emit_list) hipFree(e->d_emit_list);
            e->d_emit_list = nullptr;
            OPAL_HIP(hipMalloc(&e->d_emit_list, sizeof(unsigned) * e->cap));
            e->emit_list_cap = e->cap;
        }
        OPAL_HIP(hipMemsetAsync(&e->d_ctr->n_emit, 0, sizeof(unsigned), e->stream));
        unsigned nblk = (unsigned)((e->n_upper + 256 * LIST_ITER - 1) / (256 * LIST_ITER));
        k_list_emitters<<<nblk, 256, 0, e->stream>>>(e->d_cell[k], e->d_ex[EX_TAU], e->d_ctr, e->d_emit_list);
        OPAL_HIP(hipGetLastError());
    }
    e->emit_list_valid = false;  // consumed below: every listed electron draws a new optical depth
    size_t emitters;
    if (n_emitted) {
        if (int rc = pop_check_err(e)) return rc;  // copies the counters to e->h_ctr and synchronises
        emitters = e->h_ctr->n_emit;
        if (emitters == 0) return OPAL_OK;
    } else {
        emitters = e->n_upper;
        if (int rc = pop_tighten_upper(ph)) return rc;
    }
    if (int rc = pop_reserve(ph, emitters)) return rc;

    RadiateArgs a;
    memset(&a, 0, sizeof a);
    a.cell = e->d_cell[k]; a.x = e->d_x[k]; a.w = e->d_w[k];
    a.ux = e->d_ux[k]; a.uy = e->d_uy[k]; a.uz = e->d_uz[k];
    a.chi = e->d_ex[EX_CHI]; a.tau = e->d_ex[EX_TAU]; a.id = e->d_ex[EX_ID];
    a.ectr = e->d_ctr;
    a.list = e->d_emit_list;
    int pk = ph->cur;
    a.pcell = ph->d_cell[pk]; a.px = ph->d_x[pk]; a.pkx = ph->d_ux[pk]; a.pky = ph->d_uy[pk]; a.pkz = ph->d_uz[pk];
    a.pw = ph->d_w[pk];
    if (ph->extras) {
        a.pchi = ph->d_ex[EX_CHI]; a.py = ph->d_ex[EX_Y]; a.pz = ph->d_ex[EX_Z]; a.pid = ph->d_ex[EX_ID];
        a.ptau_abs = ph->d_ex[EX_TAU_ABS]; a.ptau_st = ph->d_ex[EX_TAU_ST]; a.pbirth = ph->d_ex[EX_BIRTH];
    }
    a.t = t;
    a.pctr = ph->d_ctr;
    a.pcap = (unsigned)ph->cap;
    a.seed = seed; a.step = step;
    a.min_energy = min_energy; a.max_angle = max_angle; a.max_formation_length = max_formation_length;
    prof_begin(KERNEL_RADIATE, e->stream);
    k_radiate<<<(unsigned)std::min<size_t>((emitters + RADIATE_BLOCK - 1) / RADIATE_BLOCK, 2048 * (256 / RADIATE_BLOCK)), RADIATE_BLOCK, 0,
                e->stream>>>(a);
    prof_end(KERNEL_RADIATE, e->stream);
    OPAL_HIP(hipGetLastError());
    ph->sorted_exact = false;
    ph->n_upper += emitters;  // upper bound: the filters may have dropped some
    ph->added_since_pub += emitters;
    ph->prev_valid = false;   // Photon::create(.., dx = 0, dt = 0): prev_x is 0/0 until the first push
    if (n_emitted) {
        *n_emitted = emitters;
        ph->added_since_snapshot += emitters;  // a snapshot still in flight does not know about these
    } else if (ph->sort_interval < 0) {
        // under the adaptive re-binning policy the photons' counters reach the host every other step anyway (policy_step takes
        // the bound down with them): no copy of the slot count on the stream every step (round 5: a 4-byte copy and its event
        // between two 0.1 ms kernels)
        ph->added_since_snapshot += emitters;
    } else if (int rc = pop_snapshot_upper(ph, emitters)) {
        return rc;
    }
    return OPAL_OK;
}

/* Device evaluation of the sampler, the emission random numbers and the rate for n inputs (host arrays):
 * the oracle-parity hooks of tests/test_gpu_qed.py. */
int opal_qed_sample_batch(size_t n, const double *chi, const double *gamma, const double *r1, const double *r2,
                          const double *r3, int classical, double *out)
{
    if (!n) return OPAL_OK;
    double *d = nullptr;
    int *d_bad = nullptr;
    OPAL_HIP(hipMalloc(&d, sizeof(double) * n * 8));
    OPAL_HIP(hipMalloc(&d_bad, sizeof(int)));
    OPAL_HIP(hipMemset(d_bad, 0, sizeof(int)));
    const double *src[5] = {chi, gamma, r1, r2, r3};
    for (int k = 0; k < 5; k++) OPAL_HIP(hipMemcpy(d + k * n, src[k], sizeof(double) * n, hipMemcpyHostToDevice));
    k_sample_batch<<<(unsigned)((n + 127) / 128), 128>>>((int)n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, classical, d + 5 * n, d_bad);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpy(out, d + 5 * n, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    int bad = 0;
    OPAL_HIP(hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(d);
    hipFree(d_bad);
    if (bad) {
        set_error("opal_qed_sample_batch: %d inputs have chi >= 100, which the reference's sampler does not implement (photon_emission.rs:143-144)", bad);
        return OPAL_ERR_ASSERT;
    }
    return OPAL_OK;
}

/* live particles of a population without bringing its cell array to the host (opal_pop_len does, for exactness after any
 * operator): slots in use minus the slots vacated -- deleted at an open edge, absorbed, or sent to a neighbour -- since the last
 * re-binning, from the device's counters (one small synchronous copy).  bench.py counts photon and electron updates by it. */
int opal_pop_live(opal_pop_t *p, size_t *n)
{
    if (!p || !n) {
        set_error("opal_pop_live: null argument");
        return OPAL_ERR_ARG;
    }
    if (int rc = pop_check_err(p)) return rc;
    *n = (size_t)p->h_ctr->n_slots - (size_t)std::min(p->h_ctr->n_dead, p->h_ctr->n_slots);
    return OPAL_OK;
}

int opal_qed_radiate_batch(size_t n, const double *chi, const double *ux, const double *uy, const double *uz, const double *r1,
                           const double *r2, const double *r3, double *out)
{
    if (!n) return OPAL_OK;
    double *d = nullptr;
    int *d_bad = nullptr;
    OPAL_HIP(hipMalloc(&d, sizeof(double) * n * 16));
    OPAL_HIP(hipMalloc(&d_bad, sizeof(int)));
    OPAL_HIP(hipMemset(d_bad, 0, sizeof(int)));
    const double *src[7] = {chi, ux, uy, uz, r1, r2, r3};
    for (int k = 0; k < 7; k++) OPAL_HIP(hipMemcpy(d + k * n, src[k], sizeof(double) * n, hipMemcpyHostToDevice));
    k_radiate_batch<<<(unsigned)((n + 127) / 128), 128>>>((int)n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 5 * n, d + 6 * n,
                                                           d + 7 * n, d_bad);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpy(out, d + 7 * n, sizeof(double) * 9 * n, hipMemcpyDeviceToHost));
    int bad = 0;
    OPAL_HIP(hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(d);
    hipFree(d_bad);
    if (bad) {
        set_error("opal_qed_radiate_batch: %d inputs have chi >= 100, which the reference's sampler does not implement (photon_emission.rs:143-144)", bad);
        return OPAL_ERR_ASSERT;
    }
    return OPAL_OK;
}

int opal_qed_randoms_batch(size_t n, uint64_t seed, const uint64_t *id, uint32_t step, double *out)
{
    if (!n) return OPAL_OK;
    unsigned long long *d_id = nullptr;
    double *d_out = nullptr;
    OPAL_HIP(hipMalloc(&d_id, sizeof(uint64_t) * n));
    OPAL_HIP(hipMalloc(&d_out, sizeof(double) * 4 * n));
    OPAL_HIP(hipMemcpy(d_id, id, sizeof(uint64_t) * n, hipMemcpyHostToDevice));
    k_randoms_batch<<<(unsigned)((n + 127) / 128), 128>>>((int)n, seed, d_id, step, d_out);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpy(out, d_out, sizeof(double) * 4 * n, hipMemcpyDeviceToHost));
    hipFree(d_id);
    hipFree(d_out);
    return OPAL_OK;
}

int opal_qed_rate_batch(size_t n, const double *chi, const double *gamma, double *out)
{
    if (!n) return OPAL_OK;
    double *d = nullptr;
    OPAL_HIP(hipMalloc(&d, sizeof(double) * 3 * n));
    OPAL_HIP(hipMemcpy(d, chi, sizeof(double) * n, hipMemcpyHostToDevice));
    OPAL_HIP(hipMemcpy(d + n, gamma, sizeof(double) * n, hipMemcpyHostToDevice));
    k_rate_batch<<<(unsigned)((n + 127) / 128), 128>>>((int)n, d, d + n, d + 2 * n);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpy(out, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    hipFree(d);
    return OPAL_OK;
}

}  // extern "C"
