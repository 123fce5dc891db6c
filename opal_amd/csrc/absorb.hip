// One-photon absorption and stimulated emission on the GPU: absorb (particle/interactions.rs:145-340) =
// Photon::interacts_with (photon.rs:312-362) of every photon with the electrons of its own cell, with the scaled
// cross sections of qed/photon_absorption.rs:16-34 and qed/stimulated_emission.rs:16-37 and the Airy function of
// qed/special_functions/airy.rs:19-69.
//
// GPU shape.  The reference walks two cell-sorted vectors; here the electrons are re-binned by cell first (the
// counting sort leaves cell_end[c], one past the last slot of cell c), and one thread per photon walks the
// electrons of that photon's cell in slot order, running down the photon's two optical depths and stopping at
// the first that falls below zero -- the photons' own order is irrelevant.  Electrons are read-only in that pass
// (the reference applies the kicks afterwards, interactions.rs:318-332); a second pass removes the absorbed
// photons, appends the stimulated ones and kicks the electrons with f64 atomics (an electron may be hit by
// several photons in a step, as in the reference).  Between the passes the host reads the two event counts, to
// make room for the new photons: this mode re-bins the electrons every step anyway.
// Random numbers: counter-based, keyed by the photon (parent id, birth time) and the step, so oracle and device
// take identical decisions whatever the layout.
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
int pop_sort_exact(opal_pop_t *p, const opal_grid_t *g);
int pop_tighten_upper(opal_pop_t *p);
int pop_snapshot_upper(opal_pop_t *p, size_t just_added);
}

namespace {

#define AIRY_TABLE_QUAL __device__ const
#include "airy_tables.inc"

// the arithmetic below follows the reference's operation order; no contraction, so that the device differs from
// the CPU restatement only by the last ulp of pow / exp / log
#pragma clang fp contract(off)

constexpr double CLASSICAL_ELECTRON_RADIUS = 2.817940e-15;  // constants.rs:28
constexpr double PHOTON_E_ECRIT_CUTOFF = 1.0e-8;            // interactions.rs:146

// llvm.powi: square-and-multiply, as compiler-rt's __powidf2
__device__ inline double powi_f64(double a, int b)
{
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}

__device__ inline double airy_quadrature(double x, const double *nodes, const double *weights, int n)
{
    const double s = 2.0 * pow(x, 1.5) / 3.0;
    const double a = AIRY_SCALE * pow(s, -1.0 / 6.0) * exp(-s);
    double integral = 0.0;
    for (int i = 0; i < n; i++) integral += weights[i] * pow(2.0 + nodes[i] / s, -1.0 / 6.0);
    return a * integral;
}

// airy_ai_for_positive, airy.rs:19-69; false for None (x < 0 or x >= 50)
__device__ bool airy_ai(double x, double &out)
{
    if (x < 0.0) return false;
    if (x < 1.0) {
        double sum = 0.0;
        for (int i = 0; i < AIRY_SERIES_USED; i++) sum += AIRY_SERIES_A[i] * powi_f64(x, AIRY_SERIES_N[i]);
        out = sum;
    } else if (x < 2.0) {
        out = airy_quadrature(x, AIRY_GL40_X, AIRY_GL40_W, 40);
    } else if (x < 10.0) {
        out = airy_quadrature(x, AIRY_GL16_X, AIRY_GL16_W, 16);
    } else if (x < 50.0) {
        out = airy_quadrature(x, AIRY_GL4_X, AIRY_GL4_W, 4);
    } else {
        return false;
    }
    return true;
}

// photon_absorption.rs:16-34 (stimulated = false) / stimulated_emission.rs:16-37 (true); false for None
__device__ bool scaled_cross_section(const double (&k)[4], const double (&p)[4], double chi_gamma, double chi_e,
                                     bool stimulated, double &out)
{
    if (stimulated) {
        if (chi_gamma >= chi_e || k[0] >= p[0] || chi_e <= 0.0 || chi_gamma <= 0.0) return false;
    } else if (chi_e <= 0.0 || chi_gamma <= 0.0) {
        return false;
    }
    const double denom = stimulated ? chi_e * (chi_e - chi_gamma) : chi_e * (chi_e + chi_gamma);
    const double g = 0.5 + 0.25 * (chi_gamma * chi_gamma) / denom;
    const double z = pow(chi_gamma / denom, 2.0 / 3.0);
    const double k_p = k[0] * p[0] - k[1] * p[1] - k[2] * p[2] - k[3] * p[3];
    const double zbar = 2.0 * z * chi_e * k_p / chi_gamma;
    const double zbar_z = 2.0 * p[0] * k_p / k[0];
    double ai;
    if (!airy_ai(zbar, ai)) return false;
    const double pre = 2.0 * M_PI * CLASSICAL_ELECTRON_RADIUS;
    out = (pre * pre) * chi_e * z * (4.0 * g * zbar_z - 1.0) * ai / (ALPHA_FINE * chi_gamma * k[0] * p[0]);
    return true;
}

enum { EVENT_NONE = 0, EVENT_ABSORBED = 1, EVENT_STIMULATED = 2 };

struct AbsorbArgs {
    // electrons, binned by cell: cell_end[c] = one past the last slot of cell c
    const unsigned *cell_end;
    int ncells;
    double *eux, *euy, *euz;
    const double *ew, *echi, *eid;
    // photons
    int *pcell;
    double *px, *pkx, *pky, *pkz, *pw;
    double *pchi, *py, *pz, *pid, *ptau_abs, *ptau_st, *pbirth;
    PopCounters *pctr;
    unsigned pcap;
    // per photon slot: what happened and with which electron
    unsigned char *event;
    unsigned *partner;
    unsigned *counts;  // [0] absorbed, [1] stimulated, [2] reference asserts tripped
    unsigned long long seed;
    unsigned step;
    double t, dt_factor;  // c dt / dx
    double max_displacement, stop_time;  // < 0: None
};

// pass 1: Photon::interacts_with against the electrons of the photon's cell (interactions.rs:171-262)
__global__ __launch_bounds__(256) void k_absorb_decide(AbsorbArgs a)
{
    const unsigned n = a.pctr->n_slots;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) a.counts[3] = n;  // for an apply pass that sizes itself on the device
    if (i >= n) return;
    a.event[i] = EVENT_NONE;
    const int cell = a.pcell[i];
    if (cell == DEAD_CELL || cell < 0 || cell >= a.ncells) return;
    const double kx = a.pkx[i], ky = a.pky[i], kz = a.pkz[i];
    const double k0 = sqrt(kx * kx + ky * ky + kz * kz);
    const double chi_gamma = a.pchi[i];
    if (chi_gamma * ELECTRON_MASS_MEV / (k0 * ELECTRON_MASS_MEV) < PHOTON_E_ECRIT_CUTOFF) return;  // :184
    const double birth = a.pbirth[i];
    if (a.stop_time >= 0.0 && a.t - birth > a.stop_time) return;                      // :188-192
    if (a.max_displacement >= 0.0 && hypot(a.py[i], a.pz[i]) > a.max_displacement) return;  // :195-199
    const unsigned start = cell ? a.cell_end[cell - 1] : 0u, end = a.cell_end[cell];
    if (start == end) return;
    unsigned long long id;
    memcpy(&id, &a.pid[i], sizeof id);
    double rnd[3];
    absorption_randoms(a.seed, id, birth, a.step, rnd);
    const double k[4] = {k0, kx, ky, kz};
    double tau_abs = a.ptau_abs[i], tau_st = a.ptau_st[i];
    int state = EVENT_NONE;
    unsigned who = 0;
    for (unsigned j = start; j < end; j++) {
        const double ux = a.eux[j], uy = a.euy[j], uz = a.euz[j];
        // the reference reads the electron's stored gamma, sqrt(1 + u^2) as of its last push / recoil / kick
        const double p[4] = {sqrt(1.0 + (ux * ux + uy * uy + uz * uz)), ux, uy, uz};
        const double chi_e = a.echi[j], we = a.ew[j];
        double prob_abs = 0.0, prob_st = 0.0, sigma;
        if (scaled_cross_section(k, p, chi_gamma, chi_e, false, sigma)) {
            const double partial = we * a.dt_factor * sigma;
            if (!(partial >= 0.0)) { atomicAdd(&a.counts[2], 1u); atomicOr(&a.pctr->err, 64); break; }  // photon.rs:325 assert
            tau_abs -= partial;
            prob_abs = partial;
        }
        if (scaled_cross_section(k, p, chi_gamma, chi_e, true, sigma)) {
            const double partial = we * a.dt_factor * sigma;
            if (!(partial >= 0.0)) { atomicAdd(&a.counts[2], 1u); atomicOr(&a.pctr->err, 64); break; }  // photon.rs:336 assert
            tau_st -= partial;
            prob_st = partial;
        }
        if (tau_abs < 0.0 && tau_st < 0.0) {
            if (rnd[0] < prob_abs / (prob_abs + prob_st)) {
                state = EVENT_ABSORBED;
            } else {
                tau_abs = -log(1.0 - rnd[1]);
                tau_st = -log(1.0 - rnd[2]);
                state = EVENT_STIMULATED;
            }
        } else if (tau_abs < 0.0) {
            state = EVENT_ABSORBED;
        } else if (tau_st < 0.0) {
            tau_st = -log(1.0 - rnd[2]);
            state = EVENT_STIMULATED;
        }
        if (state != EVENT_NONE) {
            who = j;
            break;
        }
    }
    a.ptau_abs[i] = tau_abs;
    a.ptau_st[i] = tau_st;
    if (state != EVENT_NONE) {
        a.event[i] = (unsigned char)state;
        a.partner[i] = who;
        atomicAdd(&a.counts[state == EVENT_ABSORBED ? 0 : 1], 1u);
    }
}

// pass 2: interactions.rs:290-332 -- absorbed photons go, stimulated ones are appended (copy of the stimulating
// photon with the electron's weight, fresh optical depths, born now), the electrons are kicked
// (n_before: the photon slots in use when the decide pass ran -- the host's count, or, when it did not read one back,
// what the decide pass left in counts[3])
__global__ __launch_bounds__(256) void k_absorb_apply(AbsorbArgs a, unsigned n_before)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_before == 0xffffffffu) n_before = a.counts[3];
    if (i >= n_before) return;
    const int state = a.event[i];
    if (state == EVENT_NONE) return;
    const unsigned j = a.partner[i];
    const double kx = a.pkx[i], ky = a.pky[i], kz = a.pkz[i];
    const double we = a.ew[j];
    // Electron::kick (electron.rs:256-262): momentum() is k in MeV, divided by the electron mass in MeV again
    const double scale = state == EVENT_ABSORBED ? a.pw[i] / we : we / we;
    const double sgn = state == EVENT_ABSORBED ? 1.0 : -1.0;
    atomicAdd(&a.eux[j], scale * (sgn * (kx * ELECTRON_MASS_MEV)) / ELECTRON_MASS_MEV);
    atomicAdd(&a.euy[j], scale * (sgn * (ky * ELECTRON_MASS_MEV)) / ELECTRON_MASS_MEV);
    atomicAdd(&a.euz[j], scale * (sgn * (kz * ELECTRON_MASS_MEV)) / ELECTRON_MASS_MEV);
    if (state == EVENT_ABSORBED) {
        a.pcell[i] = DEAD_CELL;
        atomicAdd(&a.pctr->n_dead, 1u);
        return;
    }
    const unsigned slot = atomicAdd(&a.pctr->n_slots, 1u);
    if (slot >= a.pcap) {  // cannot happen: the host made room for every stimulated emission it counted
        atomicSub(&a.pctr->n_slots, 1u);
        atomicAdd(&a.pctr->n_lost, 1u);
        return;
    }
    unsigned long long pid;
    memcpy(&pid, &a.pid[i], sizeof pid);
    a.pcell[slot] = a.pcell[i];
    a.px[slot] = a.px[i];
    a.pkx[slot] = kx; a.pky[slot] = ky; a.pkz[slot] = kz;
    a.pw[slot] = we;                       // photon.with_weight(e.weight()), interactions.rs:243
    a.pchi[slot] = a.pchi[i];
    a.py[slot] = a.py[i];
    a.pz[slot] = a.pz[i];
    double od[2];
    uint64_t nid;
    stimulated_birth(a.seed, pid, a.pbirth[i], a.step, nid, od);  // .with_optical_depths(rng).at_time(t), :311
    a.ptau_abs[slot] = -log(1.0 - od[0]);
    a.ptau_st[slot] = -log(1.0 - od[1]);
    a.pbirth[slot] = a.t;
    memcpy(&a.pid[slot], &nid, sizeof nid);
}

__global__ void k_airy_batch(int n, const double *x, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v;
    out[i] = airy_ai(x[i], v) ? v : NAN;
}

__global__ void k_cross_sections_batch(int n, const double *k, const double *p, const double *chi_gamma, const double *chi_e,
                                       double *out_abs, double *out_st)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double kk[4] = {k[4 * i], k[4 * i + 1], k[4 * i + 2], k[4 * i + 3]};
    const double pp[4] = {p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]};
    double v;
    out_abs[i] = scaled_cross_section(kk, pp, chi_gamma[i], chi_e[i], false, v) ? v : NAN;
    out_st[i] = scaled_cross_section(kk, pp, chi_gamma[i], chi_e[i], true, v) ? v : NAN;
}

// live values of one extras array in slot order (dead slots skipped)
int pull_live(const opal_pop_t *p, const double *src, double *dst, const std::vector<int> &hc)
{
    if (!dst) return OPAL_OK;
    std::vector<double> tmp(hc.size());
    OPAL_HIP(hipMemcpy(tmp.data(), src, sizeof(double) * hc.size(), hipMemcpyDeviceToHost));
    size_t o = 0;
    for (size_t i = 0; i < hc.size(); i++)
        if (hc[i] != DEAD_CELL) dst[o++] = tmp[i];
    return OPAL_OK;
}

}  // namespace

extern "C" {

int opal_absorb(opal_pop_t *e, opal_pop_t *ph, const opal_grid_t *grid, uint64_t seed, uint32_t step, double t, double dt,
                double max_displacement, double stop_time, size_t *n_absorbed, size_t *n_stimulated)
{
    if (n_absorbed) *n_absorbed = 0;
    if (n_stimulated) *n_stimulated = 0;
    if (!e || !ph || !grid || e->species != OPAL_ELECTRON || ph->species != OPAL_PHOTON) {
        set_error("opal_absorb: needs an electron population, a photon population and their grid");
        return OPAL_ERR_ARG;
    }
    e->cold_yz = false;  // this pass rewrites the electrons' momenta (see opal_pop::cold_yz)
    if (!e->extras || !e->d_ex[EX_CHI] || !e->d_ex[EX_ID]) {
        set_error("opal_absorb: the electrons do not carry chi and ids (opal_pop_track_fields)");
        return OPAL_ERR_ARG;
    }
    if (!ph->extras || !ph->d_ex[EX_CHI] || !ph->d_ex[EX_Y] || !ph->d_ex[EX_Z] || !ph->d_ex[EX_ID] || !ph->d_ex[EX_TAU_ABS] ||
        !ph->d_ex[EX_TAU_ST] || !ph->d_ex[EX_BIRTH]) {
        set_error("opal_absorb: the photons do not carry chi, y, z, ids, optical depths and birth times "
                  "(opal_pop_track_fields with OPAL_TRACK_CHI | OPAL_TRACK_YZ | OPAL_TRACK_ID | OPAL_TRACK_ABSORPTION)");
        return OPAL_ERR_ARG;
    }
    if (e->stream != ph->stream) {
        set_error("opal_absorb: both populations must live on the same stream");
        return OPAL_ERR_ARG;
    }
    if (e->n_upper == 0 || ph->n_upper == 0) return OPAL_OK;  // interactions.rs:148-150
    // electrons exactly binned by cell; the sort's cursors are the cell ends
    if (int rc = pop_sort_exact(e, grid)) return rc;
    if (e->n_upper == 0) return OPAL_OK;
    // The caller that wants the event counts gets them (one read-back); the time loop does not ask, and then
    // nothing is read back at all: the passes size themselves on the device from the photons' slot counter, room is
    // reserved for the most that can happen (every photon stimulates one emission; HBM is plentiful), the host's bound
    // on the photon slots in use is tightened from counter snapshots that arrive a step or two later (as
    // opal_emit_radiation does), and a tripped reference assert is reported by the next status check.
    const bool readback = n_absorbed || n_stimulated;
    unsigned nph;
    if (readback) {
        if (int rc = pop_check_err(ph)) return rc;  // also brings the photons' slot count to the host
        nph = ph->h_ctr->n_slots;
        ph->n_upper = nph;
    } else {
        if (int rc = pop_tighten_upper(ph)) return rc;
        nph = (unsigned)ph->n_upper;
        if (int rc = pop_reserve(ph, nph)) return rc;
    }
    if (nph == 0) return OPAL_OK;

    // scratch that lives with the photon population: what happened to each photon slot and with which electron
    if (ph->abs_cap < nph) {
        if (ph->d_abs_event) hipFree(ph->d_abs_event);
        if (ph->d_abs_partner) hipFree(ph->d_abs_partner);
        ph->d_abs_event = nullptr; ph->d_abs_partner = nullptr;
        const size_t cap = std::max<size_t>(ph->cap, nph);
        OPAL_HIP(hipMalloc(&ph->d_abs_event, cap));
        OPAL_HIP(hipMalloc(&ph->d_abs_partner, sizeof(unsigned) * cap));
        ph->abs_cap = cap;
    }
    if (!ph->d_abs_counts) OPAL_HIP(hipMalloc(&ph->d_abs_counts, sizeof(unsigned) * 4));
    unsigned char *d_event = ph->d_abs_event;
    unsigned *d_partner = ph->d_abs_partner, *d_counts = ph->d_abs_counts;
    OPAL_HIP(hipMemsetAsync(d_counts, 0, sizeof(unsigned) * 4, e->stream));

    AbsorbArgs a;
    memset(&a, 0, sizeof a);
    const int ek = e->cur;
    a.cell_end = e->d_count;
    a.ncells = (int)grid->size;
    a.eux = e->d_ux[ek]; a.euy = e->d_uy[ek]; a.euz = e->d_uz[ek];
    a.ew = e->d_w[ek]; a.echi = e->d_ex[EX_CHI]; a.eid = e->d_ex[EX_ID];
    auto bind_photons = [&]() {
        const int pk = ph->cur;
        a.pcell = ph->d_cell[pk]; a.px = ph->d_x[pk]; a.pkx = ph->d_ux[pk]; a.pky = ph->d_uy[pk]; a.pkz = ph->d_uz[pk];
        a.pw = ph->d_w[pk];
        a.pchi = ph->d_ex[EX_CHI]; a.py = ph->d_ex[EX_Y]; a.pz = ph->d_ex[EX_Z]; a.pid = ph->d_ex[EX_ID];
        a.ptau_abs = ph->d_ex[EX_TAU_ABS]; a.ptau_st = ph->d_ex[EX_TAU_ST]; a.pbirth = ph->d_ex[EX_BIRTH];
        a.pctr = ph->d_ctr;
        a.pcap = (unsigned)ph->cap;
    };
    bind_photons();
    a.event = d_event; a.partner = d_partner; a.counts = d_counts;
    a.seed = seed; a.step = step;
    a.t = t;
    a.dt_factor = SPEED_OF_LIGHT * dt / grid->dx;
    a.max_displacement = max_displacement; a.stop_time = stop_time;
    k_absorb_decide<<<(nph + 255) / 256, 256, 0, e->stream>>>(a);
    OPAL_HIP(hipGetLastError());
    if (!readback) {
        k_absorb_apply<<<(nph + 255) / 256, 256, 0, e->stream>>>(a, 0xffffffffu);
        OPAL_HIP(hipGetLastError());
        ph->n_upper += nph;  // upper bound: one stimulated emission per photon at the most
        ph->added_since_pub += nph;
        ph->sorted_exact = false;
        ph->emit_list_valid = false;
        e->emit_list_valid = false;
        return pop_snapshot_upper(ph, nph);
    }
    unsigned counts[4] = {0, 0, 0, 0};
    OPAL_HIP(hipMemcpyAsync(counts, d_counts, sizeof counts, hipMemcpyDeviceToHost, e->stream));
    OPAL_HIP(hipStreamSynchronize(e->stream));
    if (counts[2]) {
        OPAL_HIP(hipMemsetAsync(&ph->d_ctr->err, 0, sizeof(int), e->stream));
        set_error("%u photon-electron pairs gave a negative or non-finite interaction probability "
                  "(photon.rs:325,336 assert!(partial_prob >= 0.0))", counts[2]);
        return OPAL_ERR_ASSERT;
    }
    if (counts[0] || counts[1]) {
        if (counts[1]) {
            if (int rc = pop_reserve(ph, counts[1])) return rc;  // may move the photon arrays
            bind_photons();
        }
        k_absorb_apply<<<(nph + 255) / 256, 256, 0, e->stream>>>(a, nph);
        OPAL_HIP(hipGetLastError());
        OPAL_HIP(hipStreamSynchronize(e->stream));
        ph->n_upper = (size_t)nph + counts[1];
        ph->sorted_exact = false;
        ph->emit_list_valid = false;
        e->emit_list_valid = false;  // (the re-binning above already said so)
    }
    if (n_absorbed) *n_absorbed = counts[0];
    if (n_stimulated) *n_stimulated = counts[1];
    return OPAL_OK;
}

int opal_pop_download_absorption_state(const opal_pop_t *p, double *tau_abs, double *tau_st, double *birth_time)
{
    if (!p || !p->d_ex[EX_TAU_ABS] || !p->d_ex[EX_TAU_ST] || !p->d_ex[EX_BIRTH]) {
        set_error("opal_pop_download_absorption_state: not tracked (photons with OPAL_TRACK_ABSORPTION)");
        return OPAL_ERR_ARG;
    }
    if (int rc = pop_check_err(const_cast<opal_pop_t *>(p))) return rc;
    const size_t slots = p->h_ctr->n_slots;
    if (!slots) return OPAL_OK;
    std::vector<int> hc(slots);
    OPAL_HIP(hipMemcpy(hc.data(), p->d_cell[p->cur], sizeof(int) * slots, hipMemcpyDeviceToHost));
    if (int rc = pull_live(p, p->d_ex[EX_TAU_ABS], tau_abs, hc)) return rc;
    if (int rc = pull_live(p, p->d_ex[EX_TAU_ST], tau_st, hc)) return rc;
    return pull_live(p, p->d_ex[EX_BIRTH], birth_time, hc);
}

int opal_pop_upload_absorption_state(opal_pop_t *p, size_t n, const double *tau_abs, const double *tau_st,
                                     const double *birth_time)
{
    if (!p || !p->d_ex[EX_TAU_ABS] || !p->d_ex[EX_TAU_ST] || !p->d_ex[EX_BIRTH] || n > p->cap) {
        set_error("opal_pop_upload_absorption_state: not tracked (photons with OPAL_TRACK_ABSORPTION), or n exceeds the capacity");
        return OPAL_ERR_ARG;
    }
    if (!n) return OPAL_OK;
    if (tau_abs) OPAL_HIP(hipMemcpyAsync(p->d_ex[EX_TAU_ABS], tau_abs, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    if (tau_st) OPAL_HIP(hipMemcpyAsync(p->d_ex[EX_TAU_ST], tau_st, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    if (birth_time) OPAL_HIP(hipMemcpyAsync(p->d_ex[EX_BIRTH], birth_time, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    OPAL_HIP(hipStreamSynchronize(p->stream));
    return OPAL_OK;
}

int opal_pop_upload_chi(opal_pop_t *p, size_t n, const double *chi)
{
    if (!p || !chi || !p->d_ex[EX_CHI] || n > p->cap) {
        set_error("opal_pop_upload_chi: chi is not tracked, or n exceeds the capacity");
        return OPAL_ERR_ARG;
    }
    if (n) OPAL_HIP(hipMemcpy(p->d_ex[EX_CHI], chi, sizeof(double) * n, hipMemcpyHostToDevice));
    return OPAL_OK;
}

int opal_pop_upload_yz(opal_pop_t *p, size_t n, const double *y, const double *z)
{
    if (!p || !y || !z || !p->d_ex[EX_Y] || !p->d_ex[EX_Z] || n > p->cap) {
        set_error("opal_pop_upload_yz: y, z are not tracked, or n exceeds the capacity");
        return OPAL_ERR_ARG;
    }
    if (n) {
        OPAL_HIP(hipMemcpy(p->d_ex[EX_Y], y, sizeof(double) * n, hipMemcpyHostToDevice));
        OPAL_HIP(hipMemcpy(p->d_ex[EX_Z], z, sizeof(double) * n, hipMemcpyHostToDevice));
    }
    return OPAL_OK;
}

int opal_qed_airy_batch(size_t n, const double *x, double *out)
{
    if (!n) return OPAL_OK;
    double *d = nullptr;
    OPAL_HIP(hipMalloc(&d, sizeof(double) * 2 * n));
    OPAL_HIP(hipMemcpy(d, x, sizeof(double) * n, hipMemcpyHostToDevice));
    k_airy_batch<<<(unsigned)((n + 127) / 128), 128>>>((int)n, d, d + n);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpy(out, d + n, sizeof(double) * n, hipMemcpyDeviceToHost));
    hipFree(d);
    return OPAL_OK;
}

int opal_qed_cross_sections_batch(size_t n, const double *k, const double *p, const double *chi_gamma, const double *chi_e,
                                  double *out_absorption, double *out_stimulated)
{
    if (!n) return OPAL_OK;
    double *d = nullptr;
    OPAL_HIP(hipMalloc(&d, sizeof(double) * 12 * n));
    OPAL_HIP(hipMemcpy(d, k, sizeof(double) * 4 * n, hipMemcpyHostToDevice));
    OPAL_HIP(hipMemcpy(d + 4 * n, p, sizeof(double) * 4 * n, hipMemcpyHostToDevice));
    OPAL_HIP(hipMemcpy(d + 8 * n, chi_gamma, sizeof(double) * n, hipMemcpyHostToDevice));
    OPAL_HIP(hipMemcpy(d + 9 * n, chi_e, sizeof(double) * n, hipMemcpyHostToDevice));
    k_cross_sections_batch<<<(unsigned)((n + 127) / 128), 128>>>((int)n, d, d + 4 * n, d + 8 * n, d + 9 * n, d + 10 * n, d + 11 * n);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpy(out_absorption, d + 10 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    OPAL_HIP(hipMemcpy(out_stimulated, d + 11 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    hipFree(d);
    return OPAL_OK;
}

}  // extern "C"
