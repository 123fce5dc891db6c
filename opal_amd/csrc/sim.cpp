// The YAML-driven driver: main() of the reference (src/main.rs:48-286), photon
// emission included (absorption is out of scope), written against this library's own C ABI
// exactly as a Rust host would call it.  Also the C-ABI wrappers of the
// host-side pieces (GridDesign, loader, Config).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "config.hpp"
#include "grid_device.hpp"
#include "host_setup.hpp"
#include "link.hpp"
#include "opal_internal.hpp"

using namespace opal;

namespace opal {
int pop_check_err(opal_pop_t *p);
int pop_push_fused_noexchange(opal_pop_t *p, opal_link_t *link, opal_grid_t *g, double dt);
int pop_push_advance_noexchange(opal_pop_t *p, opal_link_t *link, opal_grid_t *g, double dt);
int pops_unpack_and_maybe_sort(opal_pop_t *const *pops, int npops, opal_link_t *link, const opal_grid_t *g,
                               const SyncArgs *sync, const AdvArgs *adv, bool peer_exchange);
int pops_unpack(opal_pop_t *const *pops, int npops, opal_link_t *link, const opal_grid_t *g);
int pops_push_fused_pair_noexchange(opal_pop_t *pe, opal_pop_t *pi, opal_link_t *link, opal_grid_t *g, double dt);
int pop_maybe_sort(opal_pop_t *p, const opal_grid_t *g);
int pops_maybe_sort(opal_pop_t *const *pops, int npops, const opal_grid_t *g);
int pops_policy(opal_pop_t *const *pops, int npops, PublishBatch *pub, bool *sort_now);
int pops_sort_decided(opal_pop_t *const *pops, int npops, const bool *sort_now, const opal_grid_t *g);
int pops_advance_pair_noexchange(opal_pop_t *pe, opal_pop_t *pi, opal_grid_t *g, double dt);
int pops_deposit_pair(opal_pop_t *pe, opal_pop_t *pi, opal_grid_t *g, double dt);
int grid_exchange_halo(opal_grid_t *g, opal_link_t *link, bool exchange, const double **recv_left, const double **recv_right,
                       hipStream_t st);
bool pops_plan_split(opal_pop_t *pe, opal_pop_t *pi, opal_link_t *link, const opal_grid_t *g);
int pops_push_fused_pair_part(opal_pop_t *pe, opal_pop_t *pi, opal_link_t *link, opal_grid_t *g, double dt, bool edges,
                              hipStream_t on);
int grid_apply_sync(opal_grid_t *g, const double *recv_left, const double *recv_right, double ly, double lz);
// a re-cut of the decomposition (sim_recut): the pieces in particles.hip and grid.hip
int pop_live_count(opal_pop_t *p, size_t *n);
int pop_cell_histogram(opal_pop_t *p, const opal_grid_t *g, std::vector<unsigned> &counts);
bool pop_live_estimate(const opal_pop_t *p, size_t *n);
int pop_debug_check_cells(opal_pop_t *p, const opal_grid_t *g, const char *when, size_t step);
void pops_reframe_begin(opal_pop_t *const *pops, int npops, int shift, size_t new_size);
void pops_reframe_end(opal_pop_t *const *pops, int npops);
int pops_recut_pack(opal_pop_t *const *pops, int npops, opal_link_t *link, size_t size, int give_left, int give_right, int delta);
int pops_recut_finish(opal_pop_t *const *pops, int npops, opal_link_t *link, const opal_grid_t *g);
int grid_recut(opal_grid_t *g, size_t size, double offset, const double *rows);
int grid_shift(opal_grid_t *g, int dL, size_t new_size, double new_offset);
int pops_sort_deferred(opal_pop_t *const *pops, int npops, const opal_grid_t *g);
const double *grid_fetch_rows(const opal_grid_t *g);
}

namespace {

// ---- Rust-compatible float formatting (main.rs:35-38 uses {:.6e}; yee.rs:768 uses {:e} and {})
void shortest_digits(double v, std::string &digits, int &exp10)
{
    // shortest decimal that round-trips; digits without dot, value = 0.d1d2... * 10^(exp10+1)
    char buf[64];
    for (int p = 0; p < 17; p++) {
        snprintf(buf, sizeof buf, "%.*e", p, v);
        if (strtod(buf, nullptr) == v) break;
    }
    std::string s(buf);
    size_t epos = s.find('e');
    std::string mant = s.substr(0, epos);
    exp10 = atoi(s.c_str() + epos + 1);
    digits.clear();
    for (char c : mant)
        if (isdigit((unsigned char)c)) digits.push_back(c);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
}

std::string rust_lower_exp(double v)  // {:e}
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    std::string sign = std::signbit(v) ? "-" : "";
    v = std::fabs(v);
    if (v == 0.0) return sign + "0e0";
    std::string d;
    int e;
    shortest_digits(v, d, e);
    std::string out = sign + d.substr(0, 1);
    if (d.size() > 1) out += "." + d.substr(1);
    return out + "e" + std::to_string(e);
}

std::string rust_lower_exp_prec(double v, int prec)  // {:.6e}
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    snprintf(buf, sizeof buf, "%.*e", prec, v);
    std::string s(buf);
    size_t epos = s.find('e');
    int e = atoi(s.c_str() + epos + 1);
    return s.substr(0, epos) + "e" + std::to_string(e);
}

std::string rust_display(double v)  // {}
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    std::string sign = std::signbit(v) ? "-" : "";
    v = std::fabs(v);
    if (v == 0.0) return sign + "0";
    std::string d;
    int e;
    shortest_digits(v, d, e);
    int n = (int)d.size();
    std::string out;
    if (e >= n - 1) {
        out = d + std::string((size_t)(e - (n - 1)), '0');
    } else if (e >= 0) {
        out = d.substr(0, (size_t)e + 1) + "." + d.substr((size_t)e + 1);
    } else {
        out = "0." + std::string((size_t)(-e - 1), '0') + d;
    }
    return sign + out;
}

// SimulationTime, setup.rs:421-438
std::string simulation_time(double t)
{
    double power = 3.0 * std::floor((std::log10(std::fabs(t)) + 0.0) / 3.0);
    power = std::max(std::min(power, 0.0), -18.0);
    int p = std::isnan(power) ? 0 : (int)power;
    const char *unit = " s";
    double scale = 1.0;
    switch (p) {
    case -18: unit = "as"; scale = 1.0e18; break;
    case -15: unit = "fs"; scale = 1.0e15; break;
    case -12: unit = "ps"; scale = 1.0e12; break;
    case -9: unit = "ns"; scale = 1.0e9; break;
    case -6: unit = "\xce\xbcs"; scale = 1.0e6; break;
    case -3: unit = "ms"; scale = 1.0e3; break;
    default: break;
    }
    char buf[64];
    snprintf(buf, sizeof buf, "%8.2f %s", scale * t, unit);
    return buf;
}

int config_fail(const ConfigError &e)
{
    set_error("%s", e.message().c_str());
    return OPAL_ERR_CONFIG;
}

double call_laser(double t, double x, void *user)
{
    auto *f = static_cast<std::function<double(const double *)> *>(user);
    double a[2] = {t, x};
    return (*f)(a);
}

}  // namespace

struct opal_sim {
    Config cfg;
    int rank = 0, numtasks = 1;
    opal_link_t *link = nullptr;
    void *stream = nullptr;
    std::string output_dir;
    size_t nx = 0, n_outputs = 0;
    double xmin = 0, dx = 0, dt = 0, tstart = 0, tend = 0;
    bool current_deposition = false, balance = true, has_laser = false;
    std::function<double(const double *)> laser_y, laser_z;
    opal_grid_t *grid = nullptr;
    opal_pop_t *electrons = nullptr, *ions = nullptr, *photons = nullptr;
    std::string ion_name;
    std::vector<std::string> electron_output, ion_output, photon_output;  // the `output:` lists (main.rs:117,134,156)
    // qed section (main.rs:78-83); limits < 0 = None
    bool photon_emission = false, photon_absorption = false;
    double absorb_max_displacement = -1.0, absorb_stop_time = -1.0;  // main.rs:84-85,247 (see sim_setup)
    double photon_energy_min = -1.0, photon_angle_max = -1.0, max_formation_length = -1.0;
    uint64_t emission_seed = 0;
    size_t step_index = 0;
    // overlapping the neighbour exchange with the bulk of the push (fused linked step)
    bool overlap = false;  // off by default: measured slower than push-then-exchange in the one-GPU rehearsal
    size_t n_overlapped = 0;  // steps that took the overlapped path (diagnostic)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_edges = nullptr, ev_comm = nullptr;
    std::vector<size_t> nxs;  // interior cells of every subdomain (GridDesign; moves with the cuts: sim_recut)
    // cuts that follow the plasma (opal_sim_set_recut; 0 = the reference's partition, made once at t = 0)
    size_t recut_every = 0, recut_max_shift = 64, recut_max_particles = (size_t)1 << 24;
    double recut_threshold = 1.10;
    size_t recut_checks = 0, recuts = 0;
    double recut_seconds = 0.0;     // host time spent in checks and re-cuts so far (what the step loop waited for them)
    double imbalance_seen = 0.0;    // max / mean of the subdomains' particle cost at the last check (0: never checked)
    double imbalance_worst = 0.0;   // ... the worst seen at any check since opal_sim_recut_stats was last asked
    std::vector<size_t> cuts;       // first global cell of every subdomain, and one past the last one's
    // a re-cut planned and not yet complete (sim_recut): the step that follows makes the hand-over
    bool recut_pending = false;
    std::vector<size_t> pending_cuts;
    unsigned pending_ship[2] = {0, 0}, pending_hold[2] = {0, 0};
    bool pending_moved[2] = {true, true};  // whether the plan moves my left / right cut at all (link_hold)
    // Cuts that DRIFT between two plans: a beam that crosses the domain at c takes its particles past a fixed cut a cell's
    // worth per step, and a plan every few dozen steps hands the same particles back.  Each interior cut therefore carries a
    // velocity (cells per step) learnt from what the plans keep asking of it, and between plans moves one cell at a time in
    // the step's own pushes (PushArgs out_shift) -- no plan, no all-reduce, no column from a neighbour: a one-cell move only
    // uncovers cells the subdomain holds as ghosts (grid.hip: grid_shift).  Every subdomain computes the same moves.
    bool drift_on = true, pending_is_drift = false;
    std::vector<double> drift_v, drift_acc;  // per interior cut i = 1 .. N-1 (index i)
    size_t drift_shifts = 0;                 // steps in which some cut drifted
    size_t last_plan_step = 0;
    double drift_min = 0.6;  // a cut drifts once the plans ask this many cells per step of it (OPAL_RECUT_DRIFT_MIN)
    opal_allreduce_fn allreduce = nullptr;
    opal_gather_fn gather = nullptr;
    void *coll_user = nullptr;
    double t = 0;
    size_t total_steps = 0, steps_bt_output = 0;
    bool fused = true;
};

namespace {

struct LaserPair {
    std::function<double(const double *)> *y, *z;
};
double laser_y_cb(double t, double x, void *user) { return call_laser(t, x, static_cast<LaserPair *>(user)->y); }
double laser_z_cb(double t, double x, void *user) { return call_laser(t, x, static_cast<LaserPair *>(user)->z); }

// Which per-particle extras a population has to carry: the optical depth machinery when emission is on,
// plus whatever its `output:` specs read (mod.rs:388-449: "r" -> y, z; "work"; "chi").  Everything else
// stays out of HBM.
int track_for_outputs(opal_sim_t *s, opal_pop_t *p)
{
    const std::vector<std::string> &specs = p->species == OPAL_ELECTRON ? s->electron_output
                                            : p->species == OPAL_ION    ? s->ion_output
                                                                        : s->photon_output;
    auto mentions = [&](const char *name) {
        for (const std::string &sp : specs) {
            size_t from = 0;
            while (from <= sp.size()) {
                size_t to = sp.find_first_of(":(;)", from);
                if (to == std::string::npos) to = sp.size();
                if (sp.compare(from, to - from, name) == 0) return true;
                from = to + 1;
            }
        }
        return false;
    };
    unsigned fields = 0;
    if (mentions("chi")) fields |= OPAL_TRACK_CHI;
    if (mentions("work")) fields |= OPAL_TRACK_WORK;
    if (mentions("r")) fields |= OPAL_TRACK_YZ;
    if (s->photon_emission) fields |= p->species == OPAL_ELECTRON ? OPAL_TRACK_EMISSION : OPAL_TRACK_ID;
    if (s->photon_absorption)  // absorb() reads both species' chi, the photons' y, z, optical depths and birth times
        fields |= p->species == OPAL_ELECTRON ? (OPAL_TRACK_CHI | OPAL_TRACK_ID)
                                              : (OPAL_TRACK_CHI | OPAL_TRACK_YZ | OPAL_TRACK_ID | OPAL_TRACK_ABSORPTION);
    if (int rc = opal_pop_track_fields(p, fields)) return rc;
    // prev_x is only read by a separate deposit pass
    return opal_pop_keep_prev_x(p, s->current_deposition ? 1 : 0);
}

int load_species(opal_sim_t *s, const char *section, const char *density_key, int species, size_t npc, Rng &rng,
                 opal_pop_t **out)
{
    ConfigError ce;
    std::function<double(const double *)> ne, ux, uy, uz;
    if (!s->cfg.func(section, density_key, {"x"}, ne, ce)) return config_fail(ce);
    if (!s->cfg.func(section, "ux", {"x", "urand", "nrand"}, ux, ce)) return config_fail(ce);
    if (!s->cfg.func(section, "uy", {"x", "urand", "nrand"}, uy, ce)) return config_fail(ce);
    if (!s->cfg.func(section, "uz", {"x", "urand", "nrand"}, uz, ce)) return config_fail(ce);
    HostSoA soa;
    auto f3 = [](const std::function<double(const double *)> &f) {
        return [f](double x, double ur, double nr) {
            double a[3] = {x, ur, nr};
            return f(a);
        };
    };
    // (the expressions are the driver's own stack programs, Config::func: safe to evaluate from several threads)
    load_population(species, npc, [&](double x) { return ne(&x); }, f3(ux), f3(uy), f3(uz), opal_grid_xmin(s->grid),
                    s->dx, opal_grid_size(s->grid), rng, soa, loader_threads(true));
    // headroom for arrivals from neighbouring subdomains between two re-binnings
    size_t cap = soa.size() + soa.size() / 8 + 65536;
    if (int rc = opal_pop_create(out, species, cap, s->stream)) return rc;
    (*out)->id_base = (unsigned long long)s->rank << 40;  // ids unique across subdomains
    if (int rc = track_for_outputs(s, *out)) return rc;
    if (int rc = opal_pop_upload_soa(*out, soa.size(), soa.cell.data(), soa.x.data(), soa.ux.data(), soa.uy.data(),
                                     soa.uz.data(), soa.w.data(), s->dx, s->dt))
        return rc;
    if (s->photon_emission && species == OPAL_ELECTRON)  // Population::new ... .with_optical_depths(rng), mod.rs:193
        return opal_pop_upload_emission_state(*out, soa.size(), soa.tau.data(), nullptr);
    if (s->photon_absorption && species == OPAL_PHOTON && soa.size())
        return opal_pop_upload_absorption_state(*out, soa.size(), soa.tau.data(), soa.tau2.data(), nullptr);
    return OPAL_OK;
}

// main.rs:63-187
int sim_setup(opal_sim_t *s)
{
    ConfigError ce;
    Config &in = s->cfg;
    in.with_context("constants");
    if (!in.read_usize("control", "nx", s->nx, ce)) return config_fail(ce);
    if (!in.read_f64("control", "xmin", s->xmin, ce)) return config_fail(ce);
    if (!in.read_f64("control", "dx", s->dx, ce)) return config_fail(ce);
    s->dt = 0.95 * s->dx / 2.997925e8;  // main.rs:71
    if (!in.read_f64("control", "start", s->tstart, ce)) return config_fail(ce);
    if (!in.read_f64("control", "end", s->tend, ce)) return config_fail(ce);
    if (!in.read_bool("control", "current_deposition", s->current_deposition, ce)) return config_fail(ce);
    if (!in.read_usize("control", "n_outputs", s->n_outputs, ce)) return config_fail(ce);
    if (!in.read_bool("control", "balance", s->balance, ce)) s->balance = true;  // main.rs:76
    if (!in.read_bool("qed", "photon_emission", s->photon_emission, ce)) return config_fail(ce);
    if (!in.read_bool("qed", "photon_absorption", s->photon_absorption, ce)) return config_fail(ce);
    {  // main.rs:80-85: all optional
        ConfigError ignored;
        double v;
        // main.rs:247 hands `disable_qed_after` to absorb() as max_displacement and `disable_absorption_after` as
        // stop_time (interactions.rs:145): reproduced as called, whatever the key names suggest
        if (in.read_f64("qed", "disable_qed_after", v, ignored)) s->absorb_max_displacement = v;
        if (in.read_f64("qed", "disable_absorption_after", v, ignored)) s->absorb_stop_time = v;
        if (in.read_f64("qed", "max_formation_length", v, ignored)) s->max_formation_length = v;
        if (in.read_f64("qed", "photon_energy_min", v, ignored)) s->photon_energy_min = 1.0e-6 * v / 1.602177e-19;  // J -> MeV
        if (in.read_f64("qed", "photon_angle_max", v, ignored)) s->photon_angle_max = v;
    }
    // main.rs:89-101
    ConfigError ly_err, lz_err;
    bool have_y = in.func("laser", "Ey", {"t", "x"}, s->laser_y, ly_err);
    bool have_z = in.func("laser", "Ez", {"t", "x"}, s->laser_z, lz_err);
    int left = OPAL_BDY_INTERNAL, right = OPAL_BDY_INTERNAL;
    if (in.contains("laser")) {
        if (!have_y) return config_fail(ly_err);
        if (!have_z) return config_fail(lz_err);
        left = OPAL_BDY_LASER;
        right = OPAL_BDY_ABSORBING;
        s->has_laser = true;
    }
    size_t min_size = opal_grid_min_size();
    std::vector<size_t> nxs((size_t)s->numtasks);
    std::vector<double> offs((size_t)s->numtasks);
    if (s->balance) {  // main.rs:105-110
        std::function<double(const double *)> ne;
        if (!in.func("electrons", "ne", {"x"}, ne, ce)) return config_fail(ce);
        if (design_balanced(s->numtasks, s->nx, s->xmin, s->dx, min_size, [&](double x) { return ne(&x); }, nxs.data(),
                            offs.data())) {
            set_error("balanced grid design failed: no partition point found (reference panics, grid/mod.rs:179)");
            return OPAL_ERR_CONFIG;
        }
    } else {
        design_unbalanced(s->numtasks, s->nx, s->dx, min_size, nxs.data(), offs.data());
    }
    if (int rc = opal_grid_build(&s->grid, s->rank, s->numtasks, left, right, s->xmin, s->dx, nxs.data(), offs.data(),
                                 s->stream))
        return rc;
    s->nxs = nxs;
    s->cuts.assign((size_t)s->numtasks + 1, 0);
    for (int r = 0; r < s->numtasks; r++) s->cuts[(size_t)r + 1] = s->cuts[(size_t)r] + nxs[(size_t)r];

    Rng rng;
    rng.seed((uint64_t)s->rank);  // main.rs:53
    size_t epc;
    std::vector<std::string> &ospec = s->ion_output;
    if (!in.read_usize("electrons", "npc", epc, ce)) return config_fail(ce);
    if (!in.read_strings("electrons", "output", s->electron_output, ce)) return config_fail(ce);
    if (epc > 0) {
        if (int rc = load_species(s, "electrons", "ne", OPAL_ELECTRON, epc, rng, &s->electrons)) return rc;
    }
    size_t ipc;
    if (!in.read_usize("ions", "npc", ipc, ce)) return config_fail(ce);
    if (ipc > 0) {
        double z, a;
        if (!in.read_strings("ions", "output", ospec, ce)) return config_fail(ce);
        if (!in.read_string("ions", "name", s->ion_name, ce)) return config_fail(ce);
        if (!in.read_f64("ions", "Z", z, ce)) return config_fail(ce);
        if (!in.read_f64("ions", "A", a, ce)) return config_fail(ce);
        if (int rc = load_species(s, "ions", "ni", OPAL_ION, ipc, rng, &s->ions)) return rc;
        if (int rc = opal_pop_set_charge_state(s->ions, z, z, a)) return rc;  // main.rs:142-147
    }
    size_t ppc = 0;
    if (s->photon_emission || s->photon_absorption) {  // main.rs:153-170
        if (!in.read_usize("photons", "npc", ppc, ce)) return config_fail(ce);
        if (!in.read_strings("photons", "output", s->photon_output, ce)) return config_fail(ce);
        if (ppc > 0) {
            if (int rc = load_species(s, "photons", "nph", OPAL_PHOTON, ppc, rng, &s->photons)) return rc;
        } else {
            if (int rc = opal_pop_create(&s->photons, OPAL_PHOTON, 1 << 20, s->stream)) return rc;
            s->photons->id_base = (unsigned long long)s->rank << 40;
            if (int rc = track_for_outputs(s, s->photons)) return rc;
        }
        // The reference goes on drawing from the generator the loader used (seeded with the rank,
        // main.rs:53); the build's counter-based generator takes one key for the whole run, so that a
        // particle's draws do not depend on which subdomain it is in.
        s->emission_seed = 0x6f70616c5f616d64ull;
    }
    // A link that grows sizes each edge by the particles that stood near the cut a few exchanges earlier (link.cpp) -- history
    // the first exchanges do not have.  What the loader put near each cut is known without any: npc particles in every cell
    // whose density is positive (mod.rs:178-182), on this side and on the neighbour's, from the input file alone -- every rank
    // evaluates the same expressions at the same cell centres and so arrives at its neighbour's number too.  An edge with
    // less room than that grows here, before the first step; from then on the zone counts carry the bound.
    if (s->link && s->link->growable) {
        const int zone = link_is_peer(s->link) ? LINK_PEER_ZONE : LINK_ZONE;
        struct Sp { const char *section, *key; size_t npc; };
        std::vector<Sp> species;
        if (s->electrons) species.push_back({"electrons", "ne", epc});
        if (s->ions) species.push_back({"ions", "ni", ipc});
        if (s->photons && ppc > 0) species.push_back({"photons", "nph", ppc});
        size_t need[2] = {0, 0};
        const int nt = s->numtasks;
        const int nb[2] = {(s->rank - 1 + nt) % nt, (s->rank + 1) % nt};
        const bool linked[2] = {opal_grid_to_left(s->grid) >= 0, opal_grid_to_right(s->grid) >= 0};
        size_t electrons_near[2] = {0, 0};
        for (const Sp &sp : species) {
            std::function<double(const double *)> n;
            if (!in.func(sp.section, sp.key, {"x"}, n, ce)) return config_fail(ce);
            auto populated = [&](int r, long c0, long c1) {  // cells [c0, c1) of subdomain r that the loader fills
                size_t k = 0;
                const double x0 = s->xmin + offs[(size_t)r];
                for (long c = std::max(c0, 0L); c < std::min(c1, (long)nxs[(size_t)r]); c++) {
                    double x = x0 + ((double)c + 0.5) * s->dx;
                    if (n(&x) * s->dx > 0.0) k++;
                }
                return k;
            };
            for (int side = 0; side < 2; side++) {
                if (!linked[side]) continue;
                const long mine_n = (long)nxs[(size_t)s->rank], theirs_n = (long)nxs[(size_t)nb[side]];
                size_t cells = side == 0 ? populated(s->rank, 0, zone) + populated(nb[0], theirs_n - zone, theirs_n)
                                         : populated(s->rank, mine_n - zone, mine_n) + populated(nb[1], 0, zone);
                size_t count = cells * sp.npc;
                if (sp.section[0] == 'e') electrons_near[side] = count;
                need[side] = std::max(need[side], count);
            }
        }
        if (s->photons && s->link->nfields >= 12 && s->link->nfields < 15)  // photons born of those electrons (link_plan)
            for (int side = 0; side < 2; side++) need[side] = std::max(need[side], (size_t)zone * electrons_near[side]);
        size_t grow_to[2] = {0, 0};
        for (int side = 0; side < 2; side++) {
            if (!need[side]) continue;
            size_t slots = (need[side] + LINK_SLACK + LINK_ROWS - 1) / LINK_ROWS * LINK_ROWS;
            if (slots > s->link->cap_side[side]) grow_to[side] = link_room_for(s->link, slots);
        }
        if (grow_to[0] || grow_to[1])
            if (int rc = link_grow(s->link, grow_to)) return rc;
    }
    s->t = s->tstart;
    s->total_steps = (size_t)((s->tend - s->tstart) / s->dt);  // main.rs:186 (`as usize` truncates)
    s->steps_bt_output = s->n_outputs ? s->total_steps / s->n_outputs : 0;
    return OPAL_OK;
}

// main.rs:23-42
int write_energies(opal_sim_t *s, size_t index)
{
    double f, e, i, ph = 0.0;
    if (int rc = opal_sim_energies(s, &f, &e, &i)) return rc;
    if (s->photons)
        if (int rc = opal_pop_total_kinetic_energy(s->photons, &ph)) return rc;
    if (s->numtasks > 1) {  // yee.rs:801-808, mod.rs:232-239: MPI_Reduce(sum) to rank 0
        double v[4] = {f, e, i, ph};
        if (s->allreduce(s->coll_user, v, 4, 0)) {
            set_error("energy reduction failed in the caller's allreduce");
            return OPAL_ERR_LINK;
        }
        f = v[0]; e = v[1]; i = v[2]; ph = v[3];
    }
    if (s->rank != 0) return OPAL_OK;
    std::string fn = s->output_dir + "/" + std::to_string(index) + "_energy.dat";
    FILE *fp = fopen(fn.c_str(), "w");
    if (!fp) {
        set_error("cannot write %s", fn.c_str());
        return OPAL_ERR_IO;
    }
    fprintf(fp, "em_field %s\n", rust_lower_exp_prec(f, 6).c_str());
    fprintf(fp, "electrons %s\n", rust_lower_exp_prec(e, 6).c_str());
    fprintf(fp, "ions %s\n", rust_lower_exp_prec(i, 6).c_str());
    fprintf(fp, "photons %s\n", rust_lower_exp_prec(ph, 6).c_str());
    fclose(fp);
    return OPAL_OK;
}

}  // namespace

extern "C" {

int opal_grid_write_data(const opal_grid_t *g, const char *dir, size_t index)
{
    if (!g || !dir) {
        set_error("opal_grid_write_data: bad arguments");
        return OPAL_ERR_ARG;
    }
    if (opal_grid_ngrids(g) != 1) {
        set_error("opal_grid_write_data: single-subdomain only; gather opal_grid_interpolate() rows on rank 0 and "
                  "call opal_grid_write_rows otherwise (opal_sim_run does)");
        return OPAL_ERR_ARG;
    }
    size_t n = opal_grid_size(g);
    std::vector<double> rows(11 * n);
    if (int rc = opal_grid_interpolate(g, rows.data())) return rc;
    return opal_grid_write_rows(rows.data(), n, dir, index);
}

/* {index}_grid.dat from `n` rows of opal_grid_interpolate (x rho jx jy jz Ex Ey Ez Bx By Bz), yee.rs:762-773 */
int opal_grid_write_rows(const double *rows_in, size_t n, const char *dir, size_t index)
{
    if (!rows_in || !dir) {
        set_error("opal_grid_write_rows: bad arguments");
        return OPAL_ERR_ARG;
    }
    const double *rows = rows_in;
    std::string fn = std::string(dir) + "/" + std::to_string(index) + "_grid.dat";
    FILE *fp = fopen(fn.c_str(), "w");
    if (!fp) {
        set_error("cannot write %s", fn.c_str());
        return OPAL_ERR_IO;
    }
    for (size_t k = 0; k < n; k++) {  // yee.rs:765-773
        const double *r = rows + 11 * k;
        std::string line;
        for (int c = 0; c < 10; c++) line += rust_lower_exp(r[c]) + " ";
        line += rust_display(r[10]);
        fprintf(fp, "%s\n", line.c_str());
    }
    fclose(fp);
    return OPAL_OK;
}

int opal_design_unbalanced(int numtasks, size_t size, double dx, size_t min_subsize, size_t *nx, double *offset)
{
    if (numtasks < 1 || !nx || !offset) {
        set_error("opal_design_unbalanced: bad arguments");
        return OPAL_ERR_ARG;
    }
    design_unbalanced(numtasks, size, dx, min_subsize, nx, offset);
    return OPAL_OK;
}

int opal_design_balanced(int numtasks, size_t size, double xmin, double dx, size_t min_subsize, opal_fn1 ne, void *user,
                         size_t *nx, double *offset)
{
    if (numtasks < 1 || !nx || !offset || !ne) {
        set_error("opal_design_balanced: bad arguments");
        return OPAL_ERR_ARG;
    }
    if (design_balanced(numtasks, size, xmin, dx, min_subsize, [&](double x) { return ne(x, user); }, nx, offset)) {
        set_error("balanced grid design failed: no partition point found (reference panics, grid/mod.rs:179)");
        return OPAL_ERR_CONFIG;
    }
    return OPAL_OK;
}

int opal_pop_new(opal_pop_t **out, int species, size_t npc, opal_fn1 number_density, opal_fn3 ux, opal_fn3 uy,
                 opal_fn3 uz, void *user, const opal_grid_t *grid, uint64_t seed, double dt, void *hip_stream)
{
    if (!out || !number_density || !ux || !uy || !uz || !grid) {
        set_error("opal_pop_new: bad arguments");
        return OPAL_ERR_ARG;
    }
    Rng rng;
    rng.seed(seed);
    HostSoA soa;
    load_population(species, npc, [&](double x) { return number_density(x, user); },
                    [&](double x, double a, double b) { return ux(x, a, b, user); },
                    [&](double x, double a, double b) { return uy(x, a, b, user); },
                    [&](double x, double a, double b) { return uz(x, a, b, user); }, opal_grid_xmin(grid),
                    opal_grid_dx(grid), opal_grid_size(grid), rng, soa, loader_threads(false));
    size_t cap = soa.size() + soa.size() / 8 + 65536;
    if (int rc = opal_pop_create(out, species, cap, hip_stream)) return rc;
    return opal_pop_upload_soa(*out, soa.size(), soa.cell.data(), soa.x.data(), soa.ux.data(), soa.uy.data(),
                               soa.uz.data(), soa.w.data(), opal_grid_dx(grid), dt);
}

// Population::new on the host only: fills caller arrays (capacity entries each) and returns the count.
int opal_loader_generate(int species, size_t npc, opal_fn1 number_density, opal_fn3 ux, opal_fn3 uy, opal_fn3 uz,
                         void *user, double grid_xmin, double dx, size_t ncells, uint64_t seed, size_t capacity,
                         size_t *n_out, int32_t *cell, double *x, double *pux, double *puy, double *puz, double *w)
{
    if (!number_density || !ux || !uy || !uz || !n_out) {
        set_error("opal_loader_generate: bad arguments");
        return OPAL_ERR_ARG;
    }
    Rng rng;
    rng.seed(seed);
    HostSoA soa;
    load_population(species, npc, [&](double xx) { return number_density(xx, user); },
                    [&](double xx, double a, double b) { return ux(xx, a, b, user); },
                    [&](double xx, double a, double b) { return uy(xx, a, b, user); },
                    [&](double xx, double a, double b) { return uz(xx, a, b, user); }, grid_xmin, dx, ncells, rng, soa,
                    loader_threads(false));
    *n_out = soa.size();
    if (soa.size() > capacity) {
        set_error("opal_loader_generate: %zu particles exceed the capacity %zu", soa.size(), capacity);
        return OPAL_ERR_CAPACITY;
    }
    if (cell) memcpy(cell, soa.cell.data(), sizeof(int32_t) * soa.size());
    if (x) memcpy(x, soa.x.data(), sizeof(double) * soa.size());
    if (pux) memcpy(pux, soa.ux.data(), sizeof(double) * soa.size());
    if (puy) memcpy(puy, soa.uy.data(), sizeof(double) * soa.size());
    if (puz) memcpy(puz, soa.uz.data(), sizeof(double) * soa.size());
    if (w) memcpy(w, soa.w.data(), sizeof(double) * soa.size());
    return OPAL_OK;
}

int opal_set_loader_threads(int n)
{
    set_loader_threads(n);
    return OPAL_OK;
}

// SimulationTime's Display (setup.rs:421-438) and the float formats of the output files
int opal_format(int kind, double v, char *buf, size_t len)
{
    std::string s = kind == 0 ? simulation_time(v) : kind == 1 ? rust_lower_exp(v) : kind == 2 ? rust_lower_exp_prec(v, 6)
                                                                                               : rust_display(v);
    if (!buf || len <= s.size()) {
        set_error("opal_format: buffer too small");
        return OPAL_ERR_ARG;
    }
    memcpy(buf, s.c_str(), s.size() + 1);
    return OPAL_OK;
}

// ----------------------------------------------------------------------- sim
static int sim_create_common(opal_sim_t **out, std::unique_ptr<opal_sim_t> s)
{
    int rc = sim_setup(s.get());
    if (rc) {
        opal_sim_free(s.release());
        return rc;
    }
    *out = s.release();
    return OPAL_OK;
}

int opal_sim_create(opal_sim_t **out, const char *yaml_path, int rank, int numtasks, opal_link_t *link, void *hip_stream)
{
    if (!out || !yaml_path || numtasks < 1 || rank < 0 || rank >= numtasks) {
        ConfigError ce{ConfigErrorKind::MissingFile, "", ""};
        return config_fail(ce);
    }
    std::unique_ptr<opal_sim_t> s(new opal_sim_t());
    ConfigError ce;
    if (!Config::from_file(yaml_path, s->cfg, ce)) return config_fail(ce);
    std::string p(yaml_path);
    size_t slash = p.find_last_of('/');
    s->output_dir = slash == std::string::npos ? "." : p.substr(0, slash);  // main.rs:60-61
    if (s->output_dir.empty()) s->output_dir = "/";
    s->rank = rank; s->numtasks = numtasks; s->link = link; s->stream = hip_stream;
    return sim_create_common(out, std::move(s));
}

int opal_sim_create_from_string(opal_sim_t **out, const char *yaml_text, const char *output_dir, int rank, int numtasks,
                                opal_link_t *link, void *hip_stream)
{
    if (!out || !yaml_text || numtasks < 1 || rank < 0 || rank >= numtasks) {
        set_error("opal_sim_create_from_string: bad arguments");
        return OPAL_ERR_ARG;
    }
    std::unique_ptr<opal_sim_t> s(new opal_sim_t());
    ConfigError ce;
    if (!Config::from_string(yaml_text, s->cfg, ce)) return config_fail(ce);
    s->output_dir = output_dir ? output_dir : ".";
    s->rank = rank; s->numtasks = numtasks; s->link = link; s->stream = hip_stream;
    return sim_create_common(out, std::move(s));
}

int opal_sim_free(opal_sim_t *s)
{
    if (!s) return OPAL_OK;
    opal_pop_free(s->electrons);
    opal_pop_free(s->ions);
    opal_pop_free(s->photons);
    opal_grid_free(s->grid);
    if (s->ev_edges) hipEventDestroy(s->ev_edges);
    if (s->ev_comm) hipEventDestroy(s->ev_comm);
    if (s->comm_stream) hipStreamDestroy(s->comm_stream);
    delete s;
    return OPAL_OK;
}

static int ensure_comm_stream(opal_sim_t *s)
{
    if (s->comm_stream) return OPAL_OK;
    OPAL_HIP(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
    OPAL_HIP(hipEventCreateWithFlags(&s->ev_edges, hipEventDisableTiming));
    OPAL_HIP(hipEventCreateWithFlags(&s->ev_comm, hipEventDisableTiming));
    return OPAL_OK;
}

// Cuts that follow the plasma.  The reference balances the decomposition once, on the electrons' initial density
// (grid/mod.rs:157-206, main.rs:103-110), and so does this driver by default; with opal_sim_set_recut the same partition
// rule is applied again every few steps to the particles where they stand (recut.cpp).  Collective: every subdomain
// enters it at the same step.
//   A. the subdomains' particle costs, all-reduced (one small message); below the threshold nothing else happens.
//   B. a per-cell histogram of every species (one pass over the cell indices), the histograms of the whole domain
//      all-reduced, the new cuts planned -- the same on every subdomain, from the same numbers; edges whose messages are too
//      small for what can cross them grow (both ends know the count).
//   C. the hand-over, one of two ways:
//      * IN THE NEXT STEP (the fused linked step, and steps without current deposition: whenever no pass reads the particles'
//        cells against the grid after the push).  That step's pushes gather and deposit on the grid as it is, but store every
//        particle's cell counted from the NEW cut and send what lies outside the NEW subdomain (particles.hip: PushArgs
//        out_shift) -- its ordinary leavers, through its ordinary exchange.  A particle changes subdomain only if it ends the
//        step on the other side of the new cut: cuts that go with a beam move nobody.  After the step's field solve the grid
//        columns that change hands (and the four ghost columns beyond) go through one all-reduce of a window only their old
//        owner fills, and the grid is rebuilt in place.  Nothing is re-binned: a uniform shift keeps the order.
//      * AT ONCE (the reference's operator order with deposition, photon absorption): particles in the cells given away are
//        packed as leavers with their cell in the receiver's frame (mod.rs:291-293, 366-368), exchanged, appended, every
//        population re-binned on the new grid.
//   The link's bound is re-seeded with what stands near the new cuts (link_hold); nxs / offsets follow, so {i}_grid.dat
//   still gathers in rank order (yee.rs:749-779).
// Nothing of the physics depends on where the cuts are (tests/test_gpu_recut.py: a re-cutting world equals the static
// oracle world), only the summation order of the deposit does.
namespace {

// the grid's columns to the new cuts: a window per moved cut -- the cells that change hands and the four beyond them, which
// become the receiver's ghost cells -- filled by their old owner, all-reduced (every other subdomain adds zeros)
int recut_grid(opal_sim_t *s, const std::vector<size_t> &cuts, const std::vector<size_t> &fresh)
{
    const int N = s->numtasks, rank = s->rank;
    opal_grid_t *g = s->grid;
    const size_t oldL = cuts[(size_t)rank], newL = fresh[(size_t)rank], newR = fresh[(size_t)rank + 1];
    const size_t new_size = newR - newL;
    const double *old_rows = grid_fetch_rows(g);
    if (!old_rows) return OPAL_ERR_HIP;
    const size_t old_lenp = g->lenp, old_len = g->len;
    const int L = g->L;
    struct Window { size_t a, b, at; int owner; };
    const size_t ghost = opal_grid_min_size() / 2;  // yee.rs:239: four ghost cells (the minimum subdomain is two ghost layers)
    std::vector<Window> win;
    size_t wtotal = 0;
    for (int i = 1; i < N; i++) {
        const size_t o = cuts[(size_t)i], n = fresh[(size_t)i];
        if (n == o) continue;
        Window w = n > o ? Window{o, n + ghost, wtotal, i} : Window{n - ghost, o, wtotal, i - 1};
        wtotal += 10 * (w.b - w.a);
        win.push_back(w);
    }
    std::vector<double> wbuf(wtotal, 0.0);
    for (const Window &w : win)
        if (w.owner == rank)
            for (int comp = 0; comp < 10; comp++)
                for (size_t c = w.a; c < w.b; c++)
                    wbuf[w.at + (size_t)comp * (w.b - w.a) + (c - w.a)] = old_rows[(size_t)(comp + 1) * old_lenp + (c - oldL) + (size_t)L];
    if (wtotal && s->allreduce(s->coll_user, wbuf.data(), wbuf.size(), 0)) {
        set_error("re-cut: the caller's allreduce failed");
        return OPAL_ERR_LINK;
    }
    const size_t new_len = new_size + (size_t)g->L + (size_t)g->R;
    std::vector<double> rows(10 * new_len, 0.0);
    for (size_t j = 0; j < new_len; j++) {
        const long gc = (long)newL + (long)j - (long)L;  // global cell of array index j (outside [newL, newR): ghost / boundary cells)
        const Window *from = nullptr;
        for (const Window &w : win)
            if (w.owner != rank && gc >= (long)w.a && gc < (long)w.b) from = &w;
        const long jo = gc - (long)oldL + (long)L;  // where my old array held that cell
        for (int comp = 0; comp < 10; comp++) {
            double val = 0.0;
            if (from) val = wbuf[from->at + (size_t)comp * (from->b - from->a) + ((size_t)gc - from->a)];
            else if (jo >= 0 && jo < (long)old_len) val = old_rows[(size_t)(comp + 1) * old_lenp + (size_t)jo];
            rows[(size_t)comp * new_len + j] = val;
        }
    }
    return grid_recut(g, new_size, (double)newL * s->dx, rows.data());
}

// the books after a re-cut: the link's bound re-seeded, the sizes every subdomain gathers the grid rows by
void recut_done(opal_sim_t *s)
{
    s->link->burst_parts = 0;
    if (!getenv("OPAL_RECUT_NO_HOLD")) link_hold(s->link, s->pending_hold, s->pending_moved);  // (the knob: a diagnostic)
    s->cuts = s->pending_cuts;
    for (int r = 0; r < s->numtasks; r++) s->nxs[(size_t)r] = s->cuts[(size_t)r + 1] - s->cuts[(size_t)r];
    s->recuts++;
    s->recut_pending = false;
    if (getenv("OPAL_RECUT_DEBUG") && s->rank == 0) {
        static const auto t_first = std::chrono::steady_clock::now();
        fprintf(stderr, "[recut] %.3f s, step %zu: sizes", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_first).count(), s->step_index);
        for (int r = 0; r < s->numtasks; r++) fprintf(stderr, " %zu", s->nxs[(size_t)r]);
        fprintf(stderr, "\n");
    }
}

// this step's messages ship, and its populations reserve room for, what the histogram says can cross each moving cut
void recut_caps(opal_sim_t *s)
{
    opal_link_t *link = s->link;
    for (int side = 0; side < 2; side++) {
        const unsigned cap = (unsigned)link->cap_side[side];
        const unsigned want = std::max(s->pending_ship[side], link->plan_cap[side]);
        link->plan_cap[side] = std::min(cap, std::max(want, (unsigned)LINK_ROWS));
        link->send_cap[side] = link_is_peer(link) ? cap : link->plan_cap[side];
    }
    // megabytes may change hands in this one exchange: twice the tail kernel's usual senders and unpackers per species and
    // side.  (Not more: they are workgroups of 1024 threads that all wait for the neighbours' messages, and the workgroup that
    // publishes this subdomain's message comes last in the launch -- with 64 per species and side, four ranks sharing a card in
    // a rehearsal kept one another's publishers off the chip until the 20 s timeout.  65 workgroups of six ranks fit at once.)
    const unsigned most = std::max(s->pending_ship[0], s->pending_ship[1]);
    link->burst_parts = most >= (1u << 17) ? 16u : 0u;
}

// after the field solve of the step whose pushes went by the new cuts: the grid follows
int sim_recut_finish(opal_sim_t *s)
{
    const auto t0 = std::chrono::steady_clock::now();
    opal_pop_t *pops[3] = {s->electrons, s->ions, s->photons};
    if (s->pending_is_drift) {
        // a drift step: the rows shift by the move of my left cut, on the device; nothing is drained, nobody is asked
        const size_t r = (size_t)s->rank;
        const int dL = (int)((long)s->pending_cuts[r] - (long)s->cuts[r]);
        const size_t new_size = s->pending_cuts[r + 1] - s->pending_cuts[r];
        if (int rc = grid_shift(s->grid, dL, new_size, (double)s->pending_cuts[r] * s->dx)) return rc;
        pops_reframe_end(pops, 3);
        if (int rc = pops_sort_deferred(pops, 3, s->grid)) return rc;
        s->cuts = s->pending_cuts;
        for (int q = 0; q < s->numtasks; q++) s->nxs[(size_t)q] = s->cuts[(size_t)q + 1] - s->cuts[(size_t)q];
        s->recut_pending = s->pending_is_drift = false;
        s->drift_shifts++;
        return OPAL_OK;
    }
    if (int rc = recut_grid(s, s->cuts, s->pending_cuts)) return rc;
    pops_reframe_end(pops, 3);
    if (int rc = pops_sort_deferred(pops, 3, s->grid)) return rc;
    recut_done(s);
    s->recut_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("OPAL_RECUT_DEBUG") && atoi(getenv("OPAL_RECUT_DEBUG")) > 2 && s->rank == 0)
        fprintf(stderr, "[recut] step %zu host ms: the grid's columns (drain of the hand-over step, all-reduce, rows rewritten) %.3f\n", s->step_index,
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return OPAL_OK;
}

// whether the hand-over of a re-cut can be made by the next step's own pushes (sim_recut, C)
bool recut_in_the_step(const opal_sim_t *s)
{
    const bool qed = s->photon_emission || s->photon_absorption;
    return !s->photon_absorption && !getenv("OPAL_RECUT_AT_ONCE") && ((s->fused && s->current_deposition && !qed) || !s->current_deposition);
}

// One step's drift: every interior cut whose accumulated velocity has reached a cell moves by one, as far as every subdomain
// keeps its minimum size; the step that follows stores cells and tests leavers in the new frame (as the hand-over of a plan).
void sim_drift_begin(opal_sim_t *s)
{
    const int N = s->numtasks;
    if (!s->drift_on || N < 2 || s->recut_pending || !s->link || s->drift_v.size() != (size_t)N + 1 || !recut_in_the_step(s)) return;
    bool any = false;
    std::vector<size_t> fresh(s->cuts);
    for (int i = 1; i < N; i++) {
        if (std::fabs(s->drift_v[(size_t)i]) < s->drift_min) continue;
        s->drift_acc[(size_t)i] += s->drift_v[(size_t)i];
        const int d = s->drift_acc[(size_t)i] >= 1.0 ? 1 : s->drift_acc[(size_t)i] <= -1.0 ? -1 : 0;
        if (!d) continue;
        s->drift_acc[(size_t)i] -= (double)d;
        fresh[(size_t)i] = (size_t)((long)fresh[(size_t)i] + d);
        any = true;
    }
    if (!any) return;
    const size_t min_size = opal_grid_min_size();
    for (int r = 0; r < N; r++)
        if (fresh[(size_t)r + 1] < fresh[(size_t)r] + min_size) return;  // (a subdomain at its minimum: this step's moves are dropped, by everybody alike)
    const size_t r = (size_t)s->rank;
    opal_pop_t *pops[3] = {s->electrons, s->ions, s->photons};
    if (s->photons && s->link->nslots < 3) pops[2] = nullptr;
    s->pending_cuts = fresh;
    s->pending_ship[0] = s->pending_ship[1] = 0;
    s->pending_is_drift = true;
    pops_reframe_begin(pops, 3, (int)((long)fresh[r] - (long)s->cuts[r]), fresh[r + 1] - fresh[r]);
    s->recut_pending = true;
}

int sim_recut(opal_sim_t *s, bool force)
{
    const int N = s->numtasks, rank = s->rank;
    if (N < 2 || s->recut_pending) return OPAL_OK;  // (pending: planned, the next step makes the hand-over)
    if (!s->link || !s->allreduce) {
        set_error("a re-cut of %d subdomains needs a link and opal_sim_set_collectives (the all-reduce its plan is made from)", N);
        return OPAL_ERR_ARG;
    }
    opal_grid_t *g = s->grid;
    opal_link_t *link = s->link;
    hipStream_t st = (hipStream_t)s->stream;
    opal_pop_t *pops[3] = {s->electrons, s->ions, s->photons};
    if (s->photons && link->nslots < 3) pops[2] = nullptr;  // (photons that never cross a cut: nothing of theirs is moved)
    // what a particle of each species costs a step, in units of the general push's 80 bytes (photons: 36 read, 12 written)
    const double weight[3] = {1.0, 1.0, 0.5};
    auto reduce = [&](double *v, size_t n) -> int {
        if (s->allreduce(s->coll_user, v, n, 0)) {
            set_error("re-cut: the caller's allreduce failed");
            return OPAL_ERR_LINK;
        }
        return OPAL_OK;
    };
    const bool to_left = opal_grid_to_left(g) >= 0, to_right = opal_grid_to_right(g) >= 0;
    const bool timing = getenv("OPAL_RECUT_DEBUG") && atoi(getenv("OPAL_RECUT_DEBUG")) > 2 && rank == 0;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return 1e3 * std::chrono::duration<double>(b - a).count(); };
    const auto tA = now();
    // ---- A
    // (the loads as the re-binning policy last saw them published -- at most a few steps old, read from mapped host memory --
    // where there is such a publication: the check then costs the host one small all-reduce and the device nothing; the host
    // runs a step or two ahead of the device and waits for its neighbours while the device works the queue down)
    std::vector<double> v(2 * (size_t)N, 0.0);
    for (int sp = 0; sp < 3; sp++)
        if (pops[sp]) {
            size_t live = 0;
            if (force || !pop_live_estimate(pops[sp], &live))
                if (int rc = pop_live_count(pops[sp], &live)) return rc;
            v[(size_t)rank] += weight[sp] * (double)live;
        }
    if (to_right && rank + 1 < N) {
        // particles per species the cut on my right can move at once: what the edge has room for, or can grow to
        double room = (double)link->cap_side[1];
        if (link->growable && !link->external) room = (double)link_room_for(link, ((size_t)1 << 31) / 5 * 4);
        v[(size_t)N + (size_t)rank] = room;
    }
    if (int rc = reduce(v.data(), v.size())) return rc;
    double total = 0.0, worst = 0.0;
    for (int r = 0; r < N; r++) {
        total += v[(size_t)r];
        worst = std::max(worst, v[(size_t)r]);
    }
    s->recut_checks++;
    s->imbalance_seen = total > 0.0 ? worst * (double)N / total : 1.0;
    s->imbalance_worst = std::max(s->imbalance_worst, s->imbalance_seen);
    if (getenv("OPAL_RECUT_DEBUG") && rank == 0 && atoi(getenv("OPAL_RECUT_DEBUG")) > 1) {
        fprintf(stderr, "[recut] step %zu check: loads", s->step_index);
        for (int r = 0; r < N; r++) fprintf(stderr, " %.0f", v[(size_t)r]);
        fprintf(stderr, " -> max / mean %.4f\n", s->imbalance_seen);
    }
    if (!force && s->imbalance_seen < s->recut_threshold) return OPAL_OK;
    const auto tB = now();
    // ---- B
    // The plan needs the histogram only near the cuts -- a cut moves max_shift cells at the most, and the prefix sum AT a cut is
    // the sum of the subdomains before it -- so what is all-reduced is a window of every species' histogram round every cut
    // (wide enough for the link's bound at the new cut too) and every subdomain's totals: a few thousand numbers whatever
    // the size of the domain.  Outside the windows a subdomain's remaining particles are spread evenly: a cut whose place
    // lies out there goes max_shift cells towards it either way.
    const size_t ncells = s->cuts[(size_t)N], size = g->size;
    const long W = (long)s->recut_max_shift + 2 * (long)(link_is_peer(link) ? LINK_PEER_ZONE : LINK_ZONE) + 3;
    struct Win { long a, b; size_t at; };
    std::vector<Win> wins;
    size_t wtot = 0;
    for (int i = 0; i <= N; i++) {
        Win w{std::max(0L, (long)s->cuts[(size_t)i] - W), std::min((long)ncells, (long)s->cuts[(size_t)i] + W), wtot};
        wtot += 3 * (size_t)(w.b - w.a);
        wins.push_back(w);
    }
    std::vector<double> wv(wtot + 3 * (size_t)N, 0.0);
    std::vector<unsigned> counts;
    const long myL = (long)s->cuts[(size_t)rank];
    for (int sp = 0; sp < 3; sp++) {
        if (!pops[sp]) continue;
        if (int rc = pop_cell_histogram(pops[sp], g, counts)) return rc;
        double total_sp = 0.0;
        for (size_t c = 0; c < size; c++) total_sp += (double)counts[c];
        wv[wtot + (size_t)sp * (size_t)N + (size_t)rank] = total_sp;
        for (const Win &w : wins)
            for (long c = std::max(w.a, myL); c < std::min(w.b, myL + (long)size); c++)
                wv[w.at + (size_t)sp * (size_t)(w.b - w.a) + (size_t)(c - w.a)] = (double)counts[(size_t)(c - myL)];
    }
    const auto tB1 = now();
    if (int rc = reduce(wv.data(), wv.size())) return rc;
    const auto tB2 = now();
    std::vector<double> hist(3 * ncells, 0.0);
    {
        std::vector<char> known(ncells, 0);
        for (int sp = 0; sp < 3; sp++)
            for (const Win &w : wins)
                for (long c = w.a; c < w.b; c++) {
                    hist[(size_t)sp * ncells + (size_t)c] = wv[w.at + (size_t)sp * (size_t)(w.b - w.a) + (size_t)(c - w.a)];
                    known[(size_t)c] = 1;
                }
        for (int r = 0; r < N; r++) {
            const size_t a = s->cuts[(size_t)r], b = s->cuts[(size_t)r + 1];
            size_t unknown = 0;
            for (size_t c = a; c < b; c++) unknown += known[c] ? 0 : 1;
            if (!unknown) continue;
            for (int sp = 0; sp < 3; sp++) {
                double in_windows = 0.0;
                for (size_t c = a; c < b; c++)
                    if (known[c]) in_windows += hist[(size_t)sp * ncells + c];
                const double each = std::max(0.0, wv[wtot + (size_t)sp * (size_t)N + (size_t)r] - in_windows) / (double)unknown;
                for (size_t c = a; c < b; c++)
                    if (!known[c]) hist[(size_t)sp * ncells + c] = each;
            }
        }
    }
    std::vector<double> cost(ncells), most(ncells);
    for (size_t c = 0; c < ncells; c++) {
        const double e = hist[c], i = hist[ncells + c], ph = hist[2 * ncells + c];
        cost[c] = weight[0] * e + weight[1] * i + weight[2] * ph;
        most[c] = std::max(e, std::max(i, ph));
    }
    std::vector<double> room((size_t)N - 1);
    for (int i = 1; i < N; i++)
        room[(size_t)i - 1] = std::max(0.0, std::min(v[(size_t)N + (size_t)i - 1] - 2.0 * LINK_SLACK, (double)s->recut_max_particles));
    const std::vector<size_t> cuts(s->cuts);
    std::vector<size_t> fresh((size_t)N + 1);
    double before = 0.0, after = 0.0;
    if (int rc = opal_recut_plan(N, cuts.data(), cost.data(), most.data(), opal_grid_min_size(), s->recut_max_shift, room.data(),
                                 fresh.data(), &before, &after))
        return rc;
    // (No test of `after` against `before`: a cut that cannot reach its place in one go -- max_shift, its edge's room -- may
    // leave its neighbour with more than it had for a check or two; every cut only ever moves towards the place the rule
    // gives it, so the partition converges all the same, and holding out for an immediate gain would pin it where it is.)
    if (fresh == cuts) return OPAL_OK;
    const size_t oldL = cuts[(size_t)rank], oldR = cuts[(size_t)rank + 1], newL = fresh[(size_t)rank], newR = fresh[(size_t)rank + 1];
    const size_t new_size = newR - newL;
    const int give_left = newL > oldL ? (int)(newL - oldL) : 0, take_left = oldL > newL ? (int)(oldL - newL) : 0;
    const int give_right = oldR > newR ? (int)(oldR - newR) : 0;
    auto most_in = [&](long a, long b) {  // particles of the fullest species in the global cells [a, b) (a bound on any one species')
        double n = 0.0;
        for (long c = std::max(a, 0L); c < b && c < (long)ncells; c++) n += most[(size_t)c];
        return n;
    };
    // what can cross each of my edges in the hand-over: everything in the cells that change hands and, when the hand-over
    // is a step's push, the cell either side (a particle moves less than a cell per push)
    size_t grow_to[2] = {0, 0};
    const size_t at_old[2] = {oldL, oldR}, at_new[2] = {newL, newR};
    for (int side = 0; side < 2; side++) {
        s->pending_ship[side] = 0;
        if (!(side == 0 ? to_left : to_right)) continue;
        const long lo = (long)std::min(at_old[side], at_new[side]) - 1, hi = (long)std::max(at_old[side], at_new[side]) + 1;
        const size_t slots = ((size_t)most_in(lo, hi) + LINK_SLACK + LINK_ROWS - 1) / LINK_ROWS * LINK_ROWS;
        s->pending_ship[side] = (unsigned)slots;
        if (slots > link->cap_side[side]) grow_to[side] = link_room_for(link, slots);
    }
    if (grow_to[0] || grow_to[1]) {
        if (!link->growable) {
            set_error("re-cut: an edge would have to grow, but this link does not (opal_link_set_growth)");
            return OPAL_ERR_CAPACITY;
        }
        if (int rc = link_grow(link, grow_to)) return rc;
    }
    // what stands within the zone of my new cuts, on both sides: the link's bound until exchanges made with these cuts have
    // filed their own counts (link_hold); photons born meanwhile come from the electrons there (link_plan)
    {
        // (+ 1: the hand-over step itself, when it is a push; twice the zone where cuts may drift meanwhile: link.hpp zone_scale)
        const long zone = (long)(link_is_peer(link) ? LINK_PEER_ZONE : LINK_ZONE) * (s->drift_on ? 2 : 1) + 1;
        for (int side = 0; side < 2; side++) {
            s->pending_hold[side] = 0;
            s->pending_moved[side] = at_new[side] != at_old[side];
            if (!(side == 0 ? to_left : to_right)) continue;
            double n = 0.0, ne = 0.0;
            for (long c = (long)at_new[side] - zone; c < (long)at_new[side] + zone; c++) {
                const size_t cc = (size_t)((c % (long)ncells + (long)ncells) % (long)ncells);  // (a periodic wrap: the cut at 0 is the cut at ncells)
                n += most[cc];
                ne += hist[cc];
            }
            if (link->nfields >= 12 && link->nslots >= 3) n += (double)zone * ne;
            s->pending_hold[side] = (unsigned)std::min(n, 4.0e9);
        }
    }
    s->pending_cuts = fresh;
    if (getenv("OPAL_RECUT_DEBUG") && rank == 0)
        fprintf(stderr, "[recut] step %zu: imbalance %.3f -> %.3f (planned)\n", s->step_index, before, after);
    // ---- C
    const bool in_the_step = recut_in_the_step(s);
    // what the plan asks of each cut, per step since the last plan, is what the cut should have been doing by itself: half of
    // it goes into the cut's drift (a beam's cuts settle at its speed, and the plans then find nothing to correct; a cut
    // nobody asks anything of for a while comes to rest)
    if (s->drift_on && in_the_step && !getenv("OPAL_RECUT_NO_DRIFT")) {
        if (s->drift_v.size() != (size_t)N + 1) {
            s->drift_v.assign((size_t)N + 1, 0.0);
            s->drift_acc.assign((size_t)N + 1, 0.0);
        }
        const double steps = (double)std::max<size_t>(1, s->step_index - s->last_plan_step);
        bool drifting = false;
        for (int i = 1; i < N; i++) {
            double &v = s->drift_v[(size_t)i];
            const double rate = ((double)fresh[(size_t)i] - (double)cuts[(size_t)i]) / steps;
            // A cut that drifts left the plan only what its drift did not cover: half of that is added.  A cut that stands
            // left it everything: the rate IS the estimate (averaged with the last one) -- summing it up, as for a drifting
            // cut, took the hole-boring cuts (0.3-0.4 cells per step, caught up with every 50-75 steps) past the threshold
            // within three plans, to 0.75 cells per step, from where the plans had to pull them back (6-rank rehearsal).
            if (std::fabs(v) >= s->drift_min) v += 0.5 * rate;
            else v = 0.5 * (v + rate);
            v = std::max(-1.0, std::min(1.0, v));
            if (std::fabs(v) < 0.02) v = 0.0;
            // A cut only drifts with what moves like a beam (0.6 cells per step or more).  A cut that follows a
            // front through a hot, dense plasma -- hole boring: 0.2-0.35 cells per step -- gains nothing from moving a cell at
            // a time: the particles in the cell it crosses move every way, so a cell's worth changes hands per cell whoever
            // moves the cut, and moving it in many small steps cost the 4-rank laser-plasma rehearsal a factor of 2.5 (round 5).
            drifting = drifting || std::fabs(v) >= s->drift_min;
        }
        s->last_plan_step = s->step_index;
        link->zone_scale = drifting ? 2u : 1u;  // (counted from this step's pushes on; the hold below covers the exchanges before)
    }
    if (timing)
        fprintf(stderr, "[recut] step %zu host ms: loads + all-reduce %.3f, histograms %.3f, their all-reduce %.3f, plan + growth %.3f\n",
                s->step_index, ms(tA, tB), ms(tB, tB1), ms(tB1, tB2), ms(tB2, now()));
    if (in_the_step) {
        pops_reframe_begin(pops, 3, (int)((long)newL - (long)oldL), new_size);
        s->recut_pending = true;  // opal_sim_step: recut_caps after the step's link_begin, sim_recut_finish after its field solve
        return OPAL_OK;
    }
    if (int rc = link_begin(link, st, 0)) return rc;  // (species mask 0: this exchange's zone counts are nobody's bound)
    recut_caps(s);
    if (int rc = pops_recut_pack(pops, 3, link, size, give_left, give_right, take_left - give_left)) return rc;
    if (int rc = recut_grid(s, cuts, fresh)) return rc;
    if (int rc = link_exchange_particles(link, to_left, to_right, st)) return rc;
    if (int rc = pops_recut_finish(pops, 3, link, g)) return rc;
    recut_done(s);
    return OPAL_OK;
}

}  // namespace

// main.rs:238-267
int opal_sim_step(opal_sim_t *s, size_t n)
{
    if (!s) {
        set_error("null simulation handle");
        return OPAL_ERR_ARG;
    }
    LaserPair lp{&s->laser_y, &s->laser_z};
    const bool linked = (s->numtasks > 1 || s->grid->self_link) && (opal_grid_to_left(s->grid) >= 0 || opal_grid_to_right(s->grid) >= 0);
    if (linked && !s->link) {
        set_error("%d subdomains need a link", s->numtasks);
        return OPAL_ERR_ARG;
    }
    // photons and the QED state of every particle migrate too (the reference ships whole structs)
    const bool qed = s->photon_emission || s->photon_absorption;
    if (linked && qed && (s->link->nslots < 3 || s->link->nfields < (s->photon_absorption ? 15 : 12))) {
        set_error("photon emission / absorption across subdomains needs a link created with 3 species slots and 12 (15 "
                  "with absorption) fields per particle (opal_link_create_layout): optical depths, ids and photons "
                  "migrate too");
        return OPAL_ERR_LINK;
    }
    // absorb (main.rs:246-248), between the pushes and the emission
    auto absorb_now = [&]() -> int {
        if (!s->photon_absorption || !s->electrons || !s->photons) return OPAL_OK;
        return opal_absorb(s->electrons, s->photons, s->grid, s->emission_seed, (uint32_t)s->step_index, s->t, s->dt,
                           s->absorb_max_displacement, s->absorb_stop_time, nullptr, nullptr);
    };
    // yee.rs:456-462: the laser is injected by rank 0 at one cell, at the pre-increment time
    auto laser_now = [&](double &ly, double &lz) {
        ly = lz = 0.0;
        opal_grid_t *g = s->grid;
        if (s->has_laser && opal_grid_rank(g) == 0) {
            double xs = g->cell_x((size_t)(g->L - 2));
            double r = 2.997925e8 * s->dt / s->dx;
            ly = 2.0 * r * laser_y_cb(s->t, xs, &lp);
            lz = 2.0 * r * laser_z_cb(s->t, xs, &lp);
        }
    };
    // rho is a diagnostic (yee.rs:606-609: no operator reads it) that every step's clear erases: only the one the
    // last step of this call leaves behind can be looked at, so only that step's paired push deposits it
    struct RhoGuard {
        opal_grid_t *g;
        ~RhoGuard() { g->want_rho = true; }
    } rho_guard{s->grid};
    static const bool debug_cells = getenv("OPAL_DEBUG_CELLS") != nullptr;
    static const size_t debug_from = getenv("OPAL_DEBUG_CELLS") ? (size_t)atol(getenv("OPAL_DEBUG_CELLS")) : 0;
    for (size_t k = 0; k < n; k++) {
        // cuts that follow the plasma (off unless asked for: opal_sim_set_recut): between two steps, where nothing is in flight
        if (s->recut_every && s->numtasks > 1 && s->step_index && s->step_index % s->recut_every == 0) {
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = sim_recut(s, false);
            s->recut_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rc) return rc;
        }
        if (s->recut_every && s->numtasks > 1) sim_drift_begin(s);  // (a cut that drifts moves a cell in this step's pushes)
        s->grid->want_rho = k + 1 == n;
        if (linked && !(s->fused && s->current_deposition && !qed)) {
            // The reference's operator order across subdomains (main.rs:240-267) with the leavers of every
            // species in ONE message per neighbour.  Without current deposition the halo does not depend on
            // the particles and rides along (one exchange per step); with it the halo follows the deposit in
            // a second, 704-byte exchange.
            opal_grid_t *g = s->grid;
            hipStream_t st = (hipStream_t)s->stream;
            opal_pop_t *pops[3] = {s->electrons, s->ions, qed ? s->photons : nullptr};
            const bool to_left = opal_grid_to_left(g) >= 0, to_right = opal_grid_to_right(g) >= 0;
            if (int rc = link_begin(s->link, st, (s->electrons ? 1u : 0u) | (s->ions ? 2u : 0u) | (qed && s->photons ? 4u : 0u))) return rc;
            if (s->recut_pending) recut_caps(s);  // (the step in which the cuts move: sim_recut)
            for (opal_pop_t *p : pops)
                if (p)
                    if (int rc = pop_push_advance_noexchange(p, s->link, g, s->dt)) return rc;
            const double *rl = nullptr, *rr = nullptr;
            if (!s->current_deposition)
                if (int rc = grid_exchange_halo(g, s->link, false, &rl, &rr, nullptr)) return rc;
            if (int rc = link_exchange_particles(s->link, to_left, to_right, st)) return rc;
            if (int rc = pops_unpack(pops, 3, s->link, g)) return rc;
            if (int rc = absorb_now()) return rc;
            if (s->photon_emission && s->electrons)  // main.rs:250-252
                if (int rc = opal_emit_radiation(s->electrons, s->photons, s->emission_seed, (uint32_t)s->step_index, s->t,
                                                 s->photon_energy_min, s->photon_angle_max, s->max_formation_length,
                                                 nullptr))
                    return rc;
            // re-binning only now: the emission consumed the list of emitters (slot numbers) the push left
            for (opal_pop_t *p : pops)
                if (p)
                    if (int rc = pop_maybe_sort(p, g)) return rc;
            if (s->current_deposition) {  // main.rs:254-258
                if (int rc = opal_grid_clear(g)) return rc;
                if (s->electrons)
                    if (int rc = opal_grid_deposit(g, s->electrons, s->dt)) return rc;
                if (s->ions)
                    if (int rc = opal_grid_deposit(g, s->ions, s->dt)) return rc;
                if (int rc = grid_exchange_halo(g, s->link, true, &rl, &rr, nullptr)) return rc;
            }
            double ly, lz;
            laser_now(ly, lz);
            if (int rc = grid_sync_advance(g, rl, rr, ly, lz, s->dt)) return rc;
            s->t += s->dt;
            s->step_index++;
            const bool was_handover = s->recut_pending;
            if (s->recut_pending)
                if (int rc = sim_recut_finish(s)) return rc;
            if (debug_cells && s->step_index >= debug_from)
                for (opal_pop_t *p : {s->electrons, s->ions})
                    if (int rc = pop_debug_check_cells(p, s->grid, was_handover ? "after the hand-over" : "after", s->step_index)) return rc;
            continue;
        }
        if (linked && s->fused && s->current_deposition && !qed) {
            // One exchange per step: leavers of both species and the halo travel in the same message.
            // Leavers have already deposited into this subdomain's ghost cells, which the halo overlay
            // hands to the neighbour (yee.rs:424-452), so migration need not precede deposition.
            opal_grid_t *g = s->grid;
            hipStream_t st = (hipStream_t)s->stream;
            if (int rc = opal_grid_clear(g)) return rc;
            // (free after the first step: the unpack kernel re-zeroes the leaver counts)
            if (int rc = link_begin(s->link, st, (s->electrons ? 1u : 0u) | (s->ions ? 2u : 0u))) return rc;
            if (s->recut_pending) recut_caps(s);  // (the step in which the cuts move: sim_recut)
            const double *rl = nullptr, *rr = nullptr;
            const bool to_left = opal_grid_to_left(g) >= 0, to_right = opal_grid_to_right(g) >= 0;
            const bool peer = link_is_peer(s->link);
            bool overlapped = false;
            // Overlap: only the particles within a few cells of the subdomain's ends can leave it or deposit
            // into the cells the halo carries, and after a re-binning they sit in the first and last chunks
            // of the arrays.  Those chunks are pushed on a second stream, BESIDE the bulk (the two launches share the
            // GPU: the handful of edge workgroups would leave it idle on their own), and the message follows them
            // on that stream; the exchange -- pack, RCCL kernel, xGMI transfer, a neighbour that is a little
            // late -- is off the critical path as long as it takes less than the bulk push.
            // (not in the step in which the cuts move: its leavers come from as deep inside as the cuts go)
            if (s->overlap && !s->recut_pending && s->electrons && s->ions && pops_plan_split(s->electrons, s->ions, s->link, g) &&
                ensure_comm_stream(s) == OPAL_OK && link_can_use_stream(s->link, s->comm_stream)) {
                OPAL_HIP(hipEventRecord(s->ev_edges, st));  // everything before this step (the clear, the last solve)
                OPAL_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_edges, 0));
                if (int rc = pops_push_fused_pair_part(s->electrons, s->ions, s->link, g, s->dt, true, s->comm_stream)) return rc;
                if (int rc = pops_push_fused_pair_part(s->electrons, s->ions, s->link, g, s->dt, false, nullptr)) return rc;
                if (int rc = grid_exchange_halo(g, s->link, false, &rl, &rr, s->comm_stream)) return rc;
                if (int rc = link_exchange_particles(s->link, to_left, to_right, s->comm_stream)) return rc;
                OPAL_HIP(hipEventRecord(s->ev_comm, s->comm_stream));
                OPAL_HIP(hipStreamWaitEvent(st, s->ev_comm, 0));
                s->n_overlapped++;
                overlapped = true;
            } else {
                if (s->electrons && s->ions) {
                    if (int rc = pops_push_fused_pair_noexchange(s->electrons, s->ions, s->link, g, s->dt)) return rc;
                } else {
                    if (s->electrons)
                        if (int rc = pop_push_fused_noexchange(s->electrons, s->link, g, s->dt)) return rc;
                    if (s->ions)
                        if (int rc = pop_push_fused_noexchange(s->ions, s->link, g, s->dt)) return rc;
                }
                // a peer link needs nothing here: the tail kernel below packs the halo, writes both messages into the
                // neighbours' mailboxes and waits for theirs (peer_device.hpp)
                if (!peer) {
                    if (int rc = grid_exchange_halo(g, s->link, false, &rl, &rr, nullptr)) return rc;
                    if (int rc = link_exchange_particles(s->link, to_left, to_right, st)) return rc;
                }
            }
            double ly, lz;
            laser_now(ly, lz);
            {
                // the arrivals, synchronize and the field solve: one launch (the tiles synchronize does not touch
                // are advanced by workgroups of their own, beside the unpack)
                SyncArgs sa;
                grid_sync_args(g, rl, rr, ly, lz, sa);
                AdvArgs adv;
                grid_advance_args(g, s->dt, adv);
                const bool fits = adv.edge_left <= ADV_EDGE && adv.edge_right <= ADV_EDGE;
                opal_pop_t *pops[2] = {s->electrons, s->ions};
                if (int rc = pops_unpack_and_maybe_sort(pops, 2, s->link, g, &sa, fits ? &adv : nullptr, peer && !overlapped)) return rc;
                if (fits) grid_advance_done(g);
                else if (int rc = opal_grid_advance(g, s->dt)) return rc;
            }
            s->t += s->dt;
            s->step_index++;
            const bool was_handover = s->recut_pending;
            if (s->recut_pending)
                if (int rc = sim_recut_finish(s)) return rc;
            if (debug_cells && s->step_index >= debug_from)
                for (opal_pop_t *p : {s->electrons, s->ions})
                    if (int rc = pop_debug_check_cells(p, s->grid, was_handover ? "after the hand-over" : "after", s->step_index)) return rc;
            continue;
        }
        opal_pop_t *both[2] = {s->electrons, s->ions};
        PublishBatch pub{};
        bool sort_now[2] = {false, false}, deferred = false, deferred_qed = false;
        opal_pop_t *qed_pops[2] = {nullptr, nullptr};
        // With photon emission on, the reference's order is advance -> emit -> deposit (main.rs:240-258) and
        // the deposit's transverse currents use the velocity after the recoil, so the fused push+deposit
        // pass does not apply: the separate operators run in the reference's order.
        if (s->fused && s->current_deposition && !qed) {
            // clear first, then one pass per species that pushes and deposits
            if (int rc = opal_grid_clear(s->grid)) return rc;
            if (s->electrons && s->ions && !linked) {
                // both species in one launch, then the re-binning bookkeeping of each
                if (int rc = pops_push_fused_pair_noexchange(s->electrons, s->ions, nullptr, s->grid, s->dt)) return rc;
                // (the policy's counters are published by the field solve's launch below; re-binnings follow it)
                if (int rc = pops_policy(both, 2, &pub, sort_now)) return rc;
                deferred = true;
            } else {
                if (s->electrons)
                    if (int rc = opal_pop_advance_deposit(s->electrons, s->link, s->grid, s->dt)) return rc;
                if (s->ions)
                    if (int rc = opal_pop_advance_deposit(s->ions, s->link, s->grid, s->dt)) return rc;
            }
        } else {
            // both charged species of a lone subdomain in one launch per operator (no extras involved)
            const bool pair = !linked && s->electrons && s->ions && !qed && s->current_deposition;
            if (pair) {
                if (int rc = pops_advance_pair_noexchange(s->electrons, s->ions, s->grid, s->dt)) return rc;
                if (int rc = pops_maybe_sort(both, 2, s->grid)) return rc;
            } else {
                if (s->electrons) {
                    // with emission on, the electrons' re-binning waits until the emission has consumed the list of
                    // emitters their push leaves behind (a lone subdomain: Population::advance is push + policy)
                    if (qed && !linked) {
                        if (int rc = pop_push_advance_noexchange(s->electrons, nullptr, s->grid, s->dt)) return rc;
                    } else if (int rc = opal_pop_advance(s->electrons, s->link, s->grid, s->dt)) {
                        return rc;
                    }
                }
                if (s->ions)
                    if (int rc = opal_pop_advance(s->ions, s->link, s->grid, s->dt)) return rc;
            }
            // a lone subdomain with emission: the step's bookkeeping rides on the field solve's launch -- the re-binning
            // policy's counters of electrons and photons (one publication instead of a launch each), the emitter count
            // cleared for the next push -- and the re-binnings it decides follow that launch
            const bool batch_qed = qed && !linked && !s->photon_absorption && s->photon_emission && s->electrons && s->photons;
            if (qed) {
                if (s->photons) {
                    if (batch_qed) {
                        if (int rc = pop_push_advance_noexchange(s->photons, nullptr, s->grid, s->dt)) return rc;  // main.rs:242-244
                    } else if (int rc = opal_pop_advance(s->photons, s->link, s->grid, s->dt)) {
                        return rc;
                    }
                }
                if (int rc = absorb_now()) return rc;
                if (s->electrons && !s->photon_emission)
                    if (int rc = pop_maybe_sort(s->electrons, s->grid)) return rc;
                if (s->electrons && s->photon_emission) {  // main.rs:250-252
                    if (int rc = opal_emit_radiation(s->electrons, s->photons, s->emission_seed, (uint32_t)s->step_index,
                                                     s->t, s->photon_energy_min, s->photon_angle_max,
                                                     s->max_formation_length, nullptr))
                        return rc;
                    if (batch_qed) {
                        qed_pops[0] = s->electrons;
                        qed_pops[1] = s->photons;
                        if (int rc = pops_policy(qed_pops, 2, &pub, sort_now)) return rc;
                        pub.zero = &s->electrons->d_ctr->n_emit;  // (consumed by the emission above)
                        s->electrons->n_emit_clean = true;
                        deferred_qed = true;
                    } else if (!linked) {
                        if (int rc = pop_maybe_sort(s->electrons, s->grid)) return rc;
                    }
                }
            }
            if (s->current_deposition) {
                if (int rc = opal_grid_clear(s->grid)) return rc;
                if (pair) {
                    if (int rc = pops_deposit_pair(s->electrons, s->ions, s->grid, s->dt)) return rc;
                } else {
                    if (s->electrons)
                        if (int rc = opal_grid_deposit(s->grid, s->electrons, s->dt)) return rc;
                    if (s->ions)
                        if (int rc = opal_grid_deposit(s->grid, s->ions, s->dt)) return rc;
                }
            }
        }
        s->step_index++;
        if (!linked) {
            // a lone subdomain: synchronize and the field solve in one launch
            double ly, lz;
            laser_now(ly, lz);
            if (int rc = grid_sync_advance(s->grid, nullptr, nullptr, ly, lz, s->dt, &pub)) return rc;
            pub.n = 0;
        } else {
            if (s->has_laser) {
                if (int rc = opal_grid_synchronize(s->grid, s->link, laser_y_cb, laser_z_cb, &lp, s->t, s->dt)) return rc;
            } else {
                if (int rc = opal_grid_synchronize(s->grid, s->link, nullptr, nullptr, nullptr, s->t, s->dt)) return rc;
            }
            if (int rc = opal_grid_advance(s->grid, s->dt)) return rc;
        }
        if (deferred)
            if (int rc = pops_sort_decided(both, 2, sort_now, s->grid)) return rc;
        if (deferred_qed)
            if (int rc = pops_sort_decided(qed_pops, 2, sort_now, s->grid)) return rc;
        s->t += s->dt;
    }
    return OPAL_OK;
}

int opal_sim_sync(opal_sim_t *s)
{
    if (!s) {
        set_error("null simulation handle");
        return OPAL_ERR_ARG;
    }
    OPAL_HIP(hipStreamSynchronize((hipStream_t)s->stream));
    if (s->electrons)
        if (int rc = pop_check_err(s->electrons)) return rc;
    if (s->ions)
        if (int rc = pop_check_err(s->ions)) return rc;
    if (s->photons)
        if (int rc = pop_check_err(s->photons)) return rc;
    // a neighbour's message that never came (peer links: the kernels' waits are bounded and carry on) ends the run here,
    // as a blocking MPI exchange that cannot complete ends the reference's (yee.rs:358-409, particle/mod.rs:303-354)
    if (int rc = link_peer_check(s->link)) return rc;
    return OPAL_OK;
}

int opal_sim_energies(opal_sim_t *s, double *em_field, double *electrons, double *ions)
{
    if (!s) {
        set_error("null simulation handle");
        return OPAL_ERR_ARG;
    }
    double f = 0, e = 0, i = 0;
    if (int rc = opal_grid_em_field_energy(s->grid, &f)) return rc;
    if (s->electrons)
        if (int rc = opal_pop_total_kinetic_energy(s->electrons, &e)) return rc;
    if (s->ions)
        if (int rc = opal_pop_total_kinetic_energy(s->ions, &i)) return rc;
    if (em_field) *em_field = f;
    if (electrons) *electrons = e;
    if (ions) *ions = i;
    return OPAL_OK;
}

int opal_sim_set_collectives(opal_sim_t *s, opal_allreduce_fn allreduce, opal_gather_fn gather, void *user)
{
    if (!s) {
        set_error("null simulation handle");
        return OPAL_ERR_ARG;
    }
    s->allreduce = allreduce;
    s->gather = gather;
    s->coll_user = user;
    return OPAL_OK;
}

namespace {

// Grid::write_data, yee.rs:749-779: every subdomain's interpolated rows, in rank order, written by rank 0
int write_grid(opal_sim_t *s, size_t index)
{
    if (s->numtasks == 1) return opal_grid_write_data(s->grid, s->output_dir.c_str(), index);
    size_t n = opal_grid_size(s->grid), total = 0;
    std::vector<double> rows(11 * n), all;
    if (int rc = opal_grid_interpolate(s->grid, rows.data())) return rc;
    std::vector<size_t> counts(s->nxs.size());
    for (size_t r = 0; r < s->nxs.size(); r++) {
        counts[r] = 11 * s->nxs[r];
        total += s->nxs[r];
    }
    if (s->rank == 0) all.resize(11 * total);
    if (s->gather(s->coll_user, rows.data(), rows.size(), s->rank == 0 ? all.data() : nullptr, counts.data(), s->numtasks)) {
        set_error("gathering the grid rows failed in the caller's gather");
        return OPAL_ERR_LINK;
    }
    if (s->rank != 0) return OPAL_OK;
    return opal_grid_write_rows(all.data(), total, s->output_dir.c_str(), index);
}

// Population::write_data (mod.rs:383-568) across subdomains: the two-phase histogram ABI with the
// reference's all-reductions (hgram.rs:190-192, 249, 260) in between.  Collective: every subdomain calls
// it with the same specs.
int write_pop(opal_sim_t *s, opal_pop_t *p, const std::string &name, const std::vector<std::string> &specs, size_t index)
{
    if (!p || specs.empty()) return OPAL_OK;
    if (s->numtasks == 1) {
        std::vector<const char *> cs;
        for (auto &sp : specs) cs.push_back(sp.c_str());
        return opal_pop_write_data(p, s->grid, s->output_dir.c_str(), index, name.c_str(), cs.data(), cs.size());
    }
    auto reduce = [&](double *v, size_t n, int op) -> int {
        if (s->allreduce(s->coll_user, v, n, op)) {
            set_error("histogram reduction failed in the caller's allreduce");
            return OPAL_ERR_LINK;
        }
        return OPAL_OK;
    };
    for (const std::string &spec : specs) {
        char path[4096];
        if (opal_hist_filename(s->output_dir.c_str(), index, name.c_str(), spec.c_str(), path, sizeof path) != OPAL_OK)
            continue;  // a spec the reference skips: skipped by every subdomain alike
        int dim;
        double mn[2], mx[2];
        size_t n;
        if (int rc = opal_pop_hist_range(p, s->grid, spec.c_str(), &dim, mn, mx, &n)) return rc;
        double cnt = (double)n;
        if (int rc = reduce(mn, 2, 1)) return rc;
        if (int rc = reduce(mx, 2, 2)) return rc;
        if (int rc = reduce(&cnt, 1, 0)) return rc;
        if (cnt == 0.0) continue;  // hgram.rs:195-197
        opal_hist_t *h = nullptr;
        if (int rc = opal_pop_hist_fill(p, s->grid, spec.c_str(), mn, mx, (size_t)cnt, &h)) return rc;
        double total = opal_hist_total(h);
        int rc = reduce(opal_hist_counts(h), opal_hist_len(h), 0);
        if (!rc) rc = reduce(&total, 1, 0);
        if (!rc) rc = opal_hist_set_total(h, total);
        if (!rc) rc = opal_hist_finish(h);
        if (!rc && s->rank == 0) rc = opal_hist_write_fits(h, path);
        opal_hist_free(h);
        if (rc) return rc;
    }
    return OPAL_OK;
}

}  // namespace

// main.rs:217-283
int opal_sim_run(opal_sim_t *s)
{
    if (!s) {
        set_error("null simulation handle");
        return OPAL_ERR_ARG;
    }
    if (s->numtasks != 1 && (!s->allreduce || !s->gather)) {
        set_error("opal_sim_run with %d subdomains needs opal_sim_set_collectives (output-time reductions)", s->numtasks);
        return OPAL_ERR_ARG;
    }
    auto write_all = [&](size_t i) -> int {  // main.rs:219-222
        if (int rc = write_grid(s, i)) return rc;
        if (int rc = write_pop(s, s->electrons, "electron", s->electron_output, i)) return rc;
        if (int rc = write_pop(s, s->ions, s->ion_name, s->ion_output, i)) return rc;
        if (int rc = write_pop(s, s->photons, "photon", s->photon_output, i)) return rc;
        return write_energies(s, i);
    };
    // PrettyDuration, setup.rs:404-419; ettc, setup.rs:372-378
    auto pretty = [](double seconds) {
        unsigned long long t = (unsigned long long)seconds;
        const unsigned long long sec = t % 60, min = (t / 60) % 60, hr = (t / 3600) % 24, d = t / 86400;
        char buf[64];
        if (d > 0) snprintf(buf, sizeof buf, "%llud %02llu:%02llu:%02llu", d, hr, min, sec);
        else snprintf(buf, sizeof buf, "%02llu:%02llu:%02llu", hr, min, sec);
        return std::string(buf);
    };
    const auto runtime = std::chrono::steady_clock::now();  // main.rs:215
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - runtime).count(); };
    for (size_t i = 0; i < s->n_outputs; i++) {
        if (int rc = write_all(i)) return rc;
        if (s->rank == 0) {  // main.rs:224-234
            // (total_steps < n_outputs leaves steps_bt_output, and so `done`, at zero: the reference panics in
            // Duration::from_secs_f64 on the NaN; here the line goes out without an estimate)
            if (i > 0 && s->steps_bt_output > 0) {
                const double rt = elapsed(), done = (double)(i * s->steps_bt_output), total = (double)(s->n_outputs * s->steps_bt_output);
                printf("Output %4zu at t = %s, RT = %s, ETTC = %s...\n", i, simulation_time(s->t).c_str(), pretty(rt).c_str(),
                       pretty(rt * (total - done) / done).c_str());
            } else {
                printf("Output %4zu at t = %s...\n", i, simulation_time(s->t).c_str());
            }
            fflush(stdout);
        }
        if (int rc = opal_sim_step(s, s->steps_bt_output)) return rc;
        if (int rc = opal_sim_sync(s)) return rc;
    }
    if (int rc = write_all(s->n_outputs)) return rc;
    if (s->rank == 0)  // main.rs:278-283
        printf("Output %4zu at t = %s, RT = %s\n", s->n_outputs, simulation_time(s->t).c_str(), pretty(elapsed()).c_str());
    return OPAL_OK;
}

double opal_sim_time(const opal_sim_t *s) { return s ? s->t : 0.0; }
double opal_sim_dt(const opal_sim_t *s) { return s ? s->dt : 0.0; }
size_t opal_sim_total_steps(const opal_sim_t *s) { return s ? s->total_steps : 0; }
size_t opal_sim_steps_between_outputs(const opal_sim_t *s) { return s ? s->steps_bt_output : 0; }
size_t opal_sim_n_outputs(const opal_sim_t *s) { return s ? s->n_outputs : 0; }
opal_grid_t *opal_sim_grid(opal_sim_t *s) { return s ? s->grid : nullptr; }
opal_pop_t *opal_sim_electrons(opal_sim_t *s) { return s ? s->electrons : nullptr; }
opal_pop_t *opal_sim_ions(opal_sim_t *s) { return s ? s->ions : nullptr; }
opal_pop_t *opal_sim_photons(opal_sim_t *s) { return s ? s->photons : nullptr; }
int opal_sim_set_fused(opal_sim_t *s, int fused)
{
    if (!s) return OPAL_ERR_ARG;
    s->fused = fused != 0;
    return OPAL_OK;
}
int opal_sim_set_overlap(opal_sim_t *s, int on)
{
    if (!s) return OPAL_ERR_ARG;
    s->overlap = on != 0;
    // the edge / bulk split of an overlapped step reads the host's copies of the cell map and the chunk table: the
    // re-binnings of these populations then wait for the device (particles.hip: do_sort)
    for (opal_pop_t *p : {s->electrons, s->ions})
        if (p) p->want_host_maps = s->overlap;
    return OPAL_OK;
}
size_t opal_sim_overlapped_steps(const opal_sim_t *s) { return s ? s->n_overlapped : 0; }

int opal_sim_set_recut(opal_sim_t *s, size_t every, double threshold, size_t max_shift)
{
    if (!s || (every && !(threshold >= 1.0))) {
        set_error("opal_sim_set_recut: bad arguments (threshold is max / mean of the subdomains' load: >= 1)");
        return OPAL_ERR_ARG;
    }
    s->recut_every = every;
    if (const char *e = getenv("OPAL_RECUT_DRIFT_MIN")) s->drift_min = atof(e);
    if (every) {
        s->recut_threshold = threshold;
        s->recut_max_shift = max_shift ? max_shift : 64;
    }
    return OPAL_OK;
}

int opal_sim_drift_stats(opal_sim_t *s, double *cells_per_step_of_every_cut, size_t *drift_steps)
{
    if (!s) {
        set_error("null simulation handle");
        return OPAL_ERR_ARG;
    }
    if (cells_per_step_of_every_cut)
        for (int i = 0; i <= s->numtasks; i++)
            cells_per_step_of_every_cut[i] = s->drift_v.size() == (size_t)s->numtasks + 1 ? s->drift_v[(size_t)i] : 0.0;
    if (drift_steps) *drift_steps = s->drift_shifts;
    return OPAL_OK;
}

int opal_sim_recut_now(opal_sim_t *s)
{
    if (!s) {
        set_error("null simulation handle");
        return OPAL_ERR_ARG;
    }
    return sim_recut(s, true);
}

int opal_sim_recut_stats(opal_sim_t *s, size_t *checks, size_t *recuts, double *imbalance_last, double *imbalance_worst,
                         size_t *nx_of_every_subdomain, double *seconds)
{
    if (!s) {
        set_error("null simulation handle");
        return OPAL_ERR_ARG;
    }
    if (checks) *checks = s->recut_checks;
    if (recuts) *recuts = s->recuts;
    if (imbalance_last) *imbalance_last = s->imbalance_seen;
    if (imbalance_worst) *imbalance_worst = s->imbalance_worst;
    s->imbalance_worst = 0.0;
    if (seconds) *seconds = s->recut_seconds;
    if (nx_of_every_subdomain)
        for (size_t r = 0; r < s->nxs.size(); r++) nx_of_every_subdomain[r] = s->nxs[r];
    return OPAL_OK;
}

// -------------------------------------------------------------------- config
struct opal_config {
    Config cfg;
};

int opal_config_from_file(opal_config_t **out, const char *path)
{
    std::unique_ptr<opal_config_t> c(new opal_config_t());
    ConfigError ce;
    if (!out || !path || !Config::from_file(path, c->cfg, ce)) {
        ce.kind = ConfigErrorKind::MissingFile;
        return config_fail(ce);
    }
    c->cfg.with_context("constants");
    *out = c.release();
    return OPAL_OK;
}

int opal_config_from_string(opal_config_t **out, const char *text)
{
    std::unique_ptr<opal_config_t> c(new opal_config_t());
    ConfigError ce;
    if (!out || !text || !Config::from_string(text, c->cfg, ce)) {
        ce.kind = ConfigErrorKind::MissingFile;
        return config_fail(ce);
    }
    c->cfg.with_context("constants");
    *out = c.release();
    return OPAL_OK;
}

int opal_config_free(opal_config_t *c)
{
    delete c;
    return OPAL_OK;
}

int opal_config_contains(const opal_config_t *c, const char *section) { return c && c->cfg.contains(section) ? 1 : 0; }

int opal_config_read_f64(const opal_config_t *c, const char *section, const char *field, double *out)
{
    ConfigError ce;
    if (!c || !c->cfg.read_f64(section, field, *out, ce)) return config_fail(ce);
    return OPAL_OK;
}

int opal_config_read_usize(const opal_config_t *c, const char *section, const char *field, size_t *out)
{
    ConfigError ce;
    if (!c || !c->cfg.read_usize(section, field, *out, ce)) return config_fail(ce);
    return OPAL_OK;
}

int opal_config_read_bool(const opal_config_t *c, const char *section, const char *field, int *out)
{
    ConfigError ce;
    bool b = false;
    if (!c || !c->cfg.read_bool(section, field, b, ce)) return config_fail(ce);
    *out = b ? 1 : 0;
    return OPAL_OK;
}

int opal_config_eval(const opal_config_t *c, const char *section, const char *field, int nargs,
                     const char *const *arg_names, const double *arg_values, double *out)
{
    ConfigError ce;
    std::vector<std::string> names;
    for (int i = 0; i < nargs; i++) names.emplace_back(arg_names[i]);
    std::function<double(const double *)> f;
    if (!c || !c->cfg.func(section, field, names, f, ce)) return config_fail(ce);
    *out = f(arg_values);
    return OPAL_OK;
}

int opal_config_eval_expr(const opal_config_t *c, const char *expr, double *out)
{
    std::string err;
    if (!c || !c->cfg.eval(expr, *out, err)) {
        set_error("could not evaluate \"%s\": %s", expr ? expr : "", err.c_str());
        return OPAL_ERR_CONFIG;
    }
    return OPAL_OK;
}

// ----------------------------------------------------------------------- link
// Multi-subdomain exchange is wired in link.cpp.

}  // extern "C"
