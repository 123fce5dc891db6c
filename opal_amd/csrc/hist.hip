// Distribution-function output: Population::write_data (particle/mod.rs:383-568) and
// Histogram::generate_1d / generate_2d / write_fits (particle/hgram.rs:168-425).
//
// The particle accessors, the min/max pass and the weighted binning run on the GPU over the SoA
// arrays; spec parsing, bin-count rules, normalisation and the FITS file are host code.  The
// multi-subdomain reductions of the reference (MPI all-reduce of min, max, count, total and bins,
// hgram.rs:190-192,249,260) are left to the caller through the two-phase C ABI
// (opal_pop_hist_range -> reduce -> opal_pop_hist_fill -> reduce -> opal_hist_write_fits);
// opal_pop_write_data is the single-subdomain composition.
#include "opal_internal.hpp"
#include "pic_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace opal;

namespace {

enum { ACC_X = 0, ACC_R, ACC_ENERGY, ACC_PX, ACC_PY, ACC_PZ, ACC_PPERP, ACC_THETA, ACC_PHI, ACC_LONGITUDE, ACC_LATITUDE,
       ACC_WORK, ACC_CHI, ACC_NONE };
enum { BIN_AUTO = 0, BIN_LOG = 1, BIN_FIXED_N = 2, BIN_FIXED_SIZE = 3 };  // hgram.rs:8-14
enum { W_WEIGHT = 0, W_ENERGY = 1 };

struct AccArgs {
    const int *cell;
    const double *x, *ux, *uy, *uz, *w;
    const double *chi, *work, *y, *z;  // electron extras or null
    const PopCounters *ctr;
    int species;
    double xmin, dx;      // grid.xmin(), grid.dx()
    double mass_mev;      // momentum unit: 0.510999 (electron.rs:157-160, photon.rs:211-214) or (A m_p / m_e) 0.510999 (ion.rs:140-143)
    int acc[2];
    int log[2];
    int dim;
    int weight_kind;
};

__device__ inline double particle_energy(const AccArgs &a, double u2)
{
    double g = sqrt(1.0 + u2);
    if (a.species == OPAL_ELECTRON) return g * ELECTRON_MASS_MEV;  // electron.rs:118-120
    if (a.species == OPAL_PHOTON) return sqrt(u2) * ELECTRON_MASS_MEV;  // photon.rs:222-224
    return (u2 / (1.0 + g)) * a.mass_mev;                          // ion.rs:128-130
}

// the accessor closures of mod.rs:388-449
__device__ inline double access(const AccArgs &a, int id, size_t i)
{
    double px = a.ux[i] * a.mass_mev, py = a.uy[i] * a.mass_mev, pz = a.uz[i] * a.mass_mev;
    switch (id) {
    case ACC_X: return a.xmin + (((double)a.cell[i]) + a.x[i]) * a.dx;
    case ACC_R: return (a.species != OPAL_ION && a.y) ? hypot(a.y[i], a.z[i]) : 0.0;  // ion.rs has no y / z
    case ACC_ENERGY: return particle_energy(a, a.ux[i] * a.ux[i] + a.uy[i] * a.uy[i] + a.uz[i] * a.uz[i]);
    case ACC_PX: return px;
    case ACC_PY: return py;
    case ACC_PZ: return pz;
    case ACC_PPERP: return hypot(py, pz);
    case ACC_THETA: return acos(px / sqrt(px * px + py * py + pz * pz));
    case ACC_PHI: return atan2(pz, py);
    case ACC_LONGITUDE: return atan2(py, -px);
    case ACC_LATITUDE: return asin(pz / sqrt(px * px + py * py + pz * pz));
    case ACC_WORK: return a.work ? a.work[i] : 0.0;
    case ACC_CHI: return a.chi ? a.chi[i] : 0.0;
    default: return 0.0;
    }
}

// min_max_by (hgram.rs:105-127) per workgroup; the host folds the partials.  Like the reference, the
// running min/max start from the first particle's value whether or not it is finite: the host seeds
// the fold with that value (slot 0 of block 0 is written separately).
__global__ void k_hist_range(AccArgs a, double *__restrict__ part /* [blocks][4] */, double *__restrict__ first /* [2] */)
{
    __shared__ double s[4][256];
    unsigned n = a.ctr->n_slots;
    double mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (a.cell[i] == DEAD_CELL) continue;
        for (int d = 0; d < a.dim; d++) {
            double v = access(a, a.acc[d], i);
            if (a.log[d]) v = log(v);
            if (isfinite(v)) {
                mn[d] = fmin(mn[d], v);
                mx[d] = fmax(mx[d], v);
            }
        }
    }
    s[0][threadIdx.x] = mn[0]; s[1][threadIdx.x] = mx[0]; s[2][threadIdx.x] = mn[1]; s[3][threadIdx.x] = mx[1];
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) {
            s[0][threadIdx.x] = fmin(s[0][threadIdx.x], s[0][threadIdx.x + k]);
            s[1][threadIdx.x] = fmax(s[1][threadIdx.x], s[1][threadIdx.x + k]);
            s[2][threadIdx.x] = fmin(s[2][threadIdx.x], s[2][threadIdx.x + k]);
            s[3][threadIdx.x] = fmax(s[3][threadIdx.x], s[3][threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = s[threadIdx.x][0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // value of the first live particle (the reference's initial min = max, hgram.rs:109-110)
        first[0] = first[1] = NAN;
        for (size_t i = 0; i < n; i++) {
            if (a.cell[i] == DEAD_CELL) continue;
            for (int d = 0; d < a.dim; d++) {
                double v = access(a, a.acc[d], i);
                first[d] = a.log[d] ? log(v) : v;
            }
            break;
        }
    }
}

struct FillArgs {
    double gmin[2], bin_sz[2];
    unsigned nbins[2];
    unsigned len;
    int density;  // hspec is Density or ProbabilityDensity: log-scaled axes reweight (hgram.rs:236-238)
    double *cts;
    double *total_part;  // [blocks]
};

// linear_bin_vol, hgram.rs:129-131
__device__ inline double linear_bin_vol(double mn, double bin_sz, double bin) { return exp(mn + bin * bin_sz) * expm1(bin_sz); }

// Rust's saturating `f64 as usize`
__device__ inline double as_usize(double v) { return v != v ? 0.0 : (v < 0.0 ? 0.0 : (v > 1.8e19 ? 1.8e19 : floor(v))); }

// the binning loops of hgram.rs:218-247 (1-D) and :312-345 (2-D)
__global__ void k_hist_fill(AccArgs a, FillArgs f)
{
    __shared__ double s[256];
    unsigned n = a.ctr->n_slots;
    double total = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (a.cell[i] == DEAD_CELL) continue;
        double w = a.w[i];
        if (a.weight_kind == W_ENERGY)
            w = particle_energy(a, a.ux[i] * a.ux[i] + a.uy[i] * a.uy[i] + a.uz[i] * a.uz[i]) * w;
        total += w;  // "count everything, even if not binned"
        double v[2] = {0.0, 0.0};
        bool finite = true;
        for (int d = 0; d < a.dim; d++) {
            v[d] = access(a, a.acc[d], i);
            if (a.log[d]) v[d] = log(v[d]);
            finite = finite && isfinite(v[d]);
        }
        if (!finite) continue;
        double fbin;
        if (a.dim == 1) {
            double bin = as_usize(floor((v[0] - f.gmin[0]) / f.bin_sz[0]));  // bin_vol; division by 0 -> NaN/inf -> saturates
            if (a.log[0] && f.density) w = w * f.bin_sz[0] / linear_bin_vol(f.gmin[0], f.bin_sz[0], bin);
            fbin = bin;
        } else {
            double b0 = f.bin_sz[0] == 0.0 ? 0.0 : as_usize(floor((v[0] - f.gmin[0]) / f.bin_sz[0]));
            double b1 = f.bin_sz[1] == 0.0 ? 0.0 : as_usize(floor((v[1] - f.gmin[1]) / f.bin_sz[1]));
            if (a.log[0] && f.density) w *= f.bin_sz[0] == 0.0 ? 1.0 : f.bin_sz[0] / linear_bin_vol(f.gmin[0], f.bin_sz[0], b0);
            if (a.log[1] && f.density) w *= f.bin_sz[1] == 0.0 ? 1.0 : f.bin_sz[1] / linear_bin_vol(f.gmin[1], f.bin_sz[1], b1);
            fbin = b1 * (double)f.nbins[1] + b0;  // sic: hgram.rs:342
        }
        if (fbin < (double)f.len) atomicAdd(&f.cts[(unsigned)fbin], w);
    }
    s[threadIdx.x] = total;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) f.total_part[blockIdx.x] = s[0];
}

struct Spec {
    int dim = 0;
    int acc[2] = {ACC_NONE, ACC_NONE};
    std::string name[2], unit[2], weight;
    int bkind = BIN_AUTO;
    size_t fixed_n = 0;
    double fixed_dx = 0.0;
    int weight_kind = W_WEIGHT;
    bool valid = false;
};

bool rust_usize(const std::string &s, size_t &out)
{
    if (s.empty()) return false;
    size_t i = s[0] == '+' ? 1 : 0;
    if (i >= s.size()) return false;
    for (size_t k = i; k < s.size(); k++)
        if (!isdigit((unsigned char)s[k])) return false;
    out = (size_t)strtoull(s.c_str() + i, nullptr, 10);
    return true;
}

bool rust_f64(const std::string &s, double &out)
{
    char *end = nullptr;
    if (s.empty() || isspace((unsigned char)s[0])) return false;
    out = strtod(s.c_str(), &end);
    return end && *end == 0 && end != s.c_str();
}

// mod.rs:451-521
Spec parse_spec(const std::string &o)
{
    Spec sp;
    std::vector<std::string> ss;
    size_t start = 0;
    for (;;) {
        size_t c = o.find(':', start);
        ss.push_back(o.substr(start, c == std::string::npos ? std::string::npos : c - start));
        if (c == std::string::npos) break;
        start = c + 1;
    }
    std::string bspec = "auto";
    sp.weight = "weight";
    if (ss.size() >= 2 && !ss.back().empty() && ss.back().front() == '(' && ss.back().back() == ')') {
        std::string last = ss.back();
        ss.pop_back();
        size_t b = last.find_first_not_of('('), e = last.find_last_not_of(')');
        last = (b == std::string::npos) ? "" : last.substr(b, e - b + 1);
        std::vector<std::string> parts;
        size_t st = 0;
        for (;;) {
            size_t c = last.find(';', st);
            parts.push_back(last.substr(st, c == std::string::npos ? std::string::npos : c - st));
            if (c == std::string::npos) break;
            st = c + 1;
        }
        if (parts.size() == 1) sp.weight = parts[0];
        else if (parts.size() == 2) { bspec = parts[0]; sp.weight = parts[1]; }
    }
    // BinSpec::from(&str), hgram.rs:27-41
    size_t nb;
    double dxv;
    if (rust_usize(bspec, nb)) { sp.bkind = BIN_FIXED_N; sp.fixed_n = nb; }
    else if (rust_f64(bspec, dxv)) { sp.bkind = BIN_FIXED_SIZE; sp.fixed_dx = dxv; }
    else if (bspec == "log") sp.bkind = BIN_LOG;
    else sp.bkind = BIN_AUTO;
    static const struct { const char *n; int id; const char *unit; } table[] = {
        {"x", ACC_X, "m"}, {"r", ACC_R, "m"}, {"energy", ACC_ENERGY, "MeV"}, {"px", ACC_PX, "MeV/c"}, {"py", ACC_PY, "MeV/c"},
        {"pz", ACC_PZ, "MeV/c"}, {"p_perp", ACC_PPERP, "MeV/c"}, {"theta", ACC_THETA, "rad"}, {"phi", ACC_PHI, "rad"},
        {"longitude", ACC_LONGITUDE, "rad"}, {"latitude", ACC_LATITUDE, "rad"}, {"work", ACC_WORK, "J"}, {"chi", ACC_CHI, "1"}};
    if (ss.size() < 1 || ss.size() > 2) return sp;
    sp.dim = (int)ss.size();
    for (int d = 0; d < sp.dim; d++) {
        bool found = false;
        for (auto &t : table)
            if (ss[d] == t.n) { sp.acc[d] = t.id; sp.name[d] = t.n; sp.unit[d] = t.unit; found = true; }
        if (!found) return sp;
    }
    if (sp.weight == "energy") sp.weight_kind = W_ENERGY;
    else if (sp.weight == "weight" || sp.weight == "auto") sp.weight_kind = W_WEIGHT;
    else return sp;
    sp.valid = true;
    return sp;
}

AccArgs make_acc(const opal_pop_t *p, const opal_grid_t *g, const Spec &sp)
{
    AccArgs a;
    memset(&a, 0, sizeof a);
    int k = p->cur;
    a.cell = p->d_cell[k]; a.x = p->d_x[k]; a.ux = p->d_ux[k]; a.uy = p->d_uy[k]; a.uz = p->d_uz[k]; a.w = p->d_w[k];
    if (p->extras) { a.chi = p->d_ex[EX_CHI]; a.work = p->d_ex[EX_WORK]; a.y = p->d_ex[EX_Y]; a.z = p->d_ex[EX_Z]; }
    a.ctr = p->d_ctr;
    a.species = p->species;
    a.xmin = opal_grid_xmin(g);
    a.dx = g->dx;
    a.mass_mev = p->species != OPAL_ION ? ELECTRON_MASS_MEV  // photon.rs:211-214: k in units of m_e c
                                        : (p->a * PROTON_MASS / ELECTRON_MASS) * ELECTRON_MASS_MEV;
    a.dim = sp.dim;
    for (int d = 0; d < 2; d++) {
        a.acc[d] = sp.acc[d];
        a.log[d] = sp.bkind == BIN_LOG;
    }
    a.weight_kind = sp.weight_kind;
    return a;
}

// number_of_bins, hgram.rs:133-146
size_t number_of_bins(double mn, double mx, size_t n, const Spec &sp)
{
    if (mn == mx) return 1;
    switch (sp.bkind) {
    case BIN_FIXED_N: return sp.fixed_n;
    case BIN_FIXED_SIZE: {
        double v = std::ceil((mx - mn) / sp.fixed_dx);
        return v != v || v < 0 ? 0 : (size_t)v;
    }
    default: return (size_t)std::ceil(2.0 * std::cbrt((double)n));
    }
}

void put_card(std::string &hdr, const std::string &card)
{
    std::string c = card;
    c.resize(80, ' ');
    hdr += c;
}
std::string card_num(const char *key, const std::string &val, const char *comment = nullptr)
{
    char buf[96];
    snprintf(buf, sizeof buf, "%-8s= %20s", key, val.c_str());
    std::string s(buf);
    if (comment) s += std::string(" / ") + comment;
    return s;
}
std::string fits_double(double v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.15G", v);
    std::string s(buf);
    if (s.find('.') == std::string::npos && s.find('E') == std::string::npos && s.find("N") == std::string::npos &&
        s.find("I") == std::string::npos)
        s += ".";
    return s;
}
std::string card_str(const char *key, const std::string &val)
{
    std::string v = val;
    if (v.size() < 8) v.resize(8, ' ');
    char buf[160];
    snprintf(buf, sizeof buf, "%-8s= '%s'", key, v.c_str());
    return buf;
}

}  // namespace

struct opal_hist {
    Spec spec;
    int dim = 0;
    double gmin[2] = {0, 0}, gmax[2] = {0, 0};
    size_t gnum = 0;
    size_t nbins[2] = {1, 1};
    double bin_sz[2] = {0, 0};
    double bin_vol = 1.0;
    double total = 0.0;  // local after fill; global once the caller has reduced it
    std::vector<double> cts;  // raw weighted counts (local); normalised by opal_hist_finish
    bool finished = false;
};

extern "C" {

/* phase 1: local min / max of the accessors (after ln for log-scaled axes) and the local particle count
 * (hgram.rs:175-186).  min[d] = DBL_MAX, max[d] = -DBL_MAX for an empty population.  Returns
 * OPAL_ERR_CONFIG for a spec the reference would silently skip (unknown accessor or weight). */
int opal_pop_hist_range(const opal_pop_t *p, const opal_grid_t *g, const char *spec, int *dim, double *min, double *max,
                        size_t *count)
{
    if (!p || !g || !spec || !dim || !min || !max || !count) {
        set_error("opal_pop_hist_range: bad arguments");
        return OPAL_ERR_ARG;
    }
    Spec sp = parse_spec(spec);
    if (!sp.valid) {
        set_error("output spec \"%s\" names an unknown quantity or weight (the reference skips it)", spec);
        return OPAL_ERR_CONFIG;
    }
    *dim = sp.dim;
    size_t live = 0;
    if (int rc = opal_pop_len(p, &live)) return rc;
    *count = live;
    for (int d = 0; d < 2; d++) {
        min[d] = 1.7976931348623157e308;
        max[d] = -1.7976931348623157e308;
    }
    if (live == 0) return OPAL_OK;
    AccArgs a = make_acc(p, g, sp);
    int nblk = (int)std::min<size_t>((p->n_upper + 255) / 256, 1024);
    double *d_part, *d_first;
    OPAL_HIP(hipMalloc(&d_part, sizeof(double) * 4 * nblk));
    OPAL_HIP(hipMalloc(&d_first, sizeof(double) * 2));
    k_hist_range<<<nblk, 256, 0, p->stream>>>(a, d_part, d_first);
    OPAL_HIP(hipGetLastError());
    std::vector<double> part((size_t)4 * nblk);
    double first[2];
    OPAL_HIP(hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, p->stream));
    OPAL_HIP(hipMemcpyAsync(first, d_first, sizeof first, hipMemcpyDeviceToHost, p->stream));
    OPAL_HIP(hipStreamSynchronize(p->stream));
    hipFree(d_part);
    hipFree(d_first);
    for (int d = 0; d < sp.dim; d++) {
        double mn = first[d], mx = first[d];  // hgram.rs:109-110
        for (int b = 0; b < nblk; b++) {
            double pmn = part[(size_t)b * 4 + 2 * d], pmx = part[(size_t)b * 4 + 2 * d + 1];
            // f64::min / max ignore a NaN operand, as the reference's min.min(v) does
            if (std::isfinite(pmn)) mn = std::fmin(mn, pmn);
            if (std::isfinite(pmx)) mx = std::fmax(mx, pmx);
        }
        min[d] = mn;
        max[d] = mx;
    }
    return OPAL_OK;
}

/* phase 2: given the GLOBAL min / max / count, bins this subdomain's particles (hgram.rs:199-247 /
 * :296-345).  The handle carries the local raw counts and local total weight. */
int opal_pop_hist_fill(const opal_pop_t *p, const opal_grid_t *g, const char *spec, const double *gmin, const double *gmax,
                       size_t gnum, opal_hist_t **out)
{
    if (!p || !g || !spec || !gmin || !gmax || !out) {
        set_error("opal_pop_hist_fill: bad arguments");
        return OPAL_ERR_ARG;
    }
    Spec sp = parse_spec(spec);
    if (!sp.valid) {
        set_error("output spec \"%s\" names an unknown quantity or weight (the reference skips it)", spec);
        return OPAL_ERR_CONFIG;
    }
    if (gnum == 0) {
        set_error("histogram of an empty population (the reference writes nothing, hgram.rs:195-197)");
        return OPAL_ERR_ARG;
    }
    opal_hist_t *h = new opal_hist_t();
    h->spec = sp;
    h->dim = sp.dim;
    h->gnum = gnum;
    for (int d = 0; d < sp.dim; d++) {
        h->gmin[d] = gmin[d];
        h->gmax[d] = gmax[d];
        h->nbins[d] = number_of_bins(gmin[d], gmax[d], gnum, sp);
    }
    if (sp.dim == 1) {  // hgram.rs:202-211, 272
        h->bin_vol = gmin[0] == gmax[0] ? 1.0 : (sp.bkind == BIN_FIXED_SIZE ? sp.fixed_dx : (gmax[0] - gmin[0]) / (double)h->nbins[0]);
        h->bin_sz[0] = h->nbins[0] <= 1 ? 0.0 : h->bin_vol;
    } else {  // bin_size_and_volume, hgram.rs:148-166
        h->bin_vol = 1.0;
        for (int d = 0; d < 2; d++) {
            if (gmin[d] == gmax[d]) h->bin_sz[d] = 0.0;
            else {
                h->bin_sz[d] = sp.bkind == BIN_FIXED_SIZE ? sp.fixed_dx : (gmax[d] - gmin[d]) / (double)h->nbins[d];
                h->bin_vol *= h->bin_sz[d];
            }
        }
    }
    size_t len = sp.dim == 1 ? h->nbins[0] : h->nbins[0] * h->nbins[1];
    if (len > ((size_t)1 << 28)) {
        delete h;
        set_error("histogram with %zu bins is too large", len);
        return OPAL_ERR_CAPACITY;
    }
    h->cts.assign(len, 0.0);
    size_t live = p->n_upper;
    if (live && len) {
        AccArgs a = make_acc(p, g, sp);
        FillArgs f;
        memset(&f, 0, sizeof f);
        for (int d = 0; d < 2; d++) {
            f.gmin[d] = h->gmin[d];
            f.nbins[d] = (unsigned)h->nbins[d];
        }
        // the 1-D loop divides by bin_vol (which is 1 when min == max), the 2-D loop by bin_sz
        f.bin_sz[0] = sp.dim == 1 ? h->bin_vol : h->bin_sz[0];
        f.bin_sz[1] = h->bin_sz[1];
        f.len = (unsigned)len;
        f.density = 1;  // HeightSpec::Density is the only one write_data requests (mod.rs:459-466)
        int nblk = (int)std::min<size_t>((live + 255) / 256, 2048);
        double *d_tot;
        OPAL_HIP(hipMalloc(&f.cts, sizeof(double) * len));
        OPAL_HIP(hipMalloc(&d_tot, sizeof(double) * nblk));
        OPAL_HIP(hipMemsetAsync(f.cts, 0, sizeof(double) * len, p->stream));
        f.total_part = d_tot;
        k_hist_fill<<<nblk, 256, 0, p->stream>>>(a, f);
        OPAL_HIP(hipGetLastError());
        std::vector<double> tot((size_t)nblk);
        OPAL_HIP(hipMemcpyAsync(h->cts.data(), f.cts, sizeof(double) * len, hipMemcpyDeviceToHost, p->stream));
        OPAL_HIP(hipMemcpyAsync(tot.data(), d_tot, sizeof(double) * nblk, hipMemcpyDeviceToHost, p->stream));
        OPAL_HIP(hipStreamSynchronize(p->stream));
        hipFree(f.cts);
        hipFree(d_tot);
        for (double t : tot) h->total += t;
    }
    *out = h;
    return OPAL_OK;
}

size_t opal_hist_len(const opal_hist_t *h) { return h ? h->cts.size() : 0; }
double *opal_hist_counts(opal_hist_t *h) { return h ? h->cts.data() : nullptr; }
double opal_hist_total(const opal_hist_t *h) { return h ? h->total : 0.0; }
int opal_hist_set_total(opal_hist_t *h, double global_total)
{
    if (!h) return OPAL_ERR_ARG;
    h->total = global_total;
    return OPAL_OK;
}
int opal_hist_dims(const opal_hist_t *h, int *dim, size_t *nbins, double *min, double *bin_size)
{
    if (!h) return OPAL_ERR_ARG;
    if (dim) *dim = h->dim;
    for (int d = 0; d < h->dim; d++) {
        if (nbins) nbins[d] = h->nbins[d];
        if (min) min[d] = h->gmin[d];
        if (bin_size) bin_size[d] = h->bin_sz[d];
    }
    return OPAL_OK;
}

/* Density normalisation (hgram.rs:255-259 / :353-357): counts / bin volume.  Linear, so it may be applied
 * before or after the caller sums the counts over subdomains; call it exactly once. */
int opal_hist_finish(opal_hist_t *h)
{
    if (!h || h->finished) {
        set_error("opal_hist_finish: bad or already finished handle");
        return OPAL_ERR_ARG;
    }
    for (double &c : h->cts) c = c / h->bin_vol;
    h->finished = true;
    return OPAL_OK;
}

int opal_hist_free(opal_hist_t *h)
{
    delete h;
    return OPAL_OK;
}

/* Histogram::write_fits, hgram.rs:394-425: a FITS primary image of doubles with the reference's keywords.
 * `path` is the plain file name (the reference's leading '!' only tells CFITSIO to overwrite). */
int opal_hist_write_fits(const opal_hist_t *h, const char *path)
{
    if (!h || !path) {
        set_error("opal_hist_write_fits: bad arguments");
        return OPAL_ERR_ARG;
    }
    const Spec &sp = h->spec;
    std::string hdr;
    put_card(hdr, card_num("SIMPLE", "T", "file does conform to FITS standard"));
    put_card(hdr, card_num("BITPIX", "-64", "number of bits per data pixel"));
    put_card(hdr, card_num("NAXIS", std::to_string(h->dim), "number of data axes"));
    // fitsio's ImageDescription takes dimensions in row-major order: the last one is NAXIS1
    for (int ax = 1; ax <= h->dim; ax++) {
        std::string key = "NAXIS" + std::to_string(ax);
        put_card(hdr, card_num(key.c_str(), std::to_string(h->nbins[h->dim - ax]), "length of data axis"));
    }
    put_card(hdr, card_num("EXTEND", "T", "FITS dataset may contain extensions"));
    for (int i = 0; i < h->dim; i++) {
        std::string n = std::to_string(i + 1);
        put_card(hdr, card_num(("CRPIX" + n).c_str(), fits_double(1.0)));
        put_card(hdr, card_num(("CRVAL" + n).c_str(), fits_double(h->gmin[i] + 0.5 * h->bin_sz[i])));
        put_card(hdr, card_num(("CDELT" + n).c_str(), fits_double(h->bin_sz[i])));
        put_card(hdr, card_str(("CNAME" + n).c_str(), sp.name[i]));
        put_card(hdr, card_str(("CUNIT" + n).c_str(), sp.unit[i]));
    }
    std::string bunit = h->dim == 1 ? "1/" + sp.unit[0] : "1/(" + sp.unit[0] + "." + sp.unit[1] + ")";
    std::string object = h->dim == 1 ? "hgram/density/" + sp.name[0] : "hgram/density/" + sp.name[0] + "_" + sp.name[1];
    put_card(hdr, card_str("BUNIT", bunit));
    put_card(hdr, card_num("TOTAL", fits_double(h->total)));
    put_card(hdr, card_str("OBJECT", object));
    double dmin = 0.0, dmax = 0.0;
    if (!h->cts.empty()) {
        dmin = *std::min_element(h->cts.begin(), h->cts.end());
        dmax = *std::max_element(h->cts.begin(), h->cts.end());
    }
    put_card(hdr, card_num("DATAMIN", fits_double(dmin)));
    put_card(hdr, card_num("DATAMAX", fits_double(dmax)));
    put_card(hdr, "END");
    hdr.resize((hdr.size() + 2879) / 2880 * 2880, ' ');
    std::string data(h->cts.size() * 8, '\0');
    for (size_t i = 0; i < h->cts.size(); i++) {
        uint64_t bits;
        memcpy(&bits, &h->cts[i], 8);
        for (int b = 0; b < 8; b++) data[i * 8 + b] = (char)((bits >> (56 - 8 * b)) & 0xff);
    }
    data.resize((data.size() + 2879) / 2880 * 2880, '\0');
    FILE *fp = fopen(path, "wb");
    if (!fp) {
        set_error("cannot write %s", path);
        return OPAL_ERR_IO;
    }
    fwrite(hdr.data(), 1, hdr.size(), fp);
    fwrite(data.data(), 1, data.size(), fp);
    fclose(fp);
    return OPAL_OK;
}

/* The file name write_data gives a spec (mod.rs:527-559), without the leading '!'.  Returns
 * OPAL_ERR_CONFIG for specs the reference skips. */
int opal_hist_filename(const char *dir, size_t index, const char *pop_name, const char *spec, char *out, size_t len)
{
    Spec sp = parse_spec(spec ? spec : "");
    if (!sp.valid || !dir || !pop_name || !out) {
        set_error("output spec \"%s\" names an unknown quantity or weight (the reference skips it)", spec ? spec : "");
        return OPAL_ERR_CONFIG;
    }
    std::string f = std::string(dir) + "/" + std::to_string(index) + "_" + pop_name + "_" + sp.name[0];
    if (sp.dim == 2) f += "-" + sp.name[1];
    if (sp.weight != "weight") f += "_" + sp.weight;
    if (sp.bkind == BIN_LOG) f += "_log";
    f += ".fits";
    if (f.size() + 1 > len) {
        set_error("opal_hist_filename: buffer too small");
        return OPAL_ERR_ARG;
    }
    memcpy(out, f.c_str(), f.size() + 1);
    return OPAL_OK;
}

/* Population::write_data for a single subdomain: every spec of `specs` (skipping, like the reference,
 * the ones it cannot map) becomes "{dir}/{index}_{name}_{...}.fits". */
int opal_pop_write_data(const opal_pop_t *p, const opal_grid_t *g, const char *dir, size_t index, const char *pop_name,
                        const char *const *specs, size_t nspecs)
{
    if (!p || !g || !dir || !pop_name || (nspecs && !specs)) {
        set_error("opal_pop_write_data: bad arguments");
        return OPAL_ERR_ARG;
    }
    for (size_t k = 0; k < nspecs; k++) {
        char path[4096];
        if (opal_hist_filename(dir, index, pop_name, specs[k], path, sizeof path) != OPAL_OK) continue;  // skipped spec
        int dim;
        double mn[2], mx[2];
        size_t n;
        if (int rc = opal_pop_hist_range(p, g, specs[k], &dim, mn, mx, &n)) return rc;
        if (n == 0) continue;  // hgram.rs:195-197: None -> nothing written
        opal_hist_t *h = nullptr;
        if (int rc = opal_pop_hist_fill(p, g, specs[k], mn, mx, n, &h)) return rc;
        int rc = opal_hist_finish(h);
        if (!rc) rc = opal_hist_write_fits(h, path);
        opal_hist_free(h);
        if (rc) return rc;
    }
    return OPAL_OK;
}

}  // extern "C"
