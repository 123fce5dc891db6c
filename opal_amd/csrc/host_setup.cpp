// Host-side set-up that the time loop needs: GridDesign (domain partition) and
// the particle loader.  Plain C++ on the CPU; runs once per simulation.
#include "host_setup.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>


namespace opal {

// ---- RNG: xoshiro256** seeded through SplitMix64, 53-bit uniform floats.
// The reference seeds rand_xoshiro's Xoshiro256StarStar with the MPI rank
// (main.rs:53); its rand/rand_distr crates are not vendored, so normal and
// exponential variates here are Box-Muller / inverse-CDF: the initial particle
// state is statistically, not bitwise, that of the reference.
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void Rng::seed(uint64_t sd)
{
    for (int i = 0; i < 4; i++) {
        uint64_t z = (sd += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        s[i] = z ^ (z >> 31);
    }
}

uint64_t Rng::next()
{
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}

double Rng::uniform() { return (double)(next() >> 11) * 0x1.0p-53; }

double Rng::normal()
{
    double u1 = uniform(), u2 = uniform();
    return std::sqrt(-2.0 * std::log(1.0 - u1)) * std::cos(2.0 * 3.14159265358979323846264338327950288 * u2);
}

double Rng::exp1() { return -std::log(1.0 - uniform()); }

// GridDesign::unbalanced, grid/mod.rs:126-151 (note: remainder cells are dropped, :128)
int design_unbalanced(int numtasks, size_t size, double dx, size_t min_subsize, size_t *nx, double *offset)
{
    size_t subsize = size / (size_t)numtasks;
    if (subsize < min_subsize) subsize = min_subsize;
    size_t rt = 0;
    for (int p = 0; p < numtasks; p++) {
        nx[p] = subsize;
        offset[p] = ((double)rt) * dx;
        rt += subsize;
    }
    return 0;
}

// GridDesign::balanced, grid/mod.rs:157-206
int design_balanced(int numtasks, size_t size, double xmin, double dx, size_t min_subsize,
                    const std::function<double(double)> &ne, size_t *nx, double *offset)
{
    if (numtasks == 0) return design_unbalanced(numtasks, size, dx, min_subsize, nx, offset);
    if (size <= min_subsize) return 1;
    size_t m = size - min_subsize;  // the last task keeps at least min_subsize cells (:164)
    std::vector<double> cumsum(m);
    double total = 0.0;
    for (size_t n = 0; n < m; n++) {
        double x = xmin + ((double)n) * dx;
        total += dx * ne(x);
        cumsum[n] = total;
    }
    double target = cumsum[m - 1] / ((double)numtasks);
    size_t start = 0, ndone = 0;
    for (int p = 1; p < numtasks; p++) {
        size_t first = start + min_subsize, i = first;
        bool found = false;
        for (; i < m; i++)
            if (cumsum[i] >= target * ((double)p)) {
                found = true;
                break;
            }
        if (!found) return 1;  // the reference panics here (:179)
        size_t pos = i - first;
        nx[p - 1] = pos + min_subsize;
        ndone += pos + min_subsize;
        start += pos + min_subsize;
    }
    nx[numtasks - 1] = size - ndone;
    size_t rt = 0;
    for (int p = 0; p < numtasks; p++) {
        offset[p] = ((double)rt) * dx;
        rt += nx[p];
    }
    return 0;
}

// Population::new, particle/mod.rs:172-203, into SoA host arrays.
// Draw order per particle: x, then (urand, nrand) for ux, uy, uz, then the
// species' optical depths (one Exp1 for electrons, electron.rs:168-172, kept
// in out.tau: it only matters when photon emission is on; none for ions,
// ion.rs:157-159; four for photons, photon.rs:126-133: the last two, the depths
// against absorption and stimulated emission, are kept in out.tau / out.tau2).
//
// threads > 1: the same particles, bit for bit, from several host threads.  Every particle takes a fixed number of
// draws from the one generator (1 for x, 3 for each (urand, nrand) pair -- Box-Muller spends two uniforms on a normal --,
// 1 / 0 / 4 exponentials), so the generator's state at the start of any cell is the state at the start of the run
// advanced by a known count.  One thread does nothing but advance it (a nanosecond a draw) and hands out the states at
// the block boundaries as it passes them; the others take the blocks in order and do the expensive part -- logarithms,
// cosines, the momentum expressions -- writing straight to each particle's place.  (xoshiro256** has no cheap jump by an
// arbitrary count; advancing it is cheap enough.)
void load_population(int species, size_t npc, const std::function<double(double)> &number_density,
                     const Fn3 &ux, const Fn3 &uy, const Fn3 &uz, double grid_xmin, double dx, size_t ncells,
                     Rng &rng, HostSoA &out, int threads)
{
    std::vector<double> weight(ncells);
    std::vector<size_t> first(ncells + 1, 0);
    for (size_t c = 0; c < ncells; c++) {
        double x = grid_xmin + (((double)c) + 0.5) * dx;
        double nreal = number_density(x) * dx;
        weight[c] = nreal / ((double)npc);
        first[c + 1] = first[c] + (nreal > 0.0 ? npc : 0);
    }
    const size_t base = out.size(), total = first[ncells];
    out.cell.resize(base + total);
    out.x.resize(base + total);
    out.ux.resize(base + total);
    out.uy.resize(base + total);
    out.uz.resize(base + total);
    out.w.resize(base + total);
    const size_t tau_base = out.tau.size(), tau2_base = out.tau2.size();
    if (species == 0 || species == 2) out.tau.resize(tau_base + total);
    if (species == 2) out.tau2.resize(tau2_base + total);
    const size_t draws = 10 + (species == 0 ? 1 : species == 2 ? 4 : 0);
    auto fill = [&](size_t c0, size_t c1, Rng r) {
        for (size_t c = c0; c < c1; c++) {
            for (size_t i = first[c]; i < first[c + 1]; i++) {
                double xf = r.uniform();
                double real_x = grid_xmin + (((double)c) + xf) * dx;
                double ur = r.uniform(), nr = r.normal();
                double vx = ux(real_x, ur, nr);
                ur = r.uniform(); nr = r.normal();
                double vy = uy(real_x, ur, nr);
                ur = r.uniform(); nr = r.normal();
                double vz = uz(real_x, ur, nr);
                if (species == 0) out.tau[tau_base + i] = r.exp1();
                else if (species == 2) {
                    // photon.rs:126-133: tau[0], tau[1] (never read), tau_abs, tau_st
                    (void)r.exp1();
                    (void)r.exp1();
                    out.tau[tau_base + i] = r.exp1();
                    out.tau2[tau2_base + i] = r.exp1();
                }
                out.cell[base + i] = (int32_t)c;
                out.x[base + i] = xf;
                out.ux[base + i] = vx;
                out.uy[base + i] = vy;
                out.uz[base + i] = vz;
                out.w[base + i] = weight[c];
            }
        }
        return r;
    };
    if (threads <= 1 || total < 200000) {
        rng = fill(0, ncells, rng);
        return;
    }
    // blocks of whole cells holding about the same number of particles each, many more than threads (the tail is short)
    const size_t nblk_want = (size_t)threads * 16;
    std::vector<size_t> cut{0};
    for (size_t b = 1; b < nblk_want; b++) {
        size_t c = (size_t)(std::lower_bound(first.begin(), first.end(), total / nblk_want * b) - first.begin());
        if (c > ncells) c = ncells;
        if (c > cut.back()) cut.push_back(c);
    }
    if (cut.back() < ncells) cut.push_back(ncells);
    const size_t nblk = cut.size() - 1;
    std::vector<Rng> start(nblk);
    std::atomic<size_t> ready{0}, next{0};
    std::vector<std::thread> workers;
    for (int t = 0; t < threads - 1; t++)
        workers.emplace_back([&]() {
            for (;;) {
                const size_t b = next.fetch_add(1);
                if (b >= nblk) return;
                while (ready.load(std::memory_order_acquire) <= b) std::this_thread::yield();
                fill(cut[b], cut[b + 1], start[b]);
            }
        });
    Rng r = rng;
    for (size_t b = 0; b < nblk; b++) {
        start[b] = r;
        ready.store(b + 1, std::memory_order_release);
        for (size_t k = (first[cut[b + 1]] - first[cut[b]]) * draws; k; k--) (void)r.next();
    }
    rng = r;
    // (the thread that advanced the generator helps with what is left)
    for (;;) {
        const size_t b = next.fetch_add(1);
        if (b >= nblk) break;
        fill(cut[b], cut[b + 1], start[b]);
    }
    for (std::thread &w : workers) w.join();
}

static int g_loader_threads = 0;  // 0: automatic for the driver's own expressions, one thread for a caller's callbacks
void set_loader_threads(int n) { g_loader_threads = n < 0 ? 0 : n; }
int loader_threads(bool own_expressions)
{
    if (g_loader_threads > 0) return g_loader_threads;
    if (!own_expressions) return 1;  // a host language's callbacks need not be callable from other threads
    unsigned hw = std::thread::hardware_concurrency();
    return (int)std::min(16u, std::max(1u, hw));
}

}  // namespace opal
