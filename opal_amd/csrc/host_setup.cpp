// Host-side set-up that the time loop needs: GridDesign (domain partition) and
// the particle loader.  Plain C++ on the CPU; runs once per simulation.
#include "host_setup.hpp"

#include <cmath>
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
void load_population(int species, size_t npc, const std::function<double(double)> &number_density,
                     const Fn3 &ux, const Fn3 &uy, const Fn3 &uz, double grid_xmin, double dx, size_t ncells,
                     Rng &rng, HostSoA &out)
{
    for (long c = 0; c < (long)ncells; c++) {
        double x = grid_xmin + (((double)c) + 0.5) * dx;
        double nreal = number_density(x) * dx;
        double weight = nreal / ((double)npc);
        size_t nsuper = nreal > 0.0 ? npc : 0;
        for (size_t i = 0; i < nsuper; i++) {
            double xf = rng.uniform();
            double real_x = grid_xmin + (((double)c) + xf) * dx;
            double ur = rng.uniform(), nr = rng.normal();
            double vx = ux(real_x, ur, nr);
            ur = rng.uniform(); nr = rng.normal();
            double vy = uy(real_x, ur, nr);
            ur = rng.uniform(); nr = rng.normal();
            double vz = uz(real_x, ur, nr);
            if (species == 0) out.tau.push_back(rng.exp1());
            else if (species == 2) {
                // photon.rs:126-133: tau[0], tau[1] (never read), tau_abs, tau_st
                (void)rng.exp1();
                (void)rng.exp1();
                out.tau.push_back(rng.exp1());
                out.tau2.push_back(rng.exp1());
            }
            out.cell.push_back((int32_t)c);
            out.x.push_back(xf);
            out.ux.push_back(vx);
            out.uy.push_back(vy);
            out.uz.push_back(vz);
            out.w.push_back(weight);
        }
    }
}

}  // namespace opal
