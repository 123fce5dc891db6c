// Host-side set-up shared by the C ABI and the YAML driver.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace opal {

struct Rng {
    uint64_t s[4];
    void seed(uint64_t sd);
    uint64_t next();
    double uniform();
    double normal();
    double exp1();
};

using Fn3 = std::function<double(double, double, double)>;

struct HostSoA {
    std::vector<int32_t> cell;
    std::vector<double> x, ux, uy, uz, w;
    std::vector<double> tau;  // electrons: the optical depth against emission drawn at load time (electron.rs:168-172);
                              // photons: the one against absorption (photon.rs:131)
    std::vector<double> tau2; // photons: the optical depth against stimulated emission (photon.rs:132)
    size_t size() const { return cell.size(); }
};

int design_unbalanced(int numtasks, size_t size, double dx, size_t min_subsize, size_t *nx, double *offset);
int design_balanced(int numtasks, size_t size, double xmin, double dx, size_t min_subsize,
                    const std::function<double(double)> &ne, size_t *nx, double *offset);
// threads > 1: the same particles from several host threads (host_setup.cpp)
void load_population(int species, size_t npc, const std::function<double(double)> &number_density, const Fn3 &ux,
                     const Fn3 &uy, const Fn3 &uz, double grid_xmin, double dx, size_t ncells, Rng &rng,
                     HostSoA &out, int threads = 1);
// how many host threads a loader uses: what opal_set_loader_threads asked for, else the cores there are (16 at the most)
// for the driver's own compiled expressions and one for a caller's callbacks
void set_loader_threads(int n);
int loader_threads(bool own_expressions);

}  // namespace opal
