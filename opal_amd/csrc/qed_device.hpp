// Quantum synchrotron emission rate on the device (photon_emission.rs:8-79): what the electron push
// needs to run down the optical depth against emission (electron.rs:284-288).
#pragma once
#include "pic_device.hpp"

#include <cstdint>

namespace opal {

constexpr double ALPHA_FINE = 7.29735257e-3;     // constants.rs
constexpr double COMPTON_TIME = 1.28808867e-21;  // constants.rs

// photon_emission.rs:8: the table's spacing in ln(chi), ln(10)/10 truncated as in the reference
constexpr double DELTA_LN_CHI = 0.230258509299;

// ln h(chi) at ln chi = ln(0.01) + i ln(10)/10, i = 0..40, with h the dimensionless emission rate
// (h -> 5 pi / 3 as chi -> 0).  These 41 values define the reference's rate between chi = 0.01 and
// 100 (photon_emission.rs:13-55, 9 significant digits) and are restated here as data, like the
// physical constants, so that optical depths agree with the reference's to rounding;
// tests/test_qed_oracle.py checks them against (2/3) int dN/du du / chi by quadrature.
__device__ const double QED_LN_H[41] = {
    1.64660829e+0, 1.64437994e+0, 1.64162210e+0, 1.63822181e+0, 1.63404803e+0, 1.62895132e+0, 1.62276454e+0,
    1.61530492e+0, 1.60637755e+0, 1.59578044e+0, 1.58331053e+0, 1.56877061e+0, 1.55197597e+0, 1.53276058e+0,
    1.51098175e+0, 1.48652339e+0, 1.45929745e+0, 1.42924408e+0, 1.39633080e+0, 1.36055104e+0, 1.32192247e+0,
    1.28048533e+0, 1.23630061e+0, 1.18944834e+0, 1.14002554e+0, 1.08814410e+0, 1.03392829e+0, 9.77512163e-1,
    9.19036694e-1, 8.58647059e-1, 7.96489918e-1, 7.32710951e-1, 6.67452686e-1, 6.00852676e-1, 5.33042051e-1,
    4.64144437e-1, 3.94275229e-1, 3.23541171e-1, 2.52040210e-1, 1.79861566e-1, 1.07085976e-1,
};
constexpr double QED_LN_H_CHI0 = -4.60517019e+0;  // photon_emission.rs:14, first abscissa as printed there

// photon_emission::rate, photon_emission.rs:59-79, per second
__device__ inline double qed_rate(double chi, double gamma)
{
    double h;
    if (chi < 0.01) {
        h = 5.0 * (M_PI / 3.0) * (1.0 - 8.0 * chi / (5.0 * sqrt(3.0)));
    } else if (chi >= 100.0) {
        double c = cbrt(chi);
        h = -1019.4661473121777 + 1786.716527650374 * c * c;
        h = 1750.6263395722715 + c * c * h;
        h = -2260.1819695887225 + c * h;
        h = 0.00296527643253334 * h / (chi * chi);
    } else {
        double index = (log(chi) - QED_LN_H_CHI0) / DELTA_LN_CHI;
        double fl = floor(index);
        double weight = index - fl;
        int i = min(max((int)fl, 0), 39);  // the reference asserts index < 40 (:72)
        h = exp((1.0 - weight) * QED_LN_H[i] + weight * QED_LN_H[i + 1]);
    }
    return sqrt(3.0) * ALPHA_FINE * chi * h / (2.0 * M_PI * gamma * COMPTON_TIME);
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11)
__device__ inline void philox(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ inline double u53(uint32_t hi, uint32_t lo)
{
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}
// the four uniforms of one emission, in the order electron.rs:209-218 draws them: new optical depth,
// rand1 (energy), rand2 (polar angle), rand3 (azimuth)
__device__ inline void emission_randoms(uint64_t seed, uint64_t id, uint32_t step, double (&out)[4])
{
    uint32_t c[4] = {(uint32_t)id, (uint32_t)(id >> 32), step, 0u};
    philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    out[0] = u53(c[0], c[1]);
    out[1] = u53(c[2], c[3]);
    uint32_t d[4] = {(uint32_t)id, (uint32_t)(id >> 32), step, 1u};
    philox(d, (uint32_t)seed, (uint32_t)(seed >> 32));
    out[2] = u53(d[0], d[1]);
    out[3] = u53(d[2], d[3]);
}

// the optical depths a photon is born with (tau_abs, tau_st: photon.rs:126-133; tau[0..2] are never read): two more
// uniforms of the stream keyed by (seed, parent electron id, step)
__device__ inline void birth_randoms(uint64_t seed, uint64_t id, uint32_t step, double (&out)[2])
{
    uint32_t c[4] = {(uint32_t)id, (uint32_t)(id >> 32), step, 2u};
    philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    out[0] = u53(c[0], c[1]);
    out[1] = u53(c[2], c[3]);
}
// the three uniforms a photon may need in absorb(): the choice between absorption and stimulated emission when both
// optical depths run out together, a new tau_abs, a new tau_st (photon.rs:341-357).  Keyed by the photon: parent id
// and birth time (an electron emits at most one photon per step), and the step.
__device__ inline void absorption_randoms(uint64_t seed, uint64_t id, double birth_time, uint32_t step, double (&out)[3])
{
    const uint64_t bt = (uint64_t)__double_as_longlong(birth_time);
    const uint32_t k0 = (uint32_t)seed ^ (uint32_t)bt, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(bt >> 32);
    uint32_t c[4] = {(uint32_t)id, (uint32_t)(id >> 32), step, 16u};
    philox(c, k0, k1);
    out[0] = u53(c[0], c[1]);
    out[1] = u53(c[2], c[3]);
    uint32_t d[4] = {(uint32_t)id, (uint32_t)(id >> 32), step, 17u};
    philox(d, k0, k1);
    out[2] = u53(d[0], d[1]);
}

// A photon created by stimulated emission (interactions.rs:243, 311): its identity and the optical depths it is born
// with.  Keyed by the STIMULATING photon (its id and birth time) and the step -- not by the emitting electron, which
// may radiate spontaneously in the same step (birth_randoms) and may be stimulated by several photons at once: every
// new photon gets an id of its own, so that its later draws (absorption_randoms) are independent of its siblings'.
__device__ inline void stimulated_birth(uint64_t seed, uint64_t parent_id, double parent_birth, uint32_t step, uint64_t &id,
                                        double (&depth)[2])
{
    const uint64_t bt = (uint64_t)__double_as_longlong(parent_birth);
    const uint32_t k0 = (uint32_t)seed ^ (uint32_t)bt, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(bt >> 32);
    uint32_t c[4] = {(uint32_t)parent_id, (uint32_t)(parent_id >> 32), step, 32u};
    philox(c, k0, k1);
    id = ((uint64_t)c[1] << 32) | c[0];
    depth[0] = u53(c[2], c[3]);
    uint32_t d[4] = {(uint32_t)parent_id, (uint32_t)(parent_id >> 32), step, 33u};
    philox(d, k0, k1);
    depth[1] = u53(d[0], d[1]);
}

}  // namespace opal
