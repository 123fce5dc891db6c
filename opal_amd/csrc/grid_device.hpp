// YeeGrid::synchronize's device side, shared by grid.hip (its own launch) and particles.hip (the fused linked step runs
// it as one more workgroup of the unpack launch: the two are independent and each is a 5 us kernel on its own).
#pragma once
#include "opal_internal.hpp"
#ifdef __HIPCC__
#include "pic_device.hpp"
#endif

namespace opal {
struct SyncArgs {
    double *eb;
    double *jr;
    size_t lenp;
    int len, L, R;
    const double *recv_left;   // 80 doubles or null
    const double *recv_right;
    int self_periodic;         // a lone periodic subdomain receives its own cells (yee.rs:365-370, 390-395)
    int left_bdy, right_bdy;
    double laser_y, laser_z;   // already 2 r laser(t, x), yee.rs:459-461
    const double *damp;        // (1 - sigma) for the last R-1 cells of an absorbing layer
};


#ifdef __HIPCC__
constexpr int HALO_CELLS = 2 * GHOST_SIZE;    // yee.rs:353-354
constexpr int HALO_COMPS = 10;                // rho j[3] E[3] B[3]
constexpr int HALO_DOUBLES = HALO_CELLS * HALO_COMPS;

// yee.rs:411-495.  One workgroup; the received cells are staged in registers
// before anything is written, so a self-exchange may alias send and recv.
__device__ inline void synchronize_body(const SyncArgs &a)
{
    int t = threadIdx.x;
    // (i) ghost overlay, yee.rs:424-452 + Cell::overlay_ghost / overlay (:97-113).  Every lane first reads
    // the value it received (for a self-exchange: straight from the other end of the arrays, which are
    // both source and destination), then all lanes write.
    int side = t / HALO_DOUBLES, r = t % HALO_DOUBLES;
    int k = r / HALO_COMPS, comp = r % HALO_COMPS;
    bool active = false;
    double v = 0.0;
    if (t < 2 * HALO_DOUBLES) {
        if (a.self_periodic) {
            // recv_left = what this subdomain sent right = cell[len-8+k]; recv_right = cell[k]
            int src = side == 0 ? a.len - HALO_CELLS + k : k;
            v = comp == 0 ? a.jr[3 * a.lenp + src]
                : comp < 4 ? a.jr[(size_t)(comp - 1) * a.lenp + src]
                           : a.eb[(size_t)(comp - 4) * a.lenp + src];
            active = true;
        } else {
            const double *recv = side == 0 ? a.recv_left : a.recv_right;
            if (recv) {
                v = recv[r];
                active = true;
            }
        }
    }
    __syncthreads();
    if (active) {
        int cellidx = side == 0 ? k : a.len - HALO_CELLS + k;
        bool ghost = side == 0 ? (k < GHOST_SIZE) : (k >= GHOST_SIZE);
        if (comp == 0) a.jr[3 * a.lenp + cellidx] += v;
        else if (comp < 4) a.jr[(size_t)(comp - 1) * a.lenp + cellidx] += v;
        else if (ghost) a.eb[(size_t)(comp - 4) * a.lenp + cellidx] = v;
    }
    __syncthreads();
    // (ii) laser, yee.rs:456-462: one cell, start = left_bdy_size - 2
    if (a.left_bdy == OPAL_BDY_LASER && t == 0) {
        int start = a.L - 2;
        a.eb[1 * a.lenp + start] += a.laser_y;
        a.eb[2 * a.lenp + start] += a.laser_z;
    }
    // (iii) absorbing layer, yee.rs:464-479
    if (a.right_bdy == OPAL_BDY_ABSORBING) {
        int first = a.len - a.R + 1;
        for (int i = first + t; i < a.len; i += blockDim.x) {
            double f = a.damp[i - first];
            bool zero = i >= a.len - 2;
            for (int c = 0; c < 6; c++) {
                double v = f * a.eb[(size_t)c * a.lenp + i];
                a.eb[(size_t)c * a.lenp + i] = zero ? 0.0 : v;
            }
        }
    } else if (a.right_bdy == OPAL_BDY_CONDUCTING) {  // yee.rs:480-495
        int start = a.len - a.R;
        // every source index is < start + 1 and every destination > start except E[1], E[2], B[0]
        // at start + 1 which read cell `start` itself: no destination is another lane's source.
        if (t == 0) {
            a.eb[0 * a.lenp + start] = 0.0;
            a.eb[4 * a.lenp + start] = 0.0;
            a.eb[5 * a.lenp + start] = 0.0;
        }
        __syncthreads();
        if (t >= 1 && t < a.R) {
            int i = t;
            a.eb[0 * a.lenp + start + i] = -a.eb[0 * a.lenp + start - i];
            a.eb[1 * a.lenp + start + i] = a.eb[1 * a.lenp + start + 1 - i];
            a.eb[2 * a.lenp + start + i] = a.eb[2 * a.lenp + start + 1 - i];
            a.eb[3 * a.lenp + start + i] = a.eb[3 * a.lenp + start + 1 - i];
            a.eb[4 * a.lenp + start + i] = -a.eb[4 * a.lenp + start - i];
            a.eb[5 * a.lenp + start + i] = -a.eb[5 * a.lenp + start - i];
        }
    }
}
#endif
}  // namespace opal
