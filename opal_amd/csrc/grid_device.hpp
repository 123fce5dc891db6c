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


// YeeGrid::advance's device side.  grid.hip's k_advance runs one tile per workgroup over the whole grid.  Launches that
// also run YeeGrid::synchronize (k_sync_advance; the tail kernel of a linked step, particles.hip) split the grid: the
// cells synchronize writes -- the first and last 8, the laser cell, the absorbing or conducting layer -- and their
// stencil neighbours belong to the ONE workgroup that runs synchronize, which advances them right afterwards
// (edge_left cells at the left end, edge_right at the right end); every other tile depends on nothing synchronize
// touches and is advanced by a workgroup of its own, concurrently -- in a linked step even while the neighbours'
// messages are still on their way.
struct AdvArgs {
    const double *src;   // field buffer read (6 x lenp)
    double *dst;         // field buffer written
    const double *jr;    // currents of this step (4 x lenp)
    double *jr_next;     // the spare current block, zeroed for the next step's deposits (YeeGrid::clear, yee.rs:543-549)
    size_t lenp;
    int len;
    double dt, dx;
    int edge_left, edge_right;  // cells the synchronize workgroup advances itself (0, 0: nobody runs synchronize here)
};

// host side (grid.hip)
void grid_sync_args(const opal_grid_t *g, const double *recv_left, const double *recv_right, double ly, double lz, SyncArgs &a);
void grid_advance_args(const opal_grid_t *g, double dt, AdvArgs &a);  // with the edge split for a synchronize workgroup
void grid_advance_done(opal_grid_t *g);  // after a launch that ran the field solve with grid_advance_args
int grid_interior_tiles(const AdvArgs &a);  // workgroups of ADV_TILE cells between the two edges
// (pub: counters the re-binning policy wants published by this launch, or null)
int grid_sync_advance(opal_grid_t *g, const double *recv_left, const double *recv_right, double ly, double lz, double dt,
                      const PublishBatch *pub = nullptr);

constexpr int ADV_TILE = 1024;  // cells per interior workgroup of the launches that also run synchronize
constexpr int ADV_EDGE = 512;   // at most this many cells per end for the synchronize workgroup (half its threads each)

#ifdef __HIPCC__
// called by (at least) the first wave of a workgroup
__device__ inline void publish_counters(const PublishBatch &b)
{
    static_assert(sizeof(PopCounters) == 64, "one word per lane below");
    const int e = threadIdx.x >> 4, w = threadIdx.x & 15;
    if (threadIdx.x < 64) {
        if (e < b.n) reinterpret_cast<volatile unsigned *>(&b.dst[e]->c)[w] = reinterpret_cast<const unsigned *>(b.src[e])[w];
        __threadfence_system();
        if (e < b.n && w == 0) b.dst[e]->seq = b.seq[e];
        if (threadIdx.x == 0 && b.zero) *b.zero = 0u;  // (behind the copy: the published counters still show the count)
    }
}

constexpr int ADV_HALO = 2;  // halo cells recomputed on each side of a tile

// YeeGrid::advance, yee.rs:345-349 = advance_B(dt/2) (:839-848), advance_E(dt) (:852-867), advance_B(dt/2), fused,
// for the cells [s, min(s + T, e)).  Each sub-step reads only the other field, so it is order-independent inside the
// sub-step; a group of T threads (t = the thread's index in it) loads its tile plus a 2-cell halo into LDS
// (5 x (T + 4) doubles), runs the three sub-steps there and writes its tile to the other field buffer.  The array
// ends follow the reference loops: B is not advanced in the last cell, cell 0 gets the Silver-Mueller update.
// Contains workgroup barriers: every thread of the workgroup must call it (groups with nothing to do pass e <= s).
// The three sub-steps and the write-out, for a tile that is in LDS already (rows of T + 4 doubles).  Bx: the tile's row of
// it, or null = taken from the source buffer as it is.
template <int T>
__device__ inline void advance_tile_steps(const AdvArgs &a, int s, int e, int t, double *Ex, double *Ey, double *Ez, double *By,
                                          double *Bz, const double *Bx)
{
    constexpr int FH = ADV_HALO, FW = T + 2 * FH;
    const size_t lenp = a.lenp;
    const int len = a.len;
    const double dt = a.dt, dx = a.dx;
    const bool work = e > s;
    // (the currents the E step needs are requested now, so that their latency overlaps the B half step)
    double jx0 = 0.0, jy0 = 0.0, jz0 = 0.0, jx1 = 0.0, jy1 = 0.0, jz1 = 0.0;
    {
        int i = s - FH + 1 + t;  // the E step's first pass: k = 1 + t
        if (work && i > 0 && i < len) { jx0 = a.jr[0 * lenp + i]; jy0 = a.jr[1 * lenp + i]; jz0 = a.jr[2 * lenp + i]; }
        i += T;                  // ... and its second (k = 1 + t + T < FW - 1 only for t < 2)
        if (work && t < FW - 2 - T && i > 0 && i < len) { jx1 = a.jr[0 * lenp + i]; jy1 = a.jr[1 * lenp + i]; jz1 = a.jr[2 * lenp + i]; }
    }
    __syncthreads();
    const double hdt = 0.5 * dt;
    // B half step on local cells k in [0, FW-1): needs E[k+1]
    if (work)
        for (int k = t; k < FW - 1; k += T) {
            int i = s - FH + k;
            if (i >= 0 && i < len - 1) {
                By[k] += hdt * (Ez[k + 1] - Ez[k]) / dx;
                Bz[k] += hdt * (Ey[k] - Ey[k + 1]) / dx;
            }
        }
    __syncthreads();
    // E full step on local cells k in [1, FW-1): needs B[k-1] (post half step)
    if (work)
        for (int k = 1 + t, pass = 0; k < FW - 1; k += T, pass++) {
            int i = s - FH + k;
            if (i == 0) {
                double kappa = 2.0 * SPEED_OF_LIGHT * dt / (SPEED_OF_LIGHT * dt + dx);
                double sigma = 1.0 - kappa;
                Ex[k] = 0.0;
                Ey[k] = sigma * Ey[k] - SPEED_OF_LIGHT * kappa * Bz[k];
                Ez[k] = sigma * Ez[k] + SPEED_OF_LIGHT * kappa * By[k];
            } else if (i > 0 && i < len) {
                double jx = pass ? jx1 : jx0, jy = pass ? jy1 : jy0, jz = pass ? jz1 : jz0;
                Ex[k] += -dt * jx / VACUUM_PERMITTIVITY;
                Ey[k] += dt * SPEED_OF_LIGHT_SQD * (Bz[k - 1] - Bz[k]) / dx - dt * jy / VACUUM_PERMITTIVITY;
                Ez[k] += dt * SPEED_OF_LIGHT_SQD * (By[k] - By[k - 1]) / dx - dt * jz / VACUUM_PERMITTIVITY;
            }
        }
    __syncthreads();
    // second B half step on the tile proper, then write out
    {
        int k = FH + t, i = s + t;
        if (work && i < len && i < e) {
            double by = By[k], bz = Bz[k];
            if (i < len - 1) {
                by += hdt * (Ez[k + 1] - Ez[k]) / dx;
                bz += hdt * (Ey[k] - Ey[k + 1]) / dx;
            }
            a.dst[0 * lenp + i] = Ex[k];
            a.dst[1 * lenp + i] = Ey[k];
            a.dst[2 * lenp + i] = Ez[k];
            a.dst[3 * lenp + i] = Bx ? Bx[k] : a.src[3 * lenp + i];  // "nothing happens to Bx in 1D" (yee.rs:844)
            a.dst[4 * lenp + i] = by;
            a.dst[5 * lenp + i] = bz;
            // YeeGrid::clear (yee.rs:543-549) for the next step, done on the spare current buffer: nobody
            // reads it during this kernel, and the currents just used stay available to write_data
            a.jr_next[0 * lenp + i] = 0.0;
            a.jr_next[1 * lenp + i] = 0.0;
            a.jr_next[2 * lenp + i] = 0.0;
            a.jr_next[3 * lenp + i] = 0.0;
        }
    }
}

template <int T>
__device__ inline void advance_tile(const AdvArgs &a, int s, int e, int t, double *lds)
{
    constexpr int FH = ADV_HALO, FW = T + 2 * FH;
    double *Ex = lds, *Ey = lds + FW, *Ez = lds + 2 * FW, *By = lds + 3 * FW, *Bz = lds + 4 * FW;
    const size_t lenp = a.lenp;
    const int len = a.len;
    if (e > s)
        for (int k = t; k < FW; k += T) {
            int i = s - FH + k;
            bool in = i >= 0 && i < len;
            Ex[k] = in ? a.src[0 * lenp + i] : 0.0;
            Ey[k] = in ? a.src[1 * lenp + i] : 0.0;
            Ez[k] = in ? a.src[2 * lenp + i] : 0.0;
            By[k] = in ? a.src[4 * lenp + i] : 0.0;
            Bz[k] = in ? a.src[5 * lenp + i] : 0.0;
        }
    advance_tile_steps<T>(a, s, e, t, Ex, Ey, Ez, By, Bz, nullptr);  // (begins with a barrier)
}

// an interior workgroup (ADV_TILE threads): tile number `tile` between the edges (the edges: sync_advance_edges below)
__device__ inline void advance_interior(const AdvArgs &a, int tile, double *lds /* 5 x (ADV_TILE + 4) doubles */)
{
    advance_tile<ADV_TILE>(a, a.edge_left + tile * ADV_TILE, a.len - a.edge_right, threadIdx.x, lds);
}

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

// synchronize followed by the field solve of the cells it touches, by ONE workgroup of 1024 threads that does both
// (k_sync_advance's first workgroup, k_link_tail's): the same results as synchronize_body + advance_edges, with what
// synchronize does to E and B done in LDS.  Those values exist for one purpose -- the field solve that follows reads them
// and writes the OTHER field buffer -- so they need never reach memory; what synchronize adds to the currents and the
// charge density does (write_data reads them).  The workgroup's chain becomes: the neighbours' halo and both edge tiles
// (one round trip to memory), the currents' halo cells (written, then read by the E step: a second), the solve -- where
// synchronize_body + advance_edges went four times (halo, overlay, absorber read-modify-write, tiles).
__device__ inline void sync_advance_edges(const SyncArgs &sa, const AdvArgs &a, double *lds /* 2 x 6 x (ADV_EDGE + 4) doubles */)
{
    constexpr int T = ADV_EDGE, FH = ADV_HALO, FW = T + 2 * FH;
    const int tid = threadIdx.x, half = tid >> 9, t = tid & (T - 1);
    const int s_of[2] = {0, a.len - a.edge_right};
    const int s = s_of[half], e = half ? a.len : a.edge_left;
    const size_t lenp = a.lenp;
    const int len = a.len;
    // (i) what the neighbours sent -- or, alone on a periodic domain, this grid's own other end: yee.rs:424-452
    int side = tid / HALO_DOUBLES, r = tid % HALO_DOUBLES;
    int k = r / HALO_COMPS, comp = r % HALO_COMPS;
    bool active = false;
    double v = 0.0;
    if (tid < 2 * HALO_DOUBLES) {
        if (sa.self_periodic) {
            int src = side == 0 ? len - HALO_CELLS + k : k;
            v = comp == 0 ? sa.jr[3 * lenp + src]
                : comp < 4 ? sa.jr[(size_t)(comp - 1) * lenp + src]
                           : sa.eb[(size_t)(comp - 4) * lenp + src];
            active = true;
        } else {
            const double *recv = side == 0 ? sa.recv_left : sa.recv_right;
            if (recv) {
                v = recv[r];
                active = true;
            }
        }
    }
    // this half's tile of E and B (all six: synchronize's overlay, absorber and mirror act on Bx too) as memory has them
    double *tile = lds + half * 6 * FW;
    if (e > s)
        for (int q = t; q < FW; q += T) {
            int i = s - FH + q;
            bool in = i >= 0 && i < len;
#pragma unroll
            for (int c = 0; c < 6; c++) tile[c * FW + q] = in ? a.src[(size_t)c * lenp + i] : 0.0;
        }
    __syncthreads();  // every value a lane adds below was read before anybody writes; the tiles are there
    // Where cell i stands in the tiles: in one of them, or -- within two cells of where they meet on a grid too short to
    // have an interior -- in both (each tile carries a two-cell halo): both copies hold the same value and get the same
    // treatment.
    const int e_of[2] = {a.edge_left, len};
    auto in_tiles = [&](int i, auto &&f) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = i - s_of[h] + FH;
            if (e_of[h] > s_of[h] && q >= 0 && q < FW) f(lds + h * 6 * FW + q);
        }
    };
    if (active) {
        int cellidx = side == 0 ? k : len - HALO_CELLS + k;
        bool ghost = side == 0 ? (k < GHOST_SIZE) : (k >= GHOST_SIZE);
        if (comp == 0) sa.jr[3 * lenp + cellidx] += v;                               // Cell::overlay, yee.rs:97-113
        else if (comp < 4) sa.jr[(size_t)(comp - 1) * lenp + cellidx] += v;
        else if (ghost) in_tiles(cellidx, [&](double *cell) { cell[(comp - 4) * FW] = v; });  // Cell::overlay_ghost
    }
    // (ii) laser, yee.rs:456-462 (a laser boundary has no neighbour on that side: the cell is nobody's ghost)
    if (sa.left_bdy == OPAL_BDY_LASER && tid == 0)
        in_tiles(sa.L - 2, [&](double *cell) {
            cell[1 * FW] += sa.laser_y;
            cell[2 * FW] += sa.laser_z;
        });
    // (iii) absorbing layer, yee.rs:464-479 / conducting wall, :480-495 (likewise: no neighbour on that side)
    if (sa.right_bdy == OPAL_BDY_ABSORBING) {
        int first = len - sa.R + 1;
        for (int i = first + tid; i < len; i += blockDim.x) {
            double f = sa.damp[i - first];
            bool zero = i >= len - 2;
            in_tiles(i, [&](double *cell) {
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double w = f * cell[c * FW];
                    cell[c * FW] = zero ? 0.0 : w;
                }
            });
        }
    } else if (sa.right_bdy == OPAL_BDY_CONDUCTING) {
        // (the wall's R = 4 cells and the R - 1 below it that they mirror lie in the right tile proper, which is at least
        // 8 cells long, and outside the left tile's halo)
        double *right = lds + 6 * FW;
        int q0 = len - sa.R - s_of[1] + FH;  // the wall's cell in the tile
        // every source is at or below the wall's cell, every destination above it: the wall's own cell first
        if (tid == 0) {
            right[0 * FW + q0] = 0.0;
            right[4 * FW + q0] = 0.0;
            right[5 * FW + q0] = 0.0;
        }
        __syncthreads();
        if (tid >= 1 && tid < sa.R) {
            int i = tid;
            right[0 * FW + q0 + i] = -right[0 * FW + q0 - i];
            right[1 * FW + q0 + i] = right[1 * FW + q0 + 1 - i];
            right[2 * FW + q0 + i] = right[2 * FW + q0 + 1 - i];
            right[3 * FW + q0 + i] = right[3 * FW + q0 + 1 - i];
            right[4 * FW + q0 + i] = -right[4 * FW + q0 - i];
            right[5 * FW + q0 + i] = -right[5 * FW + q0 - i];
        }
    }
    __syncthreads();  // the currents' halo cells are written: the E step's loads below see them
    advance_tile_steps<T>(a, s, e, t, tile, tile + FW, tile + 2 * FW, tile + 4 * FW, tile + 5 * FW, tile + 3 * FW);
}
#endif
}  // namespace opal
