// Population<Electron|Ion|Photon> on the GPU.
//
// HBM layout: structure of arrays, one contiguous array per attribute
//   cell:i32  x  ux  uy  uz  w : f64          (44 B per macro-particle)
// so that a wavefront's 64 lanes read/write 256- or 512-byte contiguous runs.
// gamma and prev_x are recomputed in registers, never stored, on the fused
// path; prev_x is materialised only for the reference's separate
// advance -> deposit operator sequence.  A second copy of the arrays exists
// only while re-binning (counting sort by cell).
//
// The hot kernel (k_push<...,FUSED>) is one pass per particle per step:
//   read 44 B, gather E,B from an LDS copy of the field tile, Vay/Boris push,
//   charge-conserving deposit with ds_add_f64 into LDS accumulators (a 64-cell
//   tile x 4 components x 8 replicas selected by lane, so at most 8 lanes of a
//   wave meet on an address; measured: 8, 16 and 32 replicas run equally fast,
//   the LDS pipe has slack, while a wide tile keeps fast particles on the fast
//   path for more steps between re-binnings), write 36 B; one global f64 atomic
//   per touched (cell, component) per workgroup at the end.  Algorithmic traffic 80 B per particle-update; the
//   bound is HBM (8 TB/s -> 1.0e11 updates/s), with f64 VALU issue close
//   behind (about 4-6 flop/B).
#include "link.hpp"
#include "opal_internal.hpp"
#include "pic_device.hpp"
#include "grid_device.hpp"
#include "peer_device.hpp"
#include "qed_device.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <type_traits>

using namespace opal;

namespace opal {
int grid_check_err(const opal_grid_t *g);
int pop_tighten_upper(opal_pop_t *p);
int pop_snapshot_upper(opal_pop_t *p, size_t just_added);
}

namespace {

constexpr int BLOCK = 256;
#ifndef OPAL_MAX_PPT
#define OPAL_MAX_PPT 8  // particles per thread of a push workgroup, at most
#endif
#ifndef OPAL_ES_PPT
#define OPAL_ES_PPT 16  // the same for a population the electrostatic launch serves (see pop_max_ppt)
#endif
#ifndef OPAL_TWO_SIDED_TAIL
#define OPAL_TWO_SIDED_TAIL 1  // chunks of arrivals are taken by two workgroups, one per half of the subdomain (push_body)
#endif
#ifndef OPAL_ES_WATCH
#define OPAL_ES_WATCH 1  // the safety net of process_particle<ES_WATCH>; 0 only for experiments
#endif
#ifndef OPAL_TILE
#define OPAL_TILE 64
#endif
#ifndef OPAL_TILE_MARGIN
#define OPAL_TILE_MARGIN 29
#endif
constexpr int TILE = OPAL_TILE;        // cells in a workgroup's LDS tile
constexpr int TILE_MARGIN = OPAL_TILE_MARGIN;  // tile cells below the chunk's lowest cell (room for leftward drift)
#ifndef OPAL_REPL
#define OPAL_REPL 8
#endif
constexpr int REPL = OPAL_REPL;        // lane-private replicas per accumulator
constexpr int TILE_LO = 3;      // a particle whose old cell sits at tile offset in [TILE_LO, TILE - 4]
constexpr int TILE_HI = TILE - 4;  // gathers (rel-1..rel+2) and deposits (rel-3..rel+3) inside the tile

enum { MODE_ADVANCE = 0, MODE_FUSED = 1, MODE_DEPOSIT = 2 };
bool g_elision = true;  // opal_set_elision

// Slots per thread of a push workgroup, at most.  The electrostatic launch moves 38 bytes per particle instead of 62-80,
// so a workgroup's fixed cost (tile set-up, the replica fold and its atomics) weighs twice as much: with 1e8 particles
// chunks twice as long are worth +6 % (the general kernel loses 5 % with them).  Only where the launch still has
// thousands of workgroups per species, though: at 1e7 particles per species the longer chunks cost 5-8 % (the chip
// holds ~2000 workgroups at a time, and a short launch is all ragged edge).
int pop_max_ppt(const opal_pop_t *p)
{
    static const int forced = getenv("OPAL_PPT") ? atoi(getenv("OPAL_PPT")) : 0;  // experiments
    if (forced > 0) return std::min(forced, 32);
    if (!(g_elision && p->cold_yz && p->species != OPAL_PHOTON)) return OPAL_MAX_PPT;
    const size_t by_size = p->n_upper / ((size_t)BLOCK * 8192);  // keep >= 8192 workgroups per species
    // (Round 4 tried chunk lengths that make the launch a whole number of rounds of the chip -- 6144 slots at one GPU's
    // share of configs[2] at N = 8: exactly one round instead of 2.98 -- and lost 15 % there, 4 % at configs[1], 2-3 % at
    // configs[2]: profiles/r04_ab/ab_ppt_rounds.txt.  Workgroups that all start together stay in step; it is the
    // staggered rounds that keep the memory system busy while some of them set up tiles or fold replicas.)
    return (int)std::min<size_t>(OPAL_ES_PPT, std::max<size_t>(OPAL_MAX_PPT, by_size));
}
enum { EDGE_WRAP = 0, EDGE_DELETE = 1, EDGE_SEND = 2 };

constexpr int ROTATE_BELOW = 96;  // steps between two re-binnings (two-stream: hundreds; the laser-plasma electrons: ~20, its ions: ~55)
struct PushArgs {
    int *cell;
    double *x, *ux, *uy, *uz;
    const double *w;
    double *prev_x;
    double *ex[NEX];  // extras (any may be null, see opal_pop)
    const double *eb;  // 6 x lenp
    double *jr;        // 4 x lenp
    size_t lenp;
    int ppt;
    int size, L, len;
    // The frame the results are stored in.  Normally the subdomain's own (out_shift = 0, out_size = size).  In the step in
    // which the cuts move (sim.cpp: sim_recut) the grid is still the old one -- gather, deposit, tiles: `size` -- while the
    // cell a particle is stored with, and the test of who has left, already belong to the NEW subdomain: its first cell is
    // the old cell out_shift, it has out_size cells.  A beam that the cuts follow changes nobody's subdomain that way.
    int out_shift, out_size;
    double dt, dt_over_dx, inv_dt, inv_dx, alpha, charge;
    int left_mode, right_mode;
    int zone;  // cells from a linked edge within which particles are counted into the message header (link_plan); 0: nobody asks
    int zone_word;  // ... into this word of the header: this exchange's set (link_h_zone) + the species slot
    PopCounters *ctr;
    // leavers (EDGE_SEND) go into the link's send messages (layout: link.hpp): header words + row blocks
    char *send_msg[2];
    unsigned send_cap[2];  // slots this step's message ships per species on that edge (link_plan)
    LinkLayout send_layout;
    int send_slot;         // this species' slot in the message
    unsigned send_nextra;  // extras carried behind the 6 base fields (0 for a plain link)
    unsigned *emit_list;   // EXTRAS == 2: slots whose optical depth ran out in this push (count in ctr->n_emit)
    // chunk table of the last re-binning (see opal_pop): workgroups [0, n_tab) take tab_slot[b] .. tab_slot[b+1],
    // the rest uniform chunks of BLOCK x ppt slots from ctr->n_sorted on.  n_tab = 0: uniform chunks throughout.
    const unsigned *tab_slot;
    unsigned n_tab;
    int rotate;  // the waves of a workgroup start a quarter of their chunk apart (push_body: slot_of)
    int tail_sides;  // 2: the chunks behind the sorted slots are each taken by two workgroups (see push_body)
    int w_uniform;  // every slot's weight is w0 unless ctr->w_mixed says otherwise (opal_pop::w_uniform)
    double w0;
    int *es_flag;  // ES kernels: raised when the launch's belief fails somewhere (the grid's mapped host word)
    int *dbg;  // diagnostic (opal_pop_debug_misses): [0] count, then (block, slot, cell, tile base cell) per tile miss
};

struct LdsFields {
    // [TILE][6]: the 18 values a particle gathers lie within 200 bytes of one address, so every ds_read of the gather
    // takes the same address register with an immediate offset (component-major rows cost one address add per read)
    const double *f;
    __device__ double operator()(int comp, int j) const { return f[j * 6 + comp]; }
};
struct GlobalFields {
    const double *eb;
    size_t lenp;
    __device__ double operator()(int comp, int j) const { return eb[(size_t)comp * lenp + j]; }
};

__device__ inline void lds_add(double *p, double v)
{
    if (v != 0.0) atomicAdd(p, v);  // ds_add_f64
}
__device__ inline void glb_add(double *p, double v)
{
    if (v != 0.0) atomicAdd(p, v);  // global_atomic_add_f64
}

template <bool RHO>
__device__ inline void scatter_lds(double *acc, int rel, int lane, const Taps &t)
{
    // acc[comp][cell][replica].  Taps that are nearly always non-zero are added unconditionally; the
    // outer jx taps (non-zero only for a particle that crossed a cell) and the transverse currents
    // (zero for a whole species in 1-D electrostatic problems) are skipped when no lane needs them.
    double *a = acc + (rel - 2) * REPL + lane;  // every tap below is this base plus a non-negative constant
    lds_add(a, t.jx[0]);
#pragma unroll
    for (int k = 1; k < 4; k++) atomicAdd(a + k * REPL, t.jx[k]);
    lds_add(a + 4 * REPL, t.jx[4]);
    if (t.qy != 0.0) {
#pragma unroll
        for (int k = 0; k < 3; k++) atomicAdd(a + (1 * TILE + k + 1) * REPL, t.jy(k));
    }
    if (t.qz != 0.0) {
#pragma unroll
        for (int k = 0; k < 3; k++) atomicAdd(a + (2 * TILE + k + 1) * REPL, t.jz(k));
    }
    if (RHO) {
#pragma unroll
        for (int k = 0; k < 4; k++) atomicAdd(a + (3 * TILE + k) * REPL, t.rho[k]);
    }
}

template <bool RHO>
__device__ inline void scatter_global(double *jr, size_t lenp, int idx, const Taps &t)
{
#pragma unroll
    for (int k = 0; k < 5; k++) glb_add(jr + 0 * lenp + idx + k - 2, t.jx[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) glb_add(jr + 1 * lenp + idx + k - 1, t.jy(k));
#pragma unroll
    for (int k = 0; k < 3; k++) glb_add(jr + 2 * lenp + idx + k - 1, t.jz(k));
    if (RHO) {
#pragma unroll
        for (int k = 0; k < 4; k++) glb_add(jr + 3 * lenp + idx + k - 2, t.rho[k]);
    }
}

// Element k of a block-uniform array with a 32-bit byte offset, so that the access takes the
// scalar-base + 32-bit vector-offset form (a chunk is at most 32 x 256 slots).
template <class T>
__device__ __forceinline__ T ldk(const T *base, unsigned k)
{
    // streamed once per step: the non-temporal hint keeps the particle arrays from displacing the grid rows
    // in L2 (measured: loads and stores together take 4 % off the fused kernel)
    return __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + k * (unsigned)sizeof(T)));
}
template <class T>
__device__ __forceinline__ void stk(T *base, unsigned k, T v)
{
    __builtin_nontemporal_store(v, reinterpret_cast<T *>(reinterpret_cast<char *>(base) + k * (unsigned)sizeof(T)));
}

// Per-chunk view of the particle arrays: block-uniform base pointers (scalar registers) indexed
// with a 32-bit slot offset.
struct ChunkPtrs {
    int *cell;
    double *x, *ux, *uy, *uz;
    const double *w;
    double *prev;
    size_t chunk0;
};

// The per-particle body shared by the three modes.  FAST = the particle's stencil lies inside the
// workgroup's LDS tile (fields from LDS, currents into the LDS accumulators); otherwise fields come
// from global memory and the taps go to the grid with global atomics.
// ES_WATCH: the general path inside an electrostatic launch (a particle outside its workgroup's tile, where the
// tile's check says nothing about the fields): transverse momentum that appears here ends the population's promise.
template <int SPECIES, int MODE, bool RHO, int EXTRAS, bool SENDX, bool FAST, bool ES = false, bool ES_WATCH = false>
__device__ __forceinline__ void process_particle(const PushArgs &a, const ChunkPtrs &cp, const double *s_field,
                                                 double *s_acc, int base, int lane, unsigned kc, int c, int c_mem,
                                                 double xi, V3 u, double wt, double prev, int &err, bool &emits)
{
    // c: the cell to work in (the periodic image when the caller picked one); c_mem: what the slot holds
    // ES: the slot's uy and uz are zero (not even loaded, see push_body) and, with no transverse field, stay zero
    if (ES) u.y = u.z = 0.0;
    const double uy_mem = u.y, uz_mem = u.z;
    constexpr bool GATHER = MODE != MODE_DEPOSIT;
    constexpr bool DEPOSIT = MODE != MODE_ADVANCE;
    int newc = c, rel, idx;
    double xn = xi, inv_g;
    double qed_tau = 0.0, qed_chi = 0.0, qed_gh = 1.0;
    if (GATHER) {
        rel = c + a.L - base;
        V3 E, B;
        if (FAST) {
            LdsFields f{s_field};
            gather<ES>(f, rel, xi, E, B);
        } else {
            GlobalFields f{a.eb, a.lenp};
            gather(f, c + a.L, xi, E, B);
        }
        Pushed r;
        if (SPECIES == OPAL_ELECTRON) {
            r = vay_push<EXTRAS != 0, ES>(u, E, B, a.alpha, a.dt_over_dx, a.dt);
            if (EXTRAS) {
                size_t i = cp.chunk0 + kc;
                // each array is optional (block-uniform tests): a run only carries what its outputs ask for
                if (a.ex[EX_CHI]) a.ex[EX_CHI][i] = r.chi;
                if (a.ex[EX_WORK]) a.ex[EX_WORK][i] += r.dwork;
                if (a.ex[EX_Y]) {
                    a.ex[EX_Y][i] += r.v_old.y * a.dt;  // electron.rs:315-316: advanced with the pre-push velocity
                    a.ex[EX_Z][i] += r.v_old.z * a.dt;
                }
                if (EXTRAS == 2) {  // electron.rs:284-288: the optical depth is run down at the END of this function
                    qed_tau = a.ex[EX_TAU][i];  // (loaded here: the latency hides behind the rest of the push)
                    qed_chi = r.chi;
                    qed_gh = r.gamma_half;
                }
            }
        } else if (SPECIES == OPAL_PHOTON) {
            r = photon_push<EXTRAS != 0>(u, E, B, a.dt_over_dx);
            if (EXTRAS) {
                size_t i = cp.chunk0 + kc;
                if (a.ex[EX_CHI]) a.ex[EX_CHI][i] = r.chi;
                if (a.ex[EX_Y]) {
                    a.ex[EX_Y][i] += r.v_old.y * a.dt;  // photon.rs:166-167
                    a.ex[EX_Z][i] += r.v_old.z * a.dt;
                }
            }
        } else {
            r = boris_push<ES>(u, E, B, a.alpha, a.dt_over_dx);
        }
        if (!(r.dxi < 1.0)) err |= 1;  // electron.rs:313, ion.rs:199
        u = r.u;
        if (ES_WATCH && (u.y != 0.0 || u.z != 0.0)) {
            a.ctr->hot_yz = 1;
            *a.es_flag = 1;
        }
        inv_g = r.inv_gamma;
        // electron.rs:311-329 / ion.rs:197-213
        xn = xi + r.dxi;
        double fl = floor(xn);
        // the reference steps the cell by the SIGN of floor(x) (one cell at most, electron.rs:319-327)
        int s;
        asm("v_med3_i32 %0, %1, -1, 1" : "=v"(s) : "v"((int)fl));  // min(max(floor, -1), 1) in one instruction
        prev = xi - fl;
        xn = xn - fl;
        newc = c_mem + s;  // from the stored cell: one wrap below brings it back into [0, size) whatever image c is
        idx = c + a.L + s;
        rel = rel + s;
    } else {
        double gam = sqrt(1.0 + dot(u, u));  // electron.rs:148-151 uses the stored gamma; ion.rs:123-126: 1 + gamma_m1
        if (SPECIES == OPAL_ION) gam = 1.0 + dot(u, u) / (1.0 + gam);
        inv_g = 1.0 / gam;
        idx = c + a.L;
        rel = idx - base;
    }

    // leavers: mod.rs:273-278, 291-293, 308-354; yee.rs:329-343.  One unsigned compare finds the rare particle
    // that stepped outside [0, size); only those lanes look at the edge modes.
    bool deleted = false;
    int send_side = -1;
    // (the ballot makes the test a scalar branch: away from the subdomain's edges no lane of the wave has left, and
    // the selects below are not even issued)
    if (GATHER) newc -= a.out_shift;  // (the frame the result is stored in: PushArgs)
    if (GATHER && __ballot((unsigned)newc >= (unsigned)a.out_size) && (unsigned)newc >= (unsigned)a.out_size) {
        const bool left = newc < 0;
        const int mode = left ? a.left_mode : a.right_mode;
        if (mode == EDGE_WRAP) {
            newc += left ? a.out_size : -a.out_size;
        } else if (mode == EDGE_DELETE) {
            deleted = true;
        } else {
            send_side = left ? 0 : 1;
            if (!left) newc -= a.out_size;  // mod.rs:291-293
        }
    }

    if (DEPOSIT && !deleted) {
        double macrocharge = wt * a.charge;
        const double cig = SPEED_OF_LIGHT * inv_g;
        double vy = cig * u.y, vz = cig * u.z;
        Taps t = deposit_taps<RHO>(xn, prev, macrocharge, vy, vz, a.inv_dt, a.inv_dx);
        if (FAST) scatter_lds<RHO>(s_acc, rel, lane, t);
        else scatter_global<RHO>(a.jr, a.lenp, idx, t);
    }

    if (GATHER) {
        if (!deleted && send_side < 0) {
            // Write-back skips what did not change: an attribute is stored only if some lane of the wave holds
            // a value whose bits differ from what it loaded.  Results are identical by construction; the saving
            // is real where a component is constant for whole waves -- the transverse momenta of a 1-D
            // electrostatic run (E_y = E_z = B = 0: 16 of the 36 bytes a particle writes), the cell index of
            // slow particles -- and the test costs one compare and one scalar branch per attribute elsewhere
            // (and two registers per tested momentum component, which is why ux is not tested).
            auto differs = [](double p, double q) { return __double_as_longlong(p) != __double_as_longlong(q); };
            // towards a neighbouring subdomain: count who stands within LINK_ZONE cells of that edge now -- nobody
            // else can cross it in the next LINK_ZONE steps, which is what sizes the coming messages (link.cpp)
            // (a.zone = 0 where there is no linked edge, or its transport does not plan message sizes: no lane passes)
            if (__ballot(newc < a.zone || newc >= a.out_size - a.zone)) {
                if (a.left_mode == EDGE_SEND && newc < a.zone)
                    atomicAdd(reinterpret_cast<unsigned *>(a.send_msg[0]) + a.zone_word, 1u);
                if (a.right_mode == EDGE_SEND && newc >= a.out_size - a.zone)
                    atomicAdd(reinterpret_cast<unsigned *>(a.send_msg[1]) + a.zone_word, 1u);
            }
            if (__ballot(newc != c_mem)) stk(cp.cell, kc, newc);
            stk(cp.x, kc, xn);
            if (SPECIES != OPAL_PHOTON) {  // a photon's momentum never changes
                stk(cp.ux, kc, u.x);  // (not tested: it changes wherever there is a field at all)
                if (__ballot(differs(u.y, uy_mem))) stk(cp.uy, kc, u.y);
                if (__ballot(differs(u.z, uz_mem))) stk(cp.uz, kc, u.z);
            }
            if (MODE == MODE_ADVANCE && cp.prev) stk(cp.prev, kc, prev);  // only kept for a later separate deposit
        } else if (deleted) {
            cp.cell[kc] = DEAD_CELL;
            atomicAdd(&a.ctr->n_deleted, 1u);
            atomicAdd(&a.ctr->n_dead, 1u);
        } else {
            atomicAdd(&a.ctr->n_dead, 1u);
            char *msg = a.send_msg[send_side];
            // a leaver stands within a cell of the cut on the far side after this exchange and may come back: it belongs
            // to the zone count of the edge it crosses (neither end's push would count it otherwise: link.cpp)
            if (a.zone) atomicAdd(reinterpret_cast<unsigned *>(msg) + a.zone_word, 1u);
            // One reservation per wave and side, not one per leaver: next to a cut that sits in a pile-up -- where cuts that
            // follow the particles do sit -- thousands leave every step, and in the step in which a cut moves hundreds of
            // thousands do, all bumping ONE counter (round 5: the 4-rank laser-plasma rehearsal spent a third of its step there).
            // (through a pointer that is the same in every lane -- one side at a time --, so that the compiler's atomic optimizer
            // turns the increments of a wave into one atomic and a prefix count, as it does for the zone counts above)
            unsigned slot = 0;
#pragma unroll
            for (int sd = 0; sd < 2; sd++)
                if (send_side == sd) slot = atomicAdd(reinterpret_cast<unsigned *>(a.send_msg[sd]) + LINK_H_COUNT + a.send_slot, 1u);
            if (slot < a.send_cap[send_side]) {
                *link_cell(msg, a.send_layout, a.send_slot, slot) = newc;
                *link_field(msg, a.send_layout, a.send_slot, 0, slot) = xn;
                *link_field(msg, a.send_layout, a.send_slot, 1, slot) = u.x;
                *link_field(msg, a.send_layout, a.send_slot, 2, slot) = u.y;
                *link_field(msg, a.send_layout, a.send_slot, 3, slot) = u.z;
                *link_field(msg, a.send_layout, a.send_slot, 4, slot) = wt;
                if (MODE == MODE_ADVANCE && cp.prev) *link_field(msg, a.send_layout, a.send_slot, 5, slot) = prev;
                // A link laid out for photon emission carries the extras too (the reference ships the whole
                // struct, mod.rs:308-354).  Only the leaver's slot is noted here; k_pack_extras copies the
                // values afterwards, which keeps the extras' pointers out of this kernel's registers.
                if (SENDX) *link_field(msg, a.send_layout, a.send_slot, 6, slot) = (double)(cp.chunk0 + kc);
            } else {
                err |= 8;
            }
            cp.cell[kc] = DEAD_CELL;
        }
    }
    if (EXTRAS == 2 && SPECIES == OPAL_ELECTRON && GATHER) {
        // electron.rs:284-288, tau -= rate(chi, gamma_half) dt -- here, behind the write-back, where nothing of the push is
        // live any more: the rate's logarithm and exponential (photon_emission.rs:59-79) are the kernel's widest stretch of
        // temporaries, and beside the momentum, the weights and the loads in flight they cost it two waves per SIMD
        // (141 registers against 96; round 5).  A particle that leaves ships the value (k_pack_extras reads the array
        // afterwards) and radiates where it arrives (or nowhere).
        const double tau = qed_tau - qed_rate(qed_chi, qed_gh) * a.dt;
        a.ex[EX_TAU][cp.chunk0 + kc] = tau;
        emits = tau < 0.0 && !(deleted || send_side >= 0);  // electron.rs:209
    }
}

// One workgroup = BLOCK x ppt consecutive particle slots.
//   MODE_ADVANCE : Population::advance's per-particle body, mod.rs:268-279 (gather, push, leaver
//                  detection); stores prev_x for a later deposit.
//   MODE_DEPOSIT : YeeGrid::deposit's per-particle body, yee.rs:573-610.
//   MODE_FUSED   : both in one pass (prev_x stays in registers).
//
// ES ("electrostatic"): the kernel was launched in the belief that nothing transverse exists -- Ey = Ez = B = 0
// everywhere and uy = uz = 0 for every particle of the population: the reference's two-stream benchmark, any 1-D
// electrostatic run.  Its fast path then neither loads uy and uz (16 of the 44 bytes a particle reads) nor gathers
// anything but Ex, and the pushers and the deposit are the general ones with those zeros put in: every term that
// drops out is an exact zero or an exact identity (pic_device.hpp), so no particle's result depends on which kernel
// ran.  The belief is checked where it matters: a workgroup whose tile holds a transverse field, or whose population
// has been flagged as carrying transverse momentum (ctr->hot_yz: arrivals, a field met earlier), sends all its
// particles down the general (global-memory) path and raises the grid's flag, which the host reads -- without
// waiting -- before later launches and then stays with the general kernel.
template <int SPECIES, int MODE, bool RHO, int EXTRAS, bool SENDX, bool ES = false>
__device__ __forceinline__ void push_body(const PushArgs &a, unsigned block, double *s_field, double *s_acc, int *s_min,
                                          int *s_max)
{
    static_assert(!ES || (MODE == MODE_FUSED && EXTRAS == 0 && SPECIES != OPAL_PHOTON), "ES: plain fused pushes only");
    constexpr bool GATHER = MODE != MODE_DEPOSIT;
    constexpr bool DEPOSIT = MODE != MODE_ADVANCE;
    const int tid = threadIdx.x;
    const int lane = tid & (REPL - 1);
    size_t chunk0;
    unsigned nloc;
    int my_half = -1;  // block-uniform
    const bool tabled = block < a.n_tab;  // block-uniform
    if (tabled) {
        chunk0 = a.tab_slot[block];
        nloc = a.tab_slot[block + 1] - (unsigned)chunk0;
        if (nloc == 0) return;
    } else {
        const unsigned n = a.ctr->n_slots;
        const unsigned tb = block - a.n_tab;
        chunk0 = (size_t)a.ctr->n_sorted + (size_t)(a.tail_sides == 2 ? tb >> 1 : tb) * BLOCK * a.ppt;
        if (chunk0 >= n) return;
        nloc = (unsigned)min((size_t)BLOCK * a.ppt, (size_t)n - chunk0);
        if (OPAL_TWO_SIDED_TAIL && a.tail_sides == 2) my_half = (int)(tb & 1u);
    }
    // Arrivals: behind the sorted slots, in the order they came -- from the left neighbour (cells near 0) and from the
    // right one (cells near size - 1) in turn.  No tile spans both, so with two neighbours every chunk of arrivals is
    // taken by TWO workgroups, one per half of the subdomain (my_half = 0 / 1; -1: everything is mine).  Each puts
    // its tile at its own end; the other half's particles are half a subdomain away from it, fall out of the main
    // loop as "pending" at no cost to that loop, and are dropped from the pending set below.  (Nobody in these chunks
    // is anywhere near the middle: arrivals enter at the edges and move less than a cell per step, and build_args stops
    // sharing the chunks long before the oldest of them could have got there -- the misses that follow bring on a
    // re-binning, which starts the count again.)
    const int iters = (int)((nloc + BLOCK - 1) / BLOCK);  // <= pop_max_ppt <= 32 (the masks below)

    ChunkPtrs cp;
    cp.cell = a.cell + chunk0;
    cp.x = a.x + chunk0; cp.ux = a.ux + chunk0; cp.uy = a.uy + chunk0; cp.uz = a.uz + chunk0;
    cp.w = a.w + chunk0;
    cp.prev = (MODE != MODE_FUSED && a.prev_x) ? a.prev_x + chunk0 : nullptr;
    cp.chunk0 = chunk0;

    // The first slots' loads go out before anything else: the first cell also decides the tile base, and
    // the other values are in flight while the tile is set up.  One slot per lane stays in flight.
    struct Slot {
        int c;
        double x, ux, uy, uz, w, prev;
    };
    // Lanes past the end of the chunk load its last slot again (same cache lines, no extra HBM traffic) and are
    // masked by `have`: unconditional loads need neither an exec-mask branch nor zero-filled registers.
    const unsigned last = nloc - 1;  // nloc >= 1
    const bool uniform_w = a.w_uniform && a.ctr->w_mixed == 0;
    auto fetch = [&](unsigned kk) {
        Slot sl;
        sl.c = ldk(cp.cell, kk); sl.x = ldk(cp.x, kk); sl.ux = ldk(cp.ux, kk);
        sl.uy = ES ? 0.0 : ldk(cp.uy, kk);
        sl.uz = ES ? 0.0 : ldk(cp.uz, kk);
        sl.w = uniform_w ? a.w0 : ldk(cp.w, kk);  // block-uniform
        sl.prev = (MODE == MODE_DEPOSIT) ? ldk(cp.prev, kk) : 0.0;
        return sl;
    };
    // The order of the loop below: iteration p of a thread takes slot_of(p).  In a chunk of the table the four waves start
    // a quarter of the chunk apart (and wrap round), so that the FIRST slots -- which place the tile -- are 64 from each
    // quarter.  A re-binning leaves the particles of a cell in the order its workgroups got there, which follows the slots
    // they came from: those that moved up into the cell first, those that moved down last.  The first 256 slots of a chunk
    // are therefore its up-movers, and a tile centred on them loses the down-movers at its low end a few steps early
    // (laser-plasma run: cell - tile base = 2 at the first misses; looking at both ends of every chunk re-binned 135 times
    // per 2000 steps instead of 141, scripts/r04_ab_lib_variants_lp.sh).  No load is added: the sample is the first iteration.
    // (Only for a population whose re-binnings come in quick succession -- a hot one, build_args: four runs of 64 slots a
    // quarter of the chunk apart instead of 256 consecutive ones cost the two-stream contract run 1 % and gave it nothing.)
    const int rot = (tabled && a.rotate) ? __builtin_amdgcn_readfirstlane(tid >> 6) * (iters >> 2) : 0;  // wave-uniform (a scalar); < iters
    auto slot_of = [&](int p) {
        int q = p + rot;
        q -= q >= iters ? iters : 0;
        return (unsigned)q * BLOCK + (unsigned)tid;
    };
    unsigned k = slot_of(0);
    Slot s_n = fetch(min(k, last));
    bool have_n = k < nloc;
    int c_n = have_n ? s_n.c : DEAD_CELL;
    if (OPAL_TWO_SIDED_TAIL && my_half >= 0 && (c_n >= (a.size >> 1)) != (my_half == 1)) c_n = DEAD_CELL;  // block-uniform test: the tile goes to MY end
    // tile base: lowest live cell among the first BLOCK slots of the chunk, minus a margin.  Slots that
    // hold far-away cells (particles that wrapped around a periodic domain, arrivals appended from both
    // neighbours) must not drag the tile away from the bulk.
    // A single periodic subdomain at least a tile long gets a MODULAR tile: tile cell t stands for interior cell
    // (base + t) mod size, so the chunks at the two ends of the array keep the particles that wrapped around
    // (they sit in slots sorted for the other end) in their tile.  The ghost cells a plain tile would have read
    // hold copies of exactly those interior values, and synchronize would have folded the currents there.
    const bool periodic = a.left_mode == EDGE_WRAP && a.right_mode == EDGE_WRAP;
    const bool modular = periodic && a.size >= TILE;
    const bool lv_n = c_n != DEAD_CELL;
    auto block_minmax = [&](int v, int &lo, int &hi, bool again) {  // over the live first slots
        int m = lv_n ? v : 0x7fffffff, M = lv_n ? v : (int)0x80000000;
        if (again) __syncthreads();  // everyone has read the previous s_min / s_max
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m = min(m, __shfl_xor(m, o));
            M = max(M, __shfl_xor(M, o));
        }
        if ((tid & 63) == 0) {
            s_min[tid >> 6] = m;
            s_max[tid >> 6] = M;
        }
        __syncthreads();
        lo = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
        hi = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    };
    if (DEPOSIT)
        for (int q = tid; q < 4 * TILE * REPL; q += BLOCK) s_acc[q] = 0.0;
    int cmin, cmax;
    block_minmax(c_n, cmin, cmax, false);
    int c_b = c_n;  // the cell this lane's first slot counts as when the base is chosen
    if (modular && cmin != 0x7fffffff && cmin < TILE && cmax >= a.size - TILE) {  // block-uniform: both ends present
        if (c_n >= a.size - TILE) c_b = c_n - a.size;  // the periodic image just below cell 0
        block_minmax(c_b, cmin, cmax, true);
    }
    if (cmin != 0x7fffffff && cmax - cmin > TILE_HI - TILE_LO) {  // block-uniform
        // still wider than a tile: the base is taken from the more populous half of the span
        const int mid = cmin + (cmax - cmin) / 2;
        const bool high = lv_n && c_b > mid;
        int mh = high ? c_b : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mh = min(mh, __shfl_xor(mh, o));
        int nh = __popcll(__ballot(high)), nl = __popcll(__ballot(lv_n && !high));
        __syncthreads();  // everyone has read s_min / s_max
        if ((tid & 63) == 0) {
            s_min[tid >> 6] = mh;
            s_max[tid >> 6] = nh - nl;
        }
        __syncthreads();
        int votes = s_max[0] + s_max[1] + s_max[2] + s_max[3];
        if (votes > 0) cmin = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
    }
    // Room for drift: a chunk of the table spans up to CHUNK_CELLS cells, so its tile is centred on the cells its
    // first slots hold NOW (a bunch that moves together takes its tile along); uniform chunks keep the fixed margin
    // below their lowest cell.
    const int span = (cmin == 0x7fffffff || cmax - cmin > TILE_HI - TILE_LO) ? 0 : cmax - cmin;
    const int margin = tabled ? (TILE - span) / 2 : TILE_MARGIN;
    // (nothing live among the first slots: the tile goes to cell 0 -- or, for the workgroup that takes the upper half of a
    // chunk of arrivals, to the last cell: its tile must never reach the other half's particles, which it does not test)
    const int c_none = my_half == 1 ? a.size - 1 : 0;
    int base = (cmin == 0x7fffffff ? c_none : cmin) + a.L - margin;  // array index of tile cell 0
    if (my_half == 0) base = min(base, (a.size >> 1) - TILE + a.L);  // (subdomains this applies to are >= 8 tiles long)
    if (my_half == 1) base = max(base, (a.size >> 1) + a.L);
    // array index behind tile cell j - base: the interior cell itself when the tile is modular
    auto tile_index = [&](int j) {
        if (modular) {
            int cj = j - a.L;
            cj += cj < 0 ? a.size : 0;
            cj -= cj >= a.size ? a.size : 0;
            j = cj + a.L;
        }
        return j;
    };
    bool tile_es = true;  // ES: no transverse field in the tile and no transverse momentum in the population, as assumed
    if (GATHER) {
        int f_seen = 0;
        for (int q = tid; q < 6 * TILE; q += BLOCK) {
            int comp = q / TILE, j = tile_index(base + (q % TILE));
            double v = (j >= 0 && j < a.len) ? a.eb[(size_t)comp * a.lenp + j] : 0.0;
            s_field[(q % TILE) * 6 + comp] = v;
            if (ES) f_seen |= (comp != 0 && v != 0.0);
        }
        if (ES) {
            tile_es = !__syncthreads_or(f_seen) && a.ctr->hot_yz == 0;
            if (!tile_es && tid == 0) {
                a.ctr->hot_yz = 1;  // the general path below may well leave transverse momentum behind
                *a.es_flag = 1;     // mapped host memory: the host's next launches go general
            }
        } else {
            __syncthreads();
        }
    }
    int err = 0;
    // A single periodic subdomain: a particle that wrapped around sits in a chunk at the other end of
    // the array.  Its periodic image lies inside this chunk's tile (a modular tile), or, on a domain shorter than
    // a tile, in a ghost cell: E and B there are copies of the interior, and synchronize folds the currents back.

    // Main loop, software-pipelined: the next slot's 44 bytes are requested before the current particle
    // is processed, so HBM latency overlaps this lane's own arithmetic instead of relying on occupancy
    // alone.  Only particles whose stencil lies inside the LDS tile are processed here; the others are
    // remembered in a bit mask and taken in the second loop, which keeps every global access of this
    // loop a streaming one (precise vmcnt waits, no drain of the prefetch).
    unsigned pending = 0, emit_mask = 0;
    for (int p = 0; p < iters; p++) {
        const unsigned kc = k;
        const bool live = have_n && s_n.c != DEAD_CELL;
        int c = s_n.c;
        const int c_mem = c;
        const double xi = s_n.x, wt = s_n.w, prev = s_n.prev;
        const V3 u = {s_n.ux, s_n.uy, s_n.uz};
        if (p + 1 < iters) {  // block-uniform; one slot per lane in flight (two measured 1.5 % slower)
            k = slot_of(p + 1);
            s_n = fetch(min(k, last));
            have_n = k < nloc;
        }
        if (!live) continue;
        int rel = c + a.L - base;
        bool ok, in_tile;
        constexpr int LO = GATHER ? TILE_LO : 2, HI = GATHER ? TILE_HI : TILE - 3, REACH = GATHER ? 3 : 2;
        if (GATHER) {
            ok = (unsigned)c < (unsigned)a.size;  // yee.rs:501-502
            if (!ok) {
                err |= 2;
                a.ctr->bad_cell = (unsigned)c;
            }
        } else {
            int idx = c + a.L;
            ok = idx >= 2 && idx < a.len - 2;  // yee.rs:578-579
            if (!ok) err |= 4;
        }
        if (!ok) continue;
        in_tile = rel >= LO && rel <= HI;
        if (ES && !tile_es) {
            pending |= 1u << p;  // block-uniform: this tile is not what the launch assumed
            continue;
        }
        if (!in_tile && periodic) {
            int c2 = rel > HI ? c - a.size : c + a.size;
            int rel2 = c2 + a.L - base, j2 = c2 + a.L;
            if (rel2 >= LO && rel2 <= HI && (modular || (j2 >= REACH && j2 + REACH < a.len))) {
                c = c2;
                in_tile = true;
            }
        }
        if (in_tile) {
            bool emits = false;
            process_particle<SPECIES, MODE, RHO, EXTRAS, SENDX, true, ES>(a, cp, s_field, s_acc, base, lane, kc, c, c_mem, xi, u, wt, prev, err, emits);
            if (EXTRAS == 2 && emits) emit_mask |= 1u << p;
        } else {
            pending |= 1u << p;
        }
    }
    // Particles outside the tile (drifted since the last re-binning, or fresh arrivals): rare.  Counted twice for the
    // re-binning policy: per particle, and per wave pass -- a wave goes through the slow path once per loop iteration
    // in which ANY of its lanes has a pending particle, whether that is one lane or all 64.  A tail chunk full of
    // fresh arrivals that all miss costs a handful of passes, not hundreds of isolated misses.
    if (OPAL_TWO_SIDED_TAIL && my_half >= 0 && pending) {  // (block-uniform first test) the other half's particles are not mine to push
        // (the other workgroup may be pushing such a particle right now: its cell reads as the old or the new one, on
        // the same side either way, or as DEAD_CELL if it has just left through that end)
        for (int p = 0; p < iters; p++) {
            if (!(pending & (1u << p))) continue;
            const int c = cp.cell[slot_of(p)];
            if (c == DEAD_CELL || (c >= (a.size >> 1)) != (my_half == 1)) pending &= ~(1u << p);
        }
    }
    unsigned any_pending = pending;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) any_pending |= __shfl_xor(any_pending, o);
    if (any_pending) {  // wave-uniform
        unsigned slow_wave = (unsigned)__popc(pending);  // this lane's, then the wave's
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) slow_wave += __shfl_xor(slow_wave, o);
        if (pending) {
            for (int p = 0; p < iters; p++) {
                if (!(pending & (1u << p))) continue;
                unsigned kc = slot_of(p);
                V3 u = {cp.ux[kc], cp.uy[kc], cp.uz[kc]};
                double prev = (MODE == MODE_DEPOSIT) ? cp.prev[kc] : 0.0;
                const int cm = cp.cell[kc];
                if (a.dbg) {
                    unsigned q = atomicAdd(reinterpret_cast<unsigned *>(a.dbg), 1u);
                    if (q < 4096) {
                        a.dbg[1 + 4 * q] = (int)block; a.dbg[2 + 4 * q] = (int)(chunk0 + kc);
                        a.dbg[3 + 4 * q] = cm; a.dbg[4 + 4 * q] = base - a.L;
                    }
                }
                bool emits = false;
                process_particle<SPECIES, MODE, RHO, EXTRAS, SENDX, false, false, ES && OPAL_ES_WATCH>(a, cp, s_field, s_acc, base, lane, kc, cm, cm, cp.x[kc], u, cp.w[kc], prev, err, emits);
                if (EXTRAS == 2 && emits) emit_mask |= 1u << p;
            }
        }
        // (a tile sent down this path whole by the electrostatic launch is not a miss)
        if ((tid & 63) == 0 && (!ES || tile_es)) {
            atomicAdd(&a.ctr->n_slow_pass, (unsigned)__popc(any_pending));
            atomicAdd(&a.ctr->n_slow, slow_wave);
        }
    }
    if (err) atomicOr(&a.ctr->err, err);

    if (EXTRAS == 2 && SPECIES == OPAL_ELECTRON && GATHER) {
        // The electrons that will radiate this step (interactions.rs:45-107 runs after the pushes) are known
        // here, with tau still in registers: compact their slots into the emitter list now -- an exclusive
        // prefix over the workgroup and ONE atomic on the list length -- instead of streaming cell and tau
        // again in a pass of their own.  s_min / s_max are free after the prologue.
        unsigned *s_cnt = reinterpret_cast<unsigned *>(s_max);
        unsigned *s_base = reinterpret_cast<unsigned *>(s_min);
        const int wl = tid & 63, wave = tid >> 6;
        unsigned mine = __popc(emit_mask), incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned v = __shfl_up(incl, o);
            if (wl >= o) incl += v;
        }
        if (wl == 63) s_cnt[wave] = incl;
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < BLOCK / 64; w2++) {
            if (w2 < wave) before += s_cnt[w2];
            total += s_cnt[w2];
        }
        if (total) {  // block-uniform
            if (tid == 0) s_base[0] = atomicAdd(&a.ctr->n_emit, total);
            __syncthreads();
            unsigned at = s_base[0] + before + incl - mine;
            for (int p = 0; p < iters; p++)
                if (emit_mask & (1u << p)) a.emit_list[at++] = (unsigned)(chunk0 + slot_of(p));
        }
    }

    if (DEPOSIT) {
        __syncthreads();
        // fold the REPL replicas of each (component, cell) and add to the grid: one lane per accumulator,
        // rotated start so that the 64 lanes of a wave read 64 different banks.
        constexpr int NACC = (RHO ? 4 : 3) * TILE;
        for (int k2 = tid; k2 < NACC; k2 += BLOCK) {
            double s = 0.0;
            const double *row = s_acc + k2 * REPL;
#pragma unroll 8
            for (int r = 0; r < REPL; r++) s += row[(r + tid) & (REPL - 1)];
            int comp = k2 / TILE, j = tile_index(base + (k2 % TILE));
            if (s != 0.0 && j >= 0 && j < a.len) atomicAdd(a.jr + (size_t)comp * a.lenp + j, s);
        }
    }
}

#ifndef OPAL_QED_WAVES
#define OPAL_QED_WAVES 4  // waves per SIMD the push that runs the optical depth down is compiled for (see process_particle's last lines)
#endif
template <int SPECIES, int MODE, bool RHO, int EXTRAS, bool SENDX>
__global__ __launch_bounds__(BLOCK, (EXTRAS == 0 && MODE == MODE_FUSED) ? 6 : (EXTRAS == 2 ? OPAL_QED_WAVES : 1)) void k_push(PushArgs a)
{
    __shared__ double s_field[MODE != MODE_DEPOSIT ? 6 * TILE : 1];
    __shared__ double s_acc[MODE != MODE_ADVANCE ? 4 * TILE * REPL : 1];
    __shared__ int s_min[BLOCK / 64], s_max[BLOCK / 64];
    push_body<SPECIES, MODE, RHO, EXTRAS, SENDX>(a, blockIdx.x, s_field, s_acc, s_min, s_max);
}

// Both charged species of a step in one launch: the first `nblk_e` workgroups take electrons, the rest
// ions.  One launch instead of two means one ragged tail of finishing workgroups per step instead of two
// (the species are independent: both read the fields and add into the same current arrays atomically).
#ifndef OPAL_PAIR_WAVES
#define OPAL_PAIR_WAVES 6  // measured: 6 waves/SIMD (80 VGPRs, no spills) beats 5 (84) and 4 (86) by 2-3 %; 7 spills
#endif
// Which workgroups (= chunks of BLOCK x ppt slots) of each species a launch covers: two runs of consecutive
// chunks per species.  A whole step is {0, all, 0, 0}; the edge launch of a split step takes the first and
// the last chunks, the bulk launch the ones in between (see pop_plan_split).
struct PairPlan {
    unsigned e0, en0, e1, en1;
    unsigned i0, in0, i1, in1;
};

#ifndef OPAL_ES_WAVES
#define OPAL_ES_WAVES 8  // the electrostatic launch fits 64 VGPRs: 8 waves per SIMD measured +2 % over 6 and 7
#endif
// (Tried, VERDICT r2 #6: reading the serving species' PushArgs THROUGH the kernel-argument segment at a workgroup-uniform
// offset -- `((const __attribute__((address_space(4))) PushArgs *)__builtin_amdgcn_kernarg_segment_ptr())[species]` --
// so that only one set of argument loads exists.  Scalar-register spills went from 111/100 to 101/96 in the electrostatic
// instantiations and from 16/15 to 10/8 in the general ones, and the kernels ran 0.6 % and 1.3 % SLOWER (same-box A/B):
// the spills are not the unused struct's -- push_body itself keeps ~120 uniform values alive against a budget of 80
// scalar registers at 8 waves per SIMD (102 at 7 or fewer: 54 spills, and 2 % slower for the lost wave).  DESIGN.md 4.1.)
template <int MODE, bool RHO, bool ES = false>
__global__ __launch_bounds__(BLOCK, ES ? OPAL_ES_WAVES : OPAL_PAIR_WAVES) void k_push_pair(PushArgs ae, PushArgs ai, PairPlan plan)
{
    __shared__ double s_field[MODE != MODE_DEPOSIT ? 6 * TILE : 1];
    __shared__ double s_acc[MODE != MODE_ADVANCE ? 4 * TILE * REPL : 1];
    __shared__ int s_min[BLOCK / 64], s_max[BLOCK / 64];
    unsigned b = blockIdx.x;
    const unsigned ne = plan.en0 + plan.en1;
    if (b < ne) {
        b = b < plan.en0 ? plan.e0 + b : plan.e1 + (b - plan.en0);
        push_body<OPAL_ELECTRON, MODE, RHO, 0, false, ES>(ae, b, s_field, s_acc, s_min, s_max);
    } else {
        b -= ne;
        b = b < plan.in0 ? plan.i0 + b : plan.i1 + (b - plan.in0);
        push_body<OPAL_ION, MODE, RHO, 0, false, ES>(ai, b, s_field, s_acc, s_min, s_max);
    }
}

// Photon::push without the machinery a photon does not need (photon.rs:150-183 with no chi / y / z tracked):
// a photon moves in a straight line at c, so the push reads cell, x and k (36 B), writes x and -- only where
// some lane of the wave changed cell -- cell (12 B at most), and touches neither the fields, nor an LDS tile,
// nor the weight.  Leavers are handled as in process_particle (deleted at open edges, packed into the link's
// message towards a neighbour: only then is the weight read).  No tile, so no tile misses: photons are
// re-binned only to squeeze out vacated slots (maybe_sort).
constexpr int PHOTON_PPT = 8;
#ifndef OPAL_PHOTON_GROUP
#define OPAL_PHOTON_GROUP 4
#endif
constexpr int PHOTON_GROUP = OPAL_PHOTON_GROUP;
static_assert(PHOTON_PPT % PHOTON_GROUP == 0, "whole groups");
template <bool SENDX>
__global__ __launch_bounds__(BLOCK) void k_push_photon(PushArgs a)
{
    const unsigned n = a.ctr->n_slots;
    const size_t chunk0 = (size_t)blockIdx.x * BLOCK * PHOTON_PPT;
    if (chunk0 >= n) return;
    int *cellp = a.cell + chunk0;
    double *xp = a.x + chunk0;
    const double *kxp = a.ux + chunk0, *kyp = a.uy + chunk0, *kzp = a.uz + chunk0;
    const unsigned nloc = (unsigned)min((size_t)BLOCK * PHOTON_PPT, (size_t)n - chunk0);
    int err = 0;
    // The slots are loaded PHOTON_GROUP at a time before any of them is processed: as one loop the compiler sinks a slot's
    // x and k loads behind the test of its cell (a dead slot needs neither), so that a wave had a 4-byte load in flight, waited,
    // then 32 bytes, waited -- two latencies per slot and 5.4 TB/s over the colliding-beams run (round 5's counters).  A slot
    // past the chunk's end re-loads its last one and is skipped below: the loads need no branch.
    for (int p0 = 0; p0 < PHOTON_PPT; p0 += PHOTON_GROUP) {
    int cs[PHOTON_GROUP];
    double xs[PHOTON_GROUP];
    V3 ks[PHOTON_GROUP];
#pragma unroll
    for (int q = 0; q < PHOTON_GROUP; q++) {
        const unsigned kl = min((unsigned)(p0 + q) * BLOCK + threadIdx.x, nloc - 1u);
        cs[q] = ldk(cellp, kl);
        xs[q] = ldk(xp, kl);
        ks[q] = V3{ldk(kxp, kl), ldk(kyp, kl), ldk(kzp, kl)};
    }
#pragma unroll
    for (int q = 0; q < PHOTON_GROUP; q++) {
        const unsigned kc = (unsigned)(p0 + q) * BLOCK + threadIdx.x;
        const int c = cs[q];
        const double xi = xs[q];
        const V3 k = ks[q];
        const bool live = kc < nloc && c != DEAD_CELL;
        const V3 none = {0.0, 0.0, 0.0};
        const Pushed r = photon_push<false>(k, none, none, a.dt_over_dx);
        double xn = xi + r.dxi;
        const double fl = floor(xn);
        const int s = min(max((int)fl, -1), 1);
        xn -= fl;
        // (asked of the whole wave, before anything depends on a slot being live: every lane's x and k are used here, so
        // their loads stay with the group's instead of moving behind the test of the cell)
        const bool stepped = __ballot(live && s != 0) != 0ull || a.out_shift != 0;
        if (!live) continue;
        if ((unsigned)c >= (unsigned)a.size) {  // yee.rs:501-502 (fields_at is still called by the reference)
            err |= 2;
            continue;
        }
        if (!(r.dxi < 1.0)) err |= 1;  // photon.rs:169
        int newc = c + s - a.out_shift;  // (the frame the result is stored in: PushArgs)
        bool deleted = false;
        int send_side = -1;
        if ((unsigned)newc >= (unsigned)a.out_size) {
            const bool left = newc < 0;
            const int mode = left ? a.left_mode : a.right_mode;
            if (mode == EDGE_WRAP) {
                newc += left ? a.out_size : -a.out_size;
            } else if (mode == EDGE_DELETE) {
                deleted = true;
            } else {
                send_side = left ? 0 : 1;
                if (!left) newc -= a.out_size;
            }
        }
        if (!deleted && send_side < 0) {
            if (a.left_mode == EDGE_SEND && newc < a.zone)
                atomicAdd(reinterpret_cast<unsigned *>(a.send_msg[0]) + a.zone_word, 1u);
            if (a.right_mode == EDGE_SEND && newc >= a.out_size - a.zone)
                atomicAdd(reinterpret_cast<unsigned *>(a.send_msg[1]) + a.zone_word, 1u);
            if (stepped && __ballot(newc != c)) stk(cellp, kc, newc);  // (no live lane left its cell: nothing to ask)
            stk(xp, kc, xn);
        } else if (deleted) {
            cellp[kc] = DEAD_CELL;
            atomicAdd(&a.ctr->n_deleted, 1u);
            atomicAdd(&a.ctr->n_dead, 1u);
        } else {
            atomicAdd(&a.ctr->n_dead, 1u);
            char *msg = a.send_msg[send_side];
            // a leaver stands within a cell of the cut on the far side after this exchange and may come back: it belongs
            // to the zone count of the edge it crosses (neither end's push would count it otherwise: link.cpp)
            if (a.zone) atomicAdd(reinterpret_cast<unsigned *>(msg) + a.zone_word, 1u);
            unsigned slot = 0;  // (one reservation per wave and side: process_particle has the reason)
#pragma unroll
            for (int sd = 0; sd < 2; sd++)
                if (send_side == sd) slot = atomicAdd(reinterpret_cast<unsigned *>(a.send_msg[sd]) + LINK_H_COUNT + a.send_slot, 1u);
            if (slot < a.send_cap[send_side]) {
                *link_cell(msg, a.send_layout, a.send_slot, slot) = newc;
                *link_field(msg, a.send_layout, a.send_slot, 0, slot) = xn;
                *link_field(msg, a.send_layout, a.send_slot, 1, slot) = k.x;
                *link_field(msg, a.send_layout, a.send_slot, 2, slot) = k.y;
                *link_field(msg, a.send_layout, a.send_slot, 3, slot) = k.z;
                *link_field(msg, a.send_layout, a.send_slot, 4, slot) = a.w[chunk0 + kc];
                if (SENDX) *link_field(msg, a.send_layout, a.send_slot, 6, slot) = (double)(chunk0 + kc);
            } else {
                err |= 8;
            }
            cellp[kc] = DEAD_CELL;
        }
    }
    }
    if (err) atomicOr(&a.ctr->err, err);
}

// ------------------------------------------------------------------ re-binning
// Counting sort by cell, tuned for nearly ordered input.  A workgroup takes SORT_PPT x 256 consecutive
// slots, whose cells fall (almost always) in a window of SORT_W cells above the chunk's lowest cell: the
// window is histogrammed in LDS and reaches global memory as one atomic per (workgroup, cell) instead
// of one per particle.  Cells outside the window (wrapped or freshly arrived particles) use global
// atomics directly.
constexpr int SORT_PPT = 8;
constexpr int SORT_W = 64;

__device__ inline int block_min_cell(const int (&c)[SORT_PPT], int *s_red)
{
    int m = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) m = min(m, c[p] == DEAD_CELL ? 0x7fffffff : c[p]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    return min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
}

__global__ __launch_bounds__(256) void k_sort_hist(const int *__restrict__ cell, const PopCounters *ctr,
                                                   unsigned *__restrict__ count)
{
    __shared__ unsigned s_cnt[SORT_W];
    __shared__ int s_red[4];
    const unsigned n = ctr->n_slots;
    const size_t chunk0 = (size_t)blockIdx.x * 256 * SORT_PPT;
    if (chunk0 >= n) return;
    int c[SORT_PPT];
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        size_t i = chunk0 + (size_t)p * 256 + threadIdx.x;
        c[p] = i < n ? cell[i] : DEAD_CELL;
    }
    if (threadIdx.x < SORT_W) s_cnt[threadIdx.x] = 0;
    const int wbase = block_min_cell(c, s_red);  // contains a barrier
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        // The lanes that hold the same cell as the wave's first live lane count as one: in a cold, dense population that is
        // the whole wave, and 64 increments of one LDS word (64 passes of the atomic unit) become one (round 4: the laser-plasma
        // run's re-binnings, cold ions and hot electrons in turn, 35.7 -> 22.7 us on average; hot electrons alone 12 -> 15).
        const bool live = c[p] != DEAD_CELL;
        const unsigned long long lm = __ballot(live);
        if (!lm) continue;  // wave-uniform
        const int lead = __ffsll((long long)lm) - 1;
        const int cl = __shfl(c[p], lead);
        const bool with_lead = live && c[p] == cl;
        const unsigned together = (unsigned)__popcll(__ballot(with_lead));
        if (live && (!with_lead || lane == lead)) {
            const unsigned k = with_lead ? together : 1u;
            const unsigned rel = (unsigned)(c[p] - wbase);
            if (rel < SORT_W) atomicAdd(&s_cnt[rel], k);
            else atomicAdd(&count[c[p]], k);
        }
    }
    __syncthreads();
    if (threadIdx.x < SORT_W && s_cnt[threadIdx.x]) atomicAdd(&count[wbase + threadIdx.x], s_cnt[threadIdx.x]);
}

// exclusive scan of count[0..ncells) in place, one workgroup of 1024; also the live total and non-empty cell count.
// Tiles of 4096 cells, four consecutive cells per thread (one 16-byte access each way, coalesced); a tile's sums are
// scanned across the workgroup with wave shuffles and one round through LDS, the running total carries over.
// (Round 4: 16.8 -> 9.4 us at 20 000 cells against a contiguous run of cells per thread, whose accesses were 80 bytes apart
// from lane to lane.)
__global__ __launch_bounds__(1024) void k_sort_scan(unsigned *count, int ncells, PopCounters *ctr)
{
    __shared__ unsigned s_wave[16], s_wave_ne[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned carry = 0, carry_ne = 0;
    for (int base = 0; base < ncells; base += 4096) {
        const int i0 = base + 4 * t;
        unsigned v[4] = {0u, 0u, 0u, 0u};
        if (i0 + 3 < ncells) {
            const uint4 q = *reinterpret_cast<const uint4 *>(count + i0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i0 + k < ncells) v[k] = count[i0 + k];
        }
        const unsigned sum = v[0] + v[1] + v[2] + v[3];
        const unsigned ne = (v[0] != 0) + (v[1] != 0) + (v[2] != 0) + (v[3] != 0);
        unsigned inc = sum, inc_ne = ne;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned a = __shfl_up(inc, o), b = __shfl_up(inc_ne, o);
            if (lane >= o) {
                inc += a;
                inc_ne += b;
            }
        }
        if (lane == 63) {
            s_wave[wave] = inc;
            s_wave_ne[wave] = inc_ne;
        }
        __syncthreads();
        unsigned before = 0, total = 0, total_ne = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            if (w < wave) before += s_wave[w];
            total += s_wave[w];
            total_ne += s_wave_ne[w];
        }
        const unsigned run = carry + before + inc - sum;  // exclusive prefix of this thread's four cells
        if (i0 + 3 < ncells) {
            *reinterpret_cast<uint4 *>(count + i0) = make_uint4(run, run + v[0], run + v[0] + v[1], run + v[0] + v[1] + v[2]);
        } else {
            unsigned r = run;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i0 + k < ncells) count[i0 + k] = r;
                r += v[k];
            }
        }
        carry += total;
        carry_ne += total_ne;
        __syncthreads();  // s_wave is written again by the next tile
    }
    if (t == 0) {
        ctr->n_live = carry;
        ctr->n_nonempty = carry_ne;
    }
}

struct SortArgs {
    const int *cell;
    const double *f[5];
    int *ocell;
    double *of[5];
    const double *prev;
    double *oprev;
    const double *ex[NEX];  // extras, each may be null
    double *oex[NEX];
};

__global__ __launch_bounds__(256) void k_sort_scatter(SortArgs a, const PopCounters *ctr, unsigned *__restrict__ cursor)
{
    __shared__ unsigned s_cnt[SORT_W], s_base[SORT_W];
    __shared__ int s_red[4];
    const unsigned n = ctr->n_slots;
    const size_t chunk0 = (size_t)blockIdx.x * 256 * SORT_PPT;
    if (chunk0 >= n) return;
    int c[SORT_PPT];
    unsigned rank[SORT_PPT];
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        size_t i = chunk0 + (size_t)p * 256 + threadIdx.x;
        c[p] = i < n ? a.cell[i] : DEAD_CELL;
    }
    if (threadIdx.x < SORT_W) s_cnt[threadIdx.x] = 0;
    const int wbase = block_min_cell(c, s_red);
    // local rank of every particle inside its cell's share of this workgroup
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        rank[p] = 0;
        if (c[p] == DEAD_CELL) continue;
        unsigned rel = (unsigned)(c[p] - wbase);
        if (rel < SORT_W) rank[p] = atomicAdd(&s_cnt[rel], 1u);
    }
    __syncthreads();
    // one global reservation per (workgroup, cell)
    if (threadIdx.x < SORT_W) {
        unsigned k = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = k ? atomicAdd(&cursor[wbase + threadIdx.x], k) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        if (c[p] == DEAD_CELL) continue;
        size_t i = chunk0 + (size_t)p * 256 + threadIdx.x;
        unsigned rel = (unsigned)(c[p] - wbase);
        unsigned d = rel < SORT_W ? s_base[rel] + rank[p] : atomicAdd(&cursor[c[p]], 1u);
        a.ocell[d] = c[p];
#pragma unroll
        for (int q = 0; q < 5; q++) a.of[q][d] = a.f[q][i];
        if (a.prev) a.oprev[d] = a.prev[i];
#pragma unroll
        for (int q = 0; q < NEX; q++)
            if (a.ex[q]) a.oex[q][d] = a.ex[q][i];
    }
}

// The same scatter with every global access a streaming one.  In k_sort_scatter a wave's 64 particles go to as many runs
// of the output as they hold distinct cells -- a population that has drifted for a dozen steps holds twenty -- so each
// store instruction touches twenty cache lines for 8 bytes apiece.  Here the workgroup first orders its particles locally:
// a particle's place among the workgroup's in-window particles (cells in order, arrival order within a cell) indexes an LDS
// buffer, one attribute at a time is staged there, and output position j is written by thread j mod 256: consecutive
// positions of one cell are consecutive in the output too, so the stores coalesce as the loads do.  Particles outside the
// window (wrapped, freshly arrived: rare) go straight to their slot as before.  Round 4, 1e7 particles up to 13 cells from
// their slot's cell: 451 -> 200 us (5.2 TB/s); in the laser-plasma interaction 220 -> 180 (profiles/r04_ab/ab_sort_staged.txt).
constexpr int SORT_N = 256 * SORT_PPT;
__global__ __launch_bounds__(256) void k_sort_scatter_staged(SortArgs a, const PopCounters *ctr, unsigned *__restrict__ cursor)
{
    __shared__ unsigned s_cnt[SORT_W], s_base[SORT_W], s_off[SORT_W + 1];
    __shared__ int s_red[4];
    __shared__ unsigned s_dst[SORT_N];
    __shared__ double s_val[SORT_N];
    const unsigned n = ctr->n_slots;
    const size_t chunk0 = (size_t)blockIdx.x * SORT_N;
    if (chunk0 >= n) return;
    const int tid = threadIdx.x;
    constexpr unsigned NONE = 0xffffffffu, DIRECT = 0xfffffffeu;
    int c[SORT_PPT];
    unsigned li[SORT_PPT];    // place in the staging buffer; NONE: a vacated slot; DIRECT: outside the window
    unsigned dd[SORT_PPT];    // DIRECT only: the slot it goes to
    unsigned dj[SORT_PPT];    // where output position p * 256 + tid of the staging buffer goes
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        size_t i = chunk0 + (size_t)p * 256 + tid;
        c[p] = i < n ? a.cell[i] : DEAD_CELL;
    }
    if (tid < SORT_W) s_cnt[tid] = 0;
    const int wbase = block_min_cell(c, s_red);
    const int lane = tid & 63;
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        li[p] = NONE;
        dd[p] = 0;
        // (rank in its cell for now.)  The lanes that share the wave's first live lane's cell take their ranks from ONE
        // increment of the cell's counter (k_sort_hist has the reason); the others one each.
        const bool live = c[p] != DEAD_CELL;
        const unsigned long long lm = __ballot(live);
        if (!lm) continue;  // wave-uniform
        const int lead = __ffsll((long long)lm) - 1;
        const int cl = __shfl(c[p], lead);
        const bool with_lead = live && c[p] == cl;
        const unsigned long long same = __ballot(with_lead);
        const unsigned rel = (unsigned)(c[p] - wbase);
        unsigned first = 0;
        if (live && rel < SORT_W && (!with_lead || lane == lead)) first = atomicAdd(&s_cnt[rel], with_lead ? (unsigned)__popcll(same) : 1u);
        const unsigned lead_first = __shfl(first, lead);
        if (!live) continue;
        if (rel >= SORT_W) li[p] = DIRECT;
        else li[p] = with_lead ? lead_first + (unsigned)__popcll(same & ((1ull << lane) - 1ull)) : first;
    }
    __syncthreads();
    if (tid < SORT_W) {  // (one wave) a global reservation per (workgroup, cell), and where each cell's run starts locally
        const unsigned k = s_cnt[tid];
        s_base[tid] = k ? atomicAdd(&cursor[wbase + tid], k) : 0u;
        unsigned inc = k;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned v = __shfl_up(inc, o);
            if (tid >= o) inc += v;
        }
        s_off[tid] = inc - k;
        if (tid == SORT_W - 1) s_off[SORT_W] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        if (li[p] == NONE) continue;
        if (li[p] == DIRECT) {
            dd[p] = atomicAdd(&cursor[c[p]], 1u);
            continue;
        }
        unsigned rel = (unsigned)(c[p] - wbase);
        unsigned d = s_base[rel] + li[p];
        li[p] += s_off[rel];
        s_dst[li[p]] = d;
    }
    const unsigned nw = s_off[SORT_W];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < SORT_PPT; p++) {
        unsigned j = (unsigned)p * 256 + tid;
        dj[p] = j < nw ? s_dst[j] : NONE;
    }
    // the cell index
    {
        int *s_ival = reinterpret_cast<int *>(s_val);
#pragma unroll
        for (int p = 0; p < SORT_PPT; p++) {
            if (li[p] == NONE) continue;
            if (li[p] == DIRECT) a.ocell[dd[p]] = c[p];
            else s_ival[li[p]] = c[p];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < SORT_PPT; p++)
            if (dj[p] != NONE) a.ocell[dj[p]] = s_ival[(unsigned)p * 256 + tid];
        __syncthreads();
    }
    auto move = [&](const double *__restrict__ src, double *__restrict__ dst) {  // block-uniform calls
        double v[SORT_PPT];
#pragma unroll
        for (int p = 0; p < SORT_PPT; p++) {
            size_t i = chunk0 + (size_t)p * 256 + tid;
            v[p] = li[p] != NONE ? src[i] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < SORT_PPT; p++) {
            if (li[p] == NONE) continue;
            if (li[p] == DIRECT) dst[dd[p]] = v[p];
            else s_val[li[p]] = v[p];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < SORT_PPT; p++)
            if (dj[p] != NONE) dst[dj[p]] = s_val[(unsigned)p * 256 + tid];
        __syncthreads();
    };
#pragma unroll
    for (int q = 0; q < 5; q++) move(a.f[q], a.of[q]);
    if (a.prev) move(a.prev, a.oprev);
#pragma unroll
    for (int q = 0; q < NEX; q++)
        if (a.ex[q]) move(a.ex[q], a.oex[q]);
}

// Chunk table (see opal_pop): the sorted merge of two monotone boundary lists, B1_k = min(k * chunk, N), k = 1..K1
// (every 256 x 8 slots) and B2_j = cell_end[min(W j, ncells) - 1], j = 1..K2 (every W cells), computed in parallel:
// a boundary's place in the merged list is its own index plus the number of boundaries of the other list in front
// of it (ties: B1 first).  cell_end[c] is one past the last slot of cell c, as the scatter's cursors stand.
constexpr int CHUNK_CELLS = 24;  // a chunk spans at most this many cells: (TILE - CHUNK_CELLS) / 2 - 3 = 17 cells of
                                 // drift relative to the chunk's bulk, either way, before a particle misses the tile
__device__ inline unsigned b2_at(const unsigned *cell_end, int ncells, unsigned j)
{
    return cell_end[min((int)(CHUNK_CELLS * j), ncells) - 1];
}
// N, K1: read from the device (ctr->n_live).  The host sizes the table and the launch with K1_ub >= K1 from its upper
// bound of the slot count (a re-binning need not wait for the device to tell it the true one): the K1_ub - K1 entries
// behind the merged list are set to N -- empty chunks, whose workgroups leave at once.
__global__ void k_chunk_table(const unsigned *__restrict__ cell_end, int ncells, unsigned chunk, const PopCounters *ctr,
                              unsigned K1_ub, unsigned K2, unsigned *__restrict__ tab)
{
    const unsigned N = ctr->n_live;
    const unsigned K1 = (N + chunk - 1) / chunk;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) tab[0] = 0;
    if (N == 0) {
        if (t < K1_ub + K2) tab[t + 1] = 0;
        return;
    }
    if (t < K1_ub) {
        const unsigned k = t + 1;
        if (k > K1) {
            tab[k + K2] = N;
            return;
        }
        const unsigned b = k == K1 ? N : k * chunk;
        unsigned lo = 1, hi = K2 + 1;  // number of j in [1, K2] with B2_j < b: B2 is non-decreasing in j
        while (lo < hi) {
            unsigned mid = (lo + hi) / 2;
            if (b2_at(cell_end, ncells, mid) < b) lo = mid + 1; else hi = mid;
        }
        tab[k + (lo - 1)] = b;
    } else if (t < K1_ub + K2) {
        const unsigned j = t - K1_ub + 1;
        const unsigned b = b2_at(cell_end, ncells, j);
        const unsigned before = min(K1 - 1, b / chunk) + (b >= N ? 1u : 0u);  // number of k with B1_k <= b
        tab[j + before] = b;
    }
}
__global__ void k_publish_counters(PublishBatch b)
{
    static_assert(sizeof(PopCounters) == 64, "one word per lane below");
    const int e = threadIdx.x >> 4, w = threadIdx.x & 15;
    if (e < b.n) reinterpret_cast<volatile unsigned *>(&b.dst[e]->c)[w] = reinterpret_cast<const unsigned *>(b.src[e])[w];
    __threadfence_system();
    if (e < b.n && w == 0) b.dst[e]->seq = b.seq[e];
}

__global__ void k_sort_finish(PopCounters *ctr)
{
    ctr->n_slots = ctr->n_live;
    ctr->n_sorted = ctr->n_live;
    ctr->n_slow = 0;
    ctr->n_slow_pass = 0;
    ctr->n_dead = 0;
}

// ------------------------------------------------------------- kinetic energy
// electron.rs:122-126 / ion.rs:128-134 / photon.rs:222-231; per-workgroup partial sums in a fixed tree, then one
// workgroup folds the partials: the result does not depend on scheduling.
template <int SPECIES>
__global__ void k_kinetic(const int *__restrict__ cell, const double *__restrict__ ux, const double *__restrict__ uy,
                          const double *__restrict__ uz, const double *__restrict__ w, const PopCounters *ctr,
                          double a_mass, double *__restrict__ partial)
{
    __shared__ double s[BLOCK];
    unsigned n = ctr->n_slots;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        if (cell[i] == DEAD_CELL) continue;
        double u2 = ux[i] * ux[i] + uy[i] * uy[i] + uz[i] * uz[i];
        double g = sqrt(1.0 + u2);
        if (SPECIES == OPAL_ELECTRON) {
            double to_joules = 1.0e6 * ELECTRON_MASS_MEV * ELEMENTARY_CHARGE;
            acc += w[i] * u2 * to_joules / (g + 1.0);
        } else if (SPECIES == OPAL_PHOTON) {  // photon.rs:222-231
            acc += w[i] * (sqrt(u2) * ELECTRON_MASS_MEV) * 1.0e6 * ELEMENTARY_CHARGE;
        } else {
            double gamma_m1 = u2 / (1.0 + g);
            double energy = gamma_m1 * (a_mass * PROTON_MASS / ELECTRON_MASS) * ELECTRON_MASS_MEV;
            acc += w[i] * energy * 1.0e6 * ELEMENTARY_CHARGE;
        }
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = BLOCK / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

__global__ void k_fold(const double *__restrict__ partial, int n, double *__restrict__ out)
{
    __shared__ double s[BLOCK];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) acc += partial[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = BLOCK / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

// T::create, electron.rs:85-103 / ion.rs:81-101 / photon.rs:95-98: prev_x from the initial momentum
template <int SPECIES>
__global__ void k_init_prev(const int *cell, const double *x, const double *ux, const double *uy, const double *uz,
                            double *prev_x, unsigned n, double dt, double dx)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double u2 = ux[i] * ux[i] + uy[i] * uy[i] + uz[i] * uz[i];
    if (SPECIES == OPAL_ELECTRON) {
        double gamma = sqrt(1.0 + u2);
        prev_x[i] = x[i] - SPEED_OF_LIGHT * ux[i] * dt / (gamma * dx);
    } else if (SPECIES == OPAL_PHOTON) {  // photon.rs:95-98
        prev_x[i] = x[i] - SPEED_OF_LIGHT * ux[i] * dt / (sqrt(u2) * dx);
    } else {
        double gamma_m1 = u2 / (1.0 + sqrt(1.0 + u2));
        double vx = SPEED_OF_LIGHT * ux[i] / (1.0 + gamma_m1);
        prev_x[i] = x[i] - vx * dt / dx;
    }
}

// Arrivals from the neighbours (mod.rs:365-373): appended behind the slots in use.  Block 0 takes what
// the left neighbour sent right (cells already re-based by the sender, mod.rs:291-293), block 1 what the
// right neighbour sent left (cell -1 -> += this subdomain's size, mod.rs:366-368).
struct UnpackArgs {
    const char *recv[2];
    LinkLayout layout;
    int slot;
    unsigned recv_cap[2];  // slots per species the message of that edge shipped
    unsigned reserve[2];   // arrivals per species this population made room for through that edge (link_plan's bound)
    int *cell;
    double *f[5];
    double *prev_x;
    double *ex[NEX];  // extras, each may be null
    unsigned nextra;  // how many of them the message carries behind the 6 base fields (0: arrivals restart from zero)
    PopCounters *ctr;
    unsigned pop_cap;
    int my_size;
    unsigned *emit_list;  // electrons of an emission run: arrivals whose optical depth has run out join the list
    int w_uniform;        // the population's weights are all w0 so far: an arrival with another raises ctr->w_mixed
    double w0;
    int *es_flag;         // an arrival with transverse momentum ends the electrostatic launches (ctr->hot_yz, grid flag)
};

// The tail of a linked step, one launch: up to three species' arrivals (blocks 2 s and 2 s + 1 belong to species s), in
// one more workgroup YeeGrid::synchronize (has_sync) followed by the edge tiles of the field solve (has_adv), and one
// workgroup per interior tile of the field solve behind them.  With a peer link (peer.enabled) the launch IS the exchange: workgroup 0 first writes this subdomain's two
// messages -- header, the halo packed straight from the grid rows, the row blocks the leavers occupy -- into the
// neighbours' mailboxes, and every workgroup waits for the neighbours' messages before it reads them (peer_device.hpp).
// The first workgroup also files this exchange's zone counts -- mine from the send headers, the neighbours' from the
// received ones -- in the link's mapped host ring (link.cpp: transports whose message sizes the host plans), zeroes the
// send headers for the next step (they have been shipped) and publishes the re-binning policy's counters when a check
// is due (pub.n: what k_publish_counters does on the unlinked path).
struct UnpackBatch {
    UnpackArgs a[3];
    int n;
    int has_sync;   // one more workgroup runs YeeGrid::synchronize's kernel (the fused linked step: independent work)
    SyncArgs sync;
    int has_adv;    // ... and then the edge tiles of YeeGrid::advance; further workgroups take the interior tiles (grid_device.hpp)
    AdvArgs adv;
    PeerArgs peer;
    PublishBatch pub;
    unsigned *zero[2];
    const unsigned *send_head[2], *recv_head[2];  // null where there is no neighbour
    LinkBounds *bounds;                            // ring entry of this exchange (mapped host memory), or null
    unsigned seq;
    int zone_at;                                   // header word of this exchange's zone counts (link_h_zone)
    int parts;      // workgroups per (species, side): 1, or TAIL_PARTS where a front may be crossing (unpack_arrivals)
    unsigned long long *first_slot;  // parts > 1: one word per (species, side), (exchange << 32) | first slot of its arrivals
};
constexpr int TAIL_PARTS = 8;

__device__ void unpack_one(const UnpackArgs &a, int side, unsigned part, unsigned of, unsigned long long *first_slot, unsigned seq);

// (8 waves per SIMD asked for: two of these 16-wave workgroups per CU.  At the 7 the compiler otherwise settles on -- 106 scalar
// registers -- a CU takes ONE, the chip 256, and six subdomains rehearsing on one card need up to 6 x 65 in the step that
// hands a re-cut over: the workgroups that wait for the neighbours' messages then keep the neighbours' senders off the chip
// until the timeout.  A subdomain that has its card to itself never needs more than 65 + its interior tiles.)
template <int NT, bool ADV>
__global__ __launch_bounds__(NT, NT >= 1024 ? 8 : 1) void k_link_tail(UnpackBatch b)
{
    static_assert(!ADV || NT == ADV_TILE, "the field solve's workgroups are ADV_TILE threads");
    __shared__ double s_adv[ADV ? 2 * 6 * (ADV_EDGE + 2 * ADV_HALO) : 1];
    const unsigned parts = (unsigned)b.parts;
    const unsigned n_unpack = b.n ? 2u * (unsigned)b.n * parts : 1u;
    if (ADV && blockIdx.x > n_unpack) {
        // an interior tile of the field solve: nothing synchronize or the neighbours' messages touch -- no waiting
        advance_interior(b.adv, (int)(blockIdx.x - n_unpack) - 1, s_adv);
        return;
    }
    // the workgroup that runs synchronize (workgroup 0 where none does) also keeps the link's books.  It is the one
    // that SENDS over a peer link: it reads this subdomain's edge cells for the halo, and must have done so before it
    // overlays the neighbours' halos on them
    const bool boss = b.has_sync ? blockIdx.x == n_unpack : blockIdx.x == 0;
    // (species index, side, share) of a workgroup that unpacks
    const unsigned pair = blockIdx.x / parts, part = blockIdx.x % parts;
    if (b.peer.enabled) {
        if (parts == 1) {
            if (boss) peer_send(b.peer);  // (ends behind a barrier: the send messages have been read)
        } else {
            // A cut a front may be crossing: the message is shared out.  The workgroup that runs synchronize sends header,
            // halo and the row blocks of species slots nobody unpacks; the workgroups of (species, side) send that species'
            // row blocks of the message that LEAVES through that side, a share each; the last ticket publishes.
            const bool unpacker = b.n && blockIdx.x < n_unpack;
            if (boss) {
                unsigned covered = 0;
                for (int i = 0; i < b.n; i++) covered |= 1u << b.a[i].slot;
                peer_send_head(b.peer, ((1u << b.peer.layout.nslots) - 1u) & ~covered);
            }
            if (unpacker) peer_send_rows(b.peer, (int)(pair & 1u), (unsigned)b.a[pair >> 1].slot, part, parts);
            if (boss || unpacker)
                if (peer_publish(b.peer, b.peer.err + 1, (b.n ? n_unpack : 0u) + (b.has_sync || !b.n ? 1u : 0u)))
                    // every sender has read the leaver counts: they (and the next push's zone counts) can be zeroed -- by
                    // whoever knows that, not by the workgroup that runs synchronize at a time of its own
                    for (int w = 0; w < 16; w++)
                        if (w < b.zone_at || w >= b.zone_at + 4) {
                            if (b.zero[0]) b.zero[0][w] = 0u;
                            if (b.zero[1]) b.zero[1][w] = 0u;
                        }
        }
        peer_wait(b.peer);
    }
    if (boss && !(b.peer.enabled && parts > 1) && threadIdx.x < 16 && (threadIdx.x < b.zone_at || threadIdx.x >= b.zone_at + 4)) {
        // the leaver counts have been shipped, and the OTHER set of zone counts is the next push's; this exchange's set
        // stays as it is until the next tail (the first workgroup is reading it)
        if (b.zero[0]) b.zero[0][threadIdx.x] = 0u;
        if (b.zero[1]) b.zero[1][threadIdx.x] = 0u;
    }
    // (the policy's counters go out from workgroup 0: where another workgroup runs synchronize and the field solve's
    // edges -- the longest chain of the launch -- the ~2 us of the write to host memory are off it)
    if (blockIdx.x == 0 && (b.pub.n || b.pub.zero)) publish_counters(b.pub);
    if (blockIdx.x == 0 && b.bounds) {
        // this exchange's zone counts -- mine from the send headers, the neighbours' from the received ones -- go to the
        // link's ring in mapped host memory (link.cpp: link_plan) from the first workgroup: the write and its
        // system-scope fence cost about a microsecond, which the workgroup that runs synchronize does not have
        if (threadIdx.x < 6) {
            const int side = threadIdx.x / 3, sp = threadIdx.x % 3;
            b.bounds->mine[side][sp] = b.send_head[side] ? b.send_head[side][b.zone_at + sp] : 0u;
            b.bounds->theirs[side][sp] = b.recv_head[side] ? b.recv_head[side][b.zone_at + sp] : 0u;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            b.bounds->seq = b.seq;
        }
    }
    if (blockIdx.x >= n_unpack) {  // (only launched when has_sync)
        // (with the field solve in this launch: synchronize and the solve of the cells it touches, and of their stencil
        // neighbours, as one -- grid_device.hpp)
        if (ADV && b.has_adv) sync_advance_edges(b.sync, b.adv, s_adv);
        else synchronize_body(b.sync);
        return;
    }
    if (b.n) unpack_one(b.a[pair >> 1], (int)(pair & 1u), part, parts, b.first_slot ? b.first_slot + pair : nullptr, b.peer.seq);
}

// arrivals of one species through one side: share `part` of `of` (whole row blocks each).  The shares of one (species, side)
// land in ONE run of slots, in message order -- the re-binning's windows and the arrival chunks' tiles count on arrivals
// from one neighbour standing together (interleaved by whichever workgroup came first, the same arrivals cost the counting
// sort 65 % more) --: share 0 reserves the run and publishes its first slot with the exchange number; the others wait for
// that word (a launch of a few dozen workgroups, all resident; bounded like every wait here).
__device__ void unpack_one(const UnpackArgs &a, int side, unsigned part, unsigned of, unsigned long long *first_slot, unsigned seq)
{
    __shared__ unsigned s_base;
    const char *msg = a.recv[side];
    if (!msg) return;
    unsigned total = reinterpret_cast<const unsigned *>(msg)[LINK_H_COUNT + a.slot];
    if (total > a.recv_cap[side]) total = a.recv_cap[side];  // the sender has flagged the overflow on its side
    // (a peer link ships whatever leaves: more arrivals than the bound the host grew its slot count by would leave
    // pushes sized from that count short of the last slots -- reported, not absorbed)
    if (total > a.reserve[side] && threadIdx.x == 0 && part == 0) atomicOr(&a.ctr->err, 128);
    const unsigned per = ((total + of - 1) / of + LINK_ROWS - 1) / LINK_ROWS * LINK_ROWS;
    const unsigned first = part * per;
    if (first >= total) return;
    const unsigned count = total - first < per ? total - first : per;
    if (threadIdx.x == 0) {
        if (of == 1) {
            s_base = atomicAdd(&a.ctr->n_slots, total);
        } else if (part == 0) {
            s_base = atomicAdd(&a.ctr->n_slots, total);
            __hip_atomic_store(first_slot, ((unsigned long long)seq << 32) | s_base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            SpinClock clock;  // (peer_device.hpp: the time spent polling, not the difference of two readings)
            unsigned long long v;
            while (((v = __hip_atomic_load(first_slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != seq) {
                if (clock.expired()) {
                    atomicOr(&a.ctr->err, 16);
                    v = (unsigned long long)a.pop_cap;  // (past the end: this share is dropped, the error ends the run)
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_base = (unsigned)v;
        }
    }
    __syncthreads();
    unsigned base = s_base;
    if (base + total > a.pop_cap || base >= a.pop_cap) {
        if (threadIdx.x == 0 && part == 0) {
            atomicOr(&a.ctr->err, 16);
            atomicSub(&a.ctr->n_slots, total);
        }
        return;
    }
    // (slot of arrival k is base + k below)
    for (unsigned k = first + threadIdx.x; k < first + count; k += blockDim.x) {
        a.cell[base + k] = *link_cell(msg, a.layout, a.slot, k) + (side == 1 ? a.my_size : 0);
#pragma unroll
        for (int c = 0; c < 5; c++) a.f[c][base + k] = *link_field(msg, a.layout, a.slot, c, k);
        if (a.w_uniform && *link_field(msg, a.layout, a.slot, 4, k) != a.w0) a.ctr->w_mixed = 1u;
        if (a.ctr->hot_yz == 0 &&
            (*link_field(msg, a.layout, a.slot, 2, k) != 0.0 || *link_field(msg, a.layout, a.slot, 3, k) != 0.0)) {
            a.ctr->hot_yz = 1u;
            *a.es_flag = 1;
        }
        if (a.prev_x) a.prev_x[base + k] = *link_field(msg, a.layout, a.slot, 5, k);
        // a plain link's message does not carry the extras (diagnostics only: a particle that changes
        // subdomain restarts them); a link laid out for photon emission carries all of them
#pragma unroll
        for (int q = 0; q < NEX; q++)
            if (a.ex[q]) a.ex[q][base + k] = q < (int)a.nextra ? *link_field(msg, a.layout, a.slot, 6 + q, k) : 0.0;
        // an electron that crossed over with its optical depth already below zero radiates here (the reference
        // migrates before emit_radiation, main.rs:240-252): rare, one atomic each
        if (a.emit_list && a.nextra > EX_TAU && *link_field(msg, a.layout, a.slot, 6 + EX_TAU, k) < 0.0)
            a.emit_list[atomicAdd(&a.ctr->n_emit, 1u)] = base + k;
    }
}

// Fills the extras columns of the leavers packed by the push (which left each leaver's slot number in the
// first extras column): one workgroup per side.
struct PackExtrasArgs {
    char *send[2];
    LinkLayout layout;
    int slot;
    unsigned send_cap[2], nextra;
    const double *ex[NEX];
};
__global__ void k_pack_extras(PackExtrasArgs a)
{
    char *msg = a.send[blockIdx.x];
    if (!msg) return;
    unsigned count = min(reinterpret_cast<const unsigned *>(msg)[LINK_H_COUNT + a.slot], a.send_cap[blockIdx.x]);
    for (unsigned k = threadIdx.x; k < count; k += blockDim.x) {
        size_t i = (size_t)*link_field(msg, a.layout, a.slot, 6, k);
#pragma unroll
        for (int q = 0; q < NEX; q++)
            if (q < (int)a.nextra) *link_field(msg, a.layout, a.slot, 6 + q, k) = a.ex[q] ? a.ex[q][i] : 0.0;
    }
}

// A re-cut of the decomposition (sim.cpp: sim_recut): the cells this subdomain hands to its neighbours change hands with
// everything in them.  Particles in the first `give_left` cells (the last cells, from `first_right` on) are packed into the
// link's send messages exactly as a push packs its leavers -- cell index in the receiver's frame: negative for the left
// neighbour, which adds its new size (mod.rs:366-368), from zero for the right one (mod.rs:291-293), all attributes, prev_x
// and the extras the link carries -- and their slots vacated; everybody else's cell index moves to the new frame.
struct RecutPackArgs {
    int *cell;
    const double *f[5];
    const double *prev;
    const double *ex[NEX];
    unsigned n;          // slots in use
    int give_left;       // cells [0, give_left) go to the left neighbour
    int first_right;     // cells [first_right, size) go to the right neighbour
    int delta;           // a stayer's cell index changes by this much: cells taken over on the left minus cells given away there
    char *send[2];
    unsigned cap[2];
    LinkLayout layout;
    int slot;
    unsigned nextra;
    PopCounters *ctr;
};
__global__ __launch_bounds__(256) void k_recut_pack(RecutPackArgs a)
{
    const int lane = threadIdx.x & 63;
    const unsigned n = min(a.n, a.ctr->n_slots);  // (a.n: the host's upper bound of the slots in use)
    const unsigned n_wave_aligned = (n + 63u) & ~63u;
    int err = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_wave_aligned; i += gridDim.x * blockDim.x) {
        const int c = i < n ? a.cell[i] : DEAD_CELL;
        int side = -1;
        if (c != DEAD_CELL) {
            if (c < a.give_left) side = 0;
            else if (c >= a.first_right) side = 1;
            else if (a.delta) a.cell[i] = c + a.delta;
        }
        for (int sd = 0; sd < 2; sd++) {  // (wave-uniform loop: one reservation per wave and side)
            const unsigned long long m = __ballot(side == sd);
            if (!m) continue;
            unsigned base = 0;
            const int lead = __ffsll((long long)m) - 1;
            if (lane == lead) base = atomicAdd(reinterpret_cast<unsigned *>(a.send[sd]) + LINK_H_COUNT + a.slot, (unsigned)__popcll(m));
            base = __shfl(base, lead);
            if (side != sd) continue;
            const unsigned k = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            a.cell[i] = DEAD_CELL;
            if (k >= a.cap[sd]) {
                err |= 8;
                continue;
            }
            char *msg = a.send[sd];
            *link_cell(msg, a.layout, a.slot, k) = sd == 0 ? c - a.give_left : c - a.first_right;
#pragma unroll
            for (int q = 0; q < 5; q++) *link_field(msg, a.layout, a.slot, q, k) = a.f[q][i];
            *link_field(msg, a.layout, a.slot, 5, k) = a.prev ? a.prev[i] : 0.0;
            for (int q = 0; q < NEX; q++)
                if (q < (int)a.nextra) *link_field(msg, a.layout, a.slot, 6 + q, k) = a.ex[q] ? a.ex[q][i] : 0.0;
        }
        const unsigned long long gone = __ballot(side >= 0);
        if (gone && lane == 0) atomicAdd(&a.ctr->n_dead, (unsigned)__popcll(gone));
    }
    if (err) atomicOr(&a.ctr->err, err);
}

int check_pop(const opal_pop_t *p)
{
    if (!p) {
        set_error("null population handle");
        return OPAL_ERR_ARG;
    }
    return OPAL_OK;
}

// Waits on the step path poll the event instead of sleeping in hipEventSynchronize / hipStreamSynchronize: the
// interrupt-driven wake-up can take long on a loaded host, and while the host is late the device runs dry.
int spin_wait(hipEvent_t ev)
{
    for (;;) {
        hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return OPAL_OK;
        if (q != hipErrorNotReady) OPAL_HIP(q);
    }
}

int fetch_counters(const opal_pop_t *p)
{
    OPAL_HIP(hipMemcpyAsync(p->h_ctr, p->d_ctr, sizeof(PopCounters), hipMemcpyDeviceToHost, p->stream));
    if (!p->fetch_event) OPAL_HIP(hipEventCreateWithFlags(const_cast<hipEvent_t *>(&p->fetch_event), hipEventDisableTiming));
    OPAL_HIP(hipEventRecord(p->fetch_event, p->stream));
    return spin_wait(p->fetch_event);
}

int check_pop_err(opal_pop_t *p)
{
    if (int rc = fetch_counters(p)) return rc;
    int err = p->h_ctr->err;
    if (p->h_ctr->n_lost) {
        OPAL_HIP(hipMemsetAsync(&p->d_ctr->n_lost, 0, sizeof(unsigned), p->stream));
        set_error("%u emitted photons found no free slot", p->h_ctr->n_lost);
        return OPAL_ERR_CAPACITY;
    }
    if (err) {
        OPAL_HIP(hipMemsetAsync(&p->d_ctr->err, 0, sizeof(int), p->stream));
        if (err & 8) {
            set_error("leaver buffer overflow: more particles crossed a subdomain edge in one step than this step's message had "
                      "room for -- the link's particle_capacity, or the size predicted from the particles near the edge "
                      "(link.cpp: link_plan; opal_link_set_bounded(link, 0) ships every slot instead)");
            return OPAL_ERR_CAPACITY;
        }
        if (err & 16) {
            set_error("population capacity exceeded by arrivals from a neighbouring subdomain");
            return OPAL_ERR_CAPACITY;
        }
        if (err & 128) {
            set_error("more particles arrived through a subdomain edge in one step than the bound the population reserved room "
                      "by (link.cpp: link_plan: the particles within the zone on both sides of the cut a few exchanges earlier)");
            return OPAL_ERR_ASSERT;
        }
        if (err & 64) {
            set_error("a photon-electron pair gave a negative or non-finite interaction probability in absorb "
                      "(photon.rs:325,336 assert!(partial_prob >= 0.0))");
            return OPAL_ERR_ASSERT;
        }
        if (err & 32) {
            set_error("photon emission at chi >= 100, which the reference's sampler does not implement "
                      "(photon_emission.rs:143-144 unimplemented!())");
            return OPAL_ERR_ASSERT;
        }
        char where[64] = "";
        if (err & 2) snprintf(where, sizeof where, " [a particle in cell %d]", (int)p->h_ctr->bad_cell);
        set_error("particle invariant violated on device (flags 0x%x):%s%s%s%s", err,
                  (err & 1) ? " moved >= 1 cell in a step (electron.rs:313 / ion.rs:199);" : "",
                  (err & 2) ? " gather index out of range (yee.rs:501-502);" : "", where,
                  (err & 4) ? " deposit index out of range (yee.rs:578-579);" : "");
        return OPAL_ERR_ASSERT;
    }
    return OPAL_OK;
}

int alloc_set(opal_pop_t *p, int k)
{
    size_t c = p->cap;
    OPAL_HIP(hipMalloc(&p->d_cell[k], sizeof(int) * c));
    OPAL_HIP(hipMalloc(&p->d_x[k], sizeof(double) * c));
    OPAL_HIP(hipMalloc(&p->d_ux[k], sizeof(double) * c));
    OPAL_HIP(hipMalloc(&p->d_uy[k], sizeof(double) * c));
    OPAL_HIP(hipMalloc(&p->d_uz[k], sizeof(double) * c));
    OPAL_HIP(hipMalloc(&p->d_w[k], sizeof(double) * c));
    return OPAL_OK;
}

void free_set(opal_pop_t *p, int k)
{
    hipFree(p->d_cell[k]); hipFree(p->d_x[k]); hipFree(p->d_ux[k]);
    hipFree(p->d_uy[k]); hipFree(p->d_uz[k]); hipFree(p->d_w[k]);
    p->d_cell[k] = nullptr;
    p->d_x[k] = p->d_ux[k] = p->d_uy[k] = p->d_uz[k] = p->d_w[k] = nullptr;  // freeing a set twice must be harmless
}

// Enlarges the population's storage (HBM is plentiful: 288 GB): new arrays, device-to-device copy of
// the slots in use, old arrays freed.  Synchronises the stream.
int grow(opal_pop_t *p, size_t new_cap)
{
    if (new_cap >= (size_t)1 << 31) {
        set_error("population would exceed 2^31 slots per subdomain");
        return OPAL_ERR_CAPACITY;
    }
    OPAL_HIP(hipStreamSynchronize(p->stream));
    size_t used = p->n_upper, old_cap = p->cap;
    auto regrow = [&](auto *&ptr) -> int {
        using T = std::remove_pointer_t<std::remove_reference_t<decltype(ptr)>>;
        T *fresh = nullptr;
        OPAL_HIP(hipMalloc(&fresh, sizeof(T) * new_cap));
        if (ptr && used) OPAL_HIP(hipMemcpy(fresh, ptr, sizeof(T) * std::min(used, old_cap), hipMemcpyDeviceToDevice));
        if (ptr) hipFree(ptr);
        ptr = fresh;
        return OPAL_OK;
    };
    int k = p->cur;
    if (int rc = regrow(p->d_cell[k])) return rc;
    if (int rc = regrow(p->d_x[k])) return rc;
    if (int rc = regrow(p->d_ux[k])) return rc;
    if (int rc = regrow(p->d_uy[k])) return rc;
    if (int rc = regrow(p->d_uz[k])) return rc;
    if (int rc = regrow(p->d_w[k])) return rc;
    if (p->d_prev_x)
        if (int rc = regrow(p->d_prev_x)) return rc;
    if (p->have_alt) {  // the spare set holds nothing between re-binnings
        free_set(p, k ^ 1);
        p->have_alt = false;
    }
    for (int q = 0; q < NEX; q++) {
        if (p->d_ex[q])
            if (int rc = regrow(p->d_ex[q])) return rc;
        if (p->d_ex_alt[q]) {
            hipFree(p->d_ex_alt[q]);
            p->d_ex_alt[q] = nullptr;
        }
    }
    p->cap = new_cap;
    p->emit_list_valid = false;  // the list is sized with the arrays (run_push re-allocates it)
    return OPAL_OK;
}

// T::create's values for the extras of slots [first, first + n): chi = work = y = z = 0,
// tau = +inf (electron.rs:98-99), id = id_base + slot
__global__ void k_reset_extras(double *chi, double *work, double *y, double *z, double *tau, double *id, double *tau_abs,
                               double *tau_st, double *birth, size_t first, size_t n, unsigned long long id_base)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i += first;
    if (chi) chi[i] = 0.0;
    if (work) work[i] = 0.0;
    if (y) y[i] = 0.0;
    if (z) z[i] = 0.0;
    if (tau) tau[i] = INFINITY;
    if (id) reinterpret_cast<unsigned long long *>(id)[i] = id_base + i;
    if (tau_abs) tau_abs[i] = INFINITY;  // photon.rs:113-115
    if (tau_st) tau_st[i] = INFINITY;
    if (birth) birth[i] = -INFINITY;
}

int reset_extras(opal_pop_t *p, size_t first, size_t n)
{
    if (!n) return OPAL_OK;
    k_reset_extras<<<(unsigned)((n + 255) / 256), 256, 0, p->stream>>>(p->d_ex[EX_CHI], p->d_ex[EX_WORK], p->d_ex[EX_Y],
                                                                       p->d_ex[EX_Z], p->d_ex[EX_TAU], p->d_ex[EX_ID],
                                                                       p->d_ex[EX_TAU_ABS], p->d_ex[EX_TAU_ST], p->d_ex[EX_BIRTH],
                                                                       first, n, p->id_base);
    OPAL_HIP(hipGetLastError());
    return OPAL_OK;
}

int ensure_prev(opal_pop_t *p)
{
    if (!p->d_prev_x) OPAL_HIP(hipMalloc(&p->d_prev_x, sizeof(double) * p->cap));
    return OPAL_OK;
}

double species_alpha(const opal_pop_t *p, double dt)
{
    if (p->species == OPAL_ELECTRON) return ELECTRON_CHARGE * dt / (2.0 * ELECTRON_MASS * SPEED_OF_LIGHT);  // electron.rs:276
    if (p->species == OPAL_PHOTON) return 0.0;
    double q = p->z_star * ELEMENTARY_CHARGE, M = p->a * PROTON_MASS;                                        // ion.rs:173-175
    return q * dt / (2.0 * M * SPEED_OF_LIGHT);
}
double species_charge(const opal_pop_t *p)
{
    if (p->species == OPAL_PHOTON) return 0.0;  // photon.rs:199-201
    return p->species == OPAL_ELECTRON ? ELECTRON_CHARGE : p->z_star * ELEMENTARY_CHARGE;  // electron.rs:141, ion.rs:116
}

template <int SPECIES, int MODE>
void launch_push(const PushArgs &a, unsigned nblk, int extras, hipStream_t s)
{
    // SENDX: leavers note their slot for k_pack_extras (links that carry the extras); a separate variant
    // because the four registers it costs would take the plain kernel from 5 to 4 waves per SIMD
    if (a.send_nextra && MODE != MODE_DEPOSIT) {
        if (SPECIES != OPAL_ION && extras == 2)
            k_push<SPECIES, MODE, true, (SPECIES == OPAL_ELECTRON ? 2 : 1), true><<<nblk, BLOCK, 0, s>>>(a);
        else if (SPECIES != OPAL_ION && extras)
            k_push<SPECIES, MODE, true, 1, true><<<nblk, BLOCK, 0, s>>>(a);
        else
            k_push<SPECIES, MODE, true, 0, true><<<nblk, BLOCK, 0, s>>>(a);
        return;
    }
    if (SPECIES != OPAL_ION && extras == 2 && MODE != MODE_DEPOSIT)
        k_push<SPECIES, MODE, true, (SPECIES == OPAL_ELECTRON ? 2 : 1), false><<<nblk, BLOCK, 0, s>>>(a);
    else if (SPECIES != OPAL_ION && extras && MODE != MODE_DEPOSIT)
        k_push<SPECIES, MODE, true, 1, false><<<nblk, BLOCK, 0, s>>>(a);
    else
        k_push<SPECIES, MODE, true, 0, false><<<nblk, BLOCK, 0, s>>>(a);
}

int build_args(opal_pop_t *p, opal_link_t *link, opal_grid_t *g, double dt, int mode, PushArgs &a, unsigned &nblk)
{
    memset(&a, 0, sizeof a);
    if (p->upper_pending)  // (a counter snapshot on its way: the slot count a re-binning left, for one)
        if (int rc = opal::pop_tighten_upper(p)) return rc;
    int k = p->cur;
    a.cell = p->d_cell[k]; a.x = p->d_x[k]; a.ux = p->d_ux[k]; a.uy = p->d_uy[k]; a.uz = p->d_uz[k]; a.w = p->d_w[k];
    a.prev_x = (mode == MODE_ADVANCE && !p->keep_prev) ? nullptr : p->d_prev_x;
    for (int q = 0; q < NEX; q++) a.ex[q] = p->extras ? p->d_ex[q] : nullptr;
    a.eb = g->d_eb[g->cur];
    a.jr = g->j();
    a.lenp = g->lenp;
    a.ppt = p->ppt;
    a.size = (int)g->size; a.L = g->L; a.len = (int)g->len;
    a.out_shift = p->reframe ? p->reframe_shift : 0;
    a.out_size = p->reframe ? p->reframe_size : a.size;
    a.dt = dt;
    a.dt_over_dx = dt / g->dx;
    a.inv_dt = 1.0 / dt;
    a.inv_dx = 1.0 / g->dx;
    a.alpha = species_alpha(p, dt);
    a.charge = species_charge(p);
    auto edge = [&](int bdy, int neighbour) {
        if (bdy != OPAL_BDY_INTERNAL) return (int)EDGE_DELETE;  // yee.rs:329-343 -> None -> leavers dropped
        return (neighbour == g->id && !g->self_link) ? (int)EDGE_WRAP : (int)EDGE_SEND;
    };
    a.left_mode = edge(g->left_bdy, g->rank_to_left);
    a.right_mode = edge(g->right_bdy, g->rank_to_right);
    a.ctr = p->d_ctr;
    if (mode != MODE_DEPOSIT && (a.left_mode == EDGE_SEND || a.right_mode == EDGE_SEND)) {
        if (!link) {
            set_error("%d subdomains: particle migration needs a link", g->ngrids);
            return OPAL_ERR_ARG;
        }
        int slot = p->species;  // electrons 0, ions 1, photons 2
        if (slot >= link->nslots) {
            set_error("this link's messages have no room for species %d (see opal_link_set_layout)", slot);
            return OPAL_ERR_LINK;
        }
        for (int side = 0; side < 2; side++) {
            a.send_msg[side] = link_send(link, side);
            a.send_cap[side] = link->send_cap[side];
        }
        a.send_layout = link->layout;
        a.send_slot = slot;
        a.send_nextra = link->nfields > 6 ? (unsigned)(link->nfields - 6) : 0u;
        // the particles next to a linked edge are counted for transports whose message sizes the host plans, and for
        // peer links, where the same bound sizes the room the populations reserve for arrivals (link_plan)
        a.zone = link_is_peer(link) ? LINK_PEER_ZONE : (link->bounded && link->exchange2) ? LINK_ZONE : 0;
        if (link->zone_scale > 1) a.zone *= (int)link->zone_scale;  // (cuts that drift: link.hpp)
        a.zone_word = link_h_zone(link->xseq) + slot;  // (the exchange this push packs for is number xseq)
    }
    size_t per_block = (size_t)BLOCK * p->ppt;
    a.w_uniform = g_elision && p->w_uniform && p->species != OPAL_PHOTON;
    a.w0 = p->w0;
    a.es_flag = g->d_esflag;
    a.dbg = p->d_dbg;
    a.tab_slot = p->d_tab_slot;
    a.n_tab = p->n_tab;
    // a population re-binned again within ROTATE_BELOW steps is hot: its chunks' first slots are not a fair sample (push_body)
    a.rotate = p->last_interval > 0 && p->last_interval < ROTATE_BELOW;
    const size_t behind = p->n_upper > p->n_sorted ? p->n_upper - p->n_sorted : 0;  // appended since the re-binning
    // arrivals from both neighbours behind a sorted population: two workgroups per chunk of them (push_body)
    // (safe while no arrival can have reached the middle of the subdomain: less than a cell per step since the last
    // re-binning, see push_body)
    // (and only where arrivals are all that is ever appended: photons are also born in place, anywhere)
    // (and not in the step in which the cuts move: the two workgroups of a chunk tell their particles apart by the half of the
    // subdomain the STORED cell lies in, and a cell already stored in the new frame may read as the other half's -- or as
    // outside the subdomain the launch goes by: round 5, six ranks, a false range error and the risk of a second push)
    a.tail_sides = (OPAL_TWO_SIDED_TAIL && !p->reframe && p->species != OPAL_PHOTON && p->n_tab && link && g->size >= 8 * TILE &&
                    (size_t)p->steps_since_sort + 2 * TILE < g->size / 2 && a.left_mode == EDGE_SEND && a.right_mode == EDGE_SEND && mode != MODE_DEPOSIT) ? 2 : 1;
    nblk = p->n_tab + (unsigned)((behind + per_block - 1) / per_block) * (unsigned)a.tail_sides;
    return OPAL_OK;
}

int ensure_emit_list(opal_pop_t *p)
{
    if (p->emit_list_cap >= p->cap && p->d_emit_list) return OPAL_OK;
    if (p->d_emit_list) hipFree(p->d_emit_list);
    p->d_emit_list = nullptr;
    OPAL_HIP(hipMalloc(&p->d_emit_list, sizeof(unsigned) * p->cap));
    p->emit_list_cap = p->cap;
    return OPAL_OK;
}

int run_push(opal_pop_t *p, opal_link_t *link, opal_grid_t *g, double dt, int mode)
{
    if (p->n_upper == 0) return OPAL_OK;
    PushArgs a;
    unsigned nblk;
    if (int rc = build_args(p, link, g, dt, mode, a, nblk)) return rc;
    if (p->n_upper == 0 || nblk == 0) return OPAL_OK;  // (a counter snapshot that landed in build_args: every slot has been vacated)
    if (mode != MODE_DEPOSIT) p->sorted_exact = false;
    if (mode != MODE_DEPOSIT) p->cold_yz = false;  // only the paired electrostatic launch keeps that promise
    if (p->species == OPAL_ELECTRON && p->extras == 2 && p->d_ex[EX_TAU] && mode != MODE_DEPOSIT) {
        // the push compacts this step's emitters as it goes (see push_body): start an empty list
        if (int rc = ensure_emit_list(p)) return rc;
        // (the driver's steps have the field solve's launch clear the count once the emission has consumed it: PublishBatch::zero)
        if (!p->n_emit_clean) OPAL_HIP(hipMemsetAsync(&p->d_ctr->n_emit, 0, sizeof(unsigned), p->stream));
        p->n_emit_clean = false;
        a.emit_list = p->d_emit_list;
        p->emit_list_valid = true;
    }
    int kid = mode == MODE_FUSED ? KERNEL_FUSED : (mode == MODE_ADVANCE ? KERNEL_ADVANCE : KERNEL_DEPOSIT);
    if (p->species == OPAL_PHOTON) kid = KERNEL_PHOTON;
    prof_begin(kid, p->stream);
    if (mode != MODE_DEPOSIT && p->species != OPAL_PHOTON)
        prof_form(FORM_SINGLE | FORM_RHO | (a.w_uniform ? (p->species == OPAL_ELECTRON ? FORM_W_ARG_E : FORM_W_ARG_I) : 0u));
    // which variant of the kernel: 2 runs down the optical depth, 1 touches chi / work / y / z, 0 neither
    int ex = !p->extras ? 0 : p->d_ex[EX_TAU] ? 2 : (p->d_ex[EX_CHI] || p->d_ex[EX_WORK] || p->d_ex[EX_Y]) ? 1 : 0;
    if (p->species == OPAL_ELECTRON) {
        if (mode == MODE_FUSED) launch_push<OPAL_ELECTRON, MODE_FUSED>(a, nblk, ex, p->stream);
        else if (mode == MODE_ADVANCE) launch_push<OPAL_ELECTRON, MODE_ADVANCE>(a, nblk, ex, p->stream);
        else launch_push<OPAL_ELECTRON, MODE_DEPOSIT>(a, nblk, ex, p->stream);
    } else if (p->species == OPAL_PHOTON) {
        // photons carry no charge: they are pushed (main.rs:242-244) but never deposited (main.rs:254-258)
        if (mode != MODE_ADVANCE) {
            set_error("photons are not deposited (photon.rs:199-201: charge 0)");
            return OPAL_ERR_ARG;
        }
        if (ex == 0) {
            // nothing tracked that needs the fields (chi, y, z): the field-free push
            const unsigned nb = (unsigned)((p->n_upper + (size_t)BLOCK * PHOTON_PPT - 1) / ((size_t)BLOCK * PHOTON_PPT));
            if (a.send_nextra) k_push_photon<true><<<nb, BLOCK, 0, p->stream>>>(a);
            else k_push_photon<false><<<nb, BLOCK, 0, p->stream>>>(a);
        } else {
            launch_push<OPAL_PHOTON, MODE_ADVANCE>(a, nblk, ex, p->stream);
        }
    } else {
        if (mode == MODE_FUSED) launch_push<OPAL_ION, MODE_FUSED>(a, nblk, 0, p->stream);
        else if (mode == MODE_ADVANCE) launch_push<OPAL_ION, MODE_ADVANCE>(a, nblk, 0, p->stream);
        else launch_push<OPAL_ION, MODE_DEPOSIT>(a, nblk, 0, p->stream);
    }
    prof_end(kid, p->stream);
    OPAL_HIP(hipGetLastError());
    return OPAL_OK;
}

enum { PART_ALL = 0, PART_EDGES = 1, PART_BULK = 2 };

bool pair_launch_possible(const opal_pop_t *pe, const opal_pop_t *pi, const opal_link_t *link)
{
    return pe->n_upper && pi->n_upper && !pe->extras && pe->stream == pi->stream && !(link && link->nfields > 6);
}

// Decides, for the push about to run, which chunks of `p` can hold a particle that leaves the subdomain or
// deposits into the cells the halo message carries.  Slots [0, n_cell_end's total) are where the last
// re-binning (s = steps_since_sort pushes ago) put them; a particle moves less than a cell per push, so one
// that was W = s + 7 or more cells from both edges is still 7 or more away now: it cannot leave, and its
// stencil (3 cells) stays clear of the 4 interior cells that travel with the halo.  Everything appended since
// (arrivals) counts as edge.  Returns false when there is nothing to gain (or no ordering information).
bool pop_plan_split(opal_pop_t *p, const opal_grid_t *g)
{
    p->split_lo = p->split_hi = 0;
    const size_t size = g->size;
    if (!p->h_cell_end || p->n_cell_end != size) return false;
    const size_t W = (size_t)p->steps_since_sort + 7;
    if (2 * W + 16 >= size) return false;
    const size_t left_end = p->h_cell_end[W - 1];             // slots of cells [0, W)
    const size_t right_begin = p->h_cell_end[size - W - 1];   // slots of cells [size - W, size) start here
    const size_t per_block = (size_t)BLOCK * p->ppt;
    const size_t behind = p->n_upper > p->n_sorted ? p->n_upper - p->n_sorted : 0;
    const size_t all = p->n_tab + (behind + per_block - 1) / per_block;
    size_t lo, hi;
    if (p->n_tab && p->h_tab_slot) {
        // chunk i holds slots [tab[i], tab[i+1]): the chunks that start below left_end are edge chunks, and so is
        // every chunk from the one that holds slot right_begin on (the uniform chunks behind the table included)
        const unsigned *tab = p->h_tab_slot;
        lo = std::lower_bound(tab, tab + p->n_tab, (unsigned)left_end) - tab;
        hi = std::upper_bound(tab, tab + p->n_tab + 1, (unsigned)right_begin) - tab;
        hi = hi ? hi - 1 : 0;
        if (hi > p->n_tab) hi = p->n_tab;
    } else {
        lo = (left_end + per_block - 1) / per_block;
        hi = right_begin / per_block;
    }
    if (lo >= hi || hi > all || (hi - lo) * 2 < all) return false;  // the bulk must be worth a launch of its own
    p->split_lo = (unsigned)lo;
    p->split_hi = (unsigned)hi;
    return true;
}

// electrons and ions of one step in a single launch (fused pass only); `part` selects the edge or the bulk
// chunks of a step that pop_plan_split has split for BOTH species
int run_push_pair(opal_pop_t *pe, opal_pop_t *pi, opal_link_t *link, opal_grid_t *g, double dt, int part = PART_ALL,
                  int mode = MODE_FUSED, hipStream_t on = nullptr)
{
    const hipStream_t st = on ? on : pe->stream;  // (the edge chunks of a split step run beside the bulk, see sim.cpp)
    if (!pair_launch_possible(pe, pi, link)) {
        if (part != PART_ALL) {
            set_error("run_push_pair: a split step needs the paired launch");
            return OPAL_ERR_ARG;
        }
        if (int rc = run_push(pe, link, g, dt, mode)) return rc;
        return run_push(pi, link, g, dt, mode);
    }
    PushArgs ae, ai;
    unsigned ne, ni;
    if (mode != MODE_DEPOSIT) pe->sorted_exact = pi->sorted_exact = false;
    if (int rc = build_args(pe, link, g, dt, mode, ae, ne)) return rc;
    if (int rc = build_args(pi, link, g, dt, mode, ai, ni)) return rc;
    PairPlan plan = {0, ne, 0, 0, 0, ni, 0, 0};
    if (part == PART_EDGES) plan = {0, pe->split_lo, pe->split_hi, ne - pe->split_hi, 0, pi->split_lo, pi->split_hi, ni - pi->split_hi};
    if (part == PART_BULK) plan = {pe->split_lo, pe->split_hi - pe->split_lo, 0, 0, pi->split_lo, pi->split_hi - pi->split_lo, 0, 0};
    unsigned total = plan.en0 + plan.en1 + plan.in0 + plan.in1;
    if (!total) return OPAL_OK;
    int kid = mode == MODE_FUSED ? KERNEL_FUSED : (mode == MODE_ADVANCE ? KERNEL_ADVANCE : KERNEL_DEPOSIT);
    prof_begin(kid, st);
    // nothing transverse as far as the host has heard: the electrostatic form (which checks, see push_body)
    const bool es = g_elision && mode == MODE_FUSED && pe->cold_yz && pi->cold_yz && *static_cast<volatile int *>(g->h_esflag) == 0;
    if (!es && mode != MODE_DEPOSIT) pe->cold_yz = pi->cold_yz = false;  // a general push may leave uy, uz behind
    const bool rho = g->want_rho || !g_elision;  // (opal_sim_step: only the last step's rho can be observed)
    if (mode != MODE_DEPOSIT)
        prof_form((es ? FORM_ES : 0u) | ((rho || mode != MODE_FUSED) ? FORM_RHO : 0u) | (ae.w_uniform ? FORM_W_ARG_E : 0u) |
                  (ai.w_uniform ? FORM_W_ARG_I : 0u));
    if (es && rho) k_push_pair<MODE_FUSED, true, true><<<total, BLOCK, 0, st>>>(ae, ai, plan);
    else if (es) k_push_pair<MODE_FUSED, false, true><<<total, BLOCK, 0, st>>>(ae, ai, plan);
    else if (mode == MODE_FUSED && rho) k_push_pair<MODE_FUSED, true><<<total, BLOCK, 0, st>>>(ae, ai, plan);
    else if (mode == MODE_FUSED) k_push_pair<MODE_FUSED, false><<<total, BLOCK, 0, st>>>(ae, ai, plan);
    else if (mode == MODE_ADVANCE) k_push_pair<MODE_ADVANCE, true><<<total, BLOCK, 0, st>>>(ae, ai, plan);
    else k_push_pair<MODE_DEPOSIT, true><<<total, BLOCK, 0, st>>>(ae, ai, plan);
    prof_end(kid, st);
    OPAL_HIP(hipGetLastError());
    return OPAL_OK;
}

int do_sort(opal_pop_t *p, const opal_grid_t *g, bool wait = true);

// room for a step's arrivals and the kernel arguments that append them to one population
int prepare_unpack(opal_pop_t *p, opal_link_t *link, const opal_grid_t *g, UnpackArgs &u)
{
    bool from_left = g->left_bdy == OPAL_BDY_INTERNAL, from_right = g->right_bdy == OPAL_BDY_INTERNAL;
    // at most what can cross each edge in this step per species (link_plan): the host's bound on the slots in use grows
    // by that much, not by the link's full capacity
    size_t incoming = (from_left ? (size_t)link->plan_cap[0] : 0) + (from_right ? (size_t)link->plan_cap[1] : 0);
    // the host only knows that at most `incoming` particles arrive per step; counter snapshots that land a
    // step or two later pull its bound on the slots in use back down to the true count
    if (int rc = opal::pop_tighten_upper(p)) return rc;
    if (p->n_upper + incoming > p->cap) {
        // The host's bound has run into the capacity.  It is a bound: `incoming` is what CAN cross (at a pile-up hundreds
        // of times what does), added every step and taken back by snapshots that land steps later.  The true slot count
        // first -- one small synchronous copy; round 3 re-binned the whole population here, 523 times in 2000 steps of the
        // 4-subdomain laser-plasma rehearsal once the links' bound followed the pile-up -- and, if that is not enough,
        // room for several steps' worth of the bound (HBM is plentiful; vacated slots are the re-binning policy's business).
        unsigned n_true = 0;
        OPAL_HIP(hipMemcpyAsync(&n_true, &p->d_ctr->n_slots, sizeof n_true, hipMemcpyDeviceToHost, p->stream));
        OPAL_HIP(hipStreamSynchronize(p->stream));
        if (int rc = opal::pop_tighten_upper(p)) return rc;  // (a snapshot in flight has landed: the stream is drained)
        p->n_upper = n_true;
        p->added_since_snapshot = 0;
        if (p->n_upper + incoming > p->cap) {
            size_t want = p->n_upper + 6 * incoming + p->n_upper / 4;
            if (int rc = grow(p, want)) return rc;
        }
    }
    int k = p->cur;
    u.recv[0] = from_left ? link_recv(link, 0) : nullptr;
    u.recv[1] = from_right ? link_recv(link, 1) : nullptr;
    u.slot = p->species;
    if (u.slot >= link->nslots) {
        set_error("this link's messages have no room for species %d (see opal_link_set_layout)", u.slot);
        return OPAL_ERR_LINK;
    }
    u.layout = link->layout;
    u.recv_cap[0] = link->send_cap[0];  // an edge's two directions ship the same number of slots
    u.recv_cap[1] = link->send_cap[1];
    u.reserve[0] = link->plan_cap[0];
    u.reserve[1] = link->plan_cap[1];
    u.cell = p->d_cell[k];
    u.f[0] = p->d_x[k]; u.f[1] = p->d_ux[k]; u.f[2] = p->d_uy[k]; u.f[3] = p->d_uz[k]; u.f[4] = p->d_w[k];
    u.prev_x = p->prev_valid ? p->d_prev_x : nullptr;
    for (int q = 0; q < NEX; q++) u.ex[q] = p->extras ? p->d_ex[q] : nullptr;
    u.nextra = link->nfields > 6 ? (unsigned)(link->nfields - 6) : 0u;
    u.ctr = p->d_ctr;
    u.pop_cap = (unsigned)p->cap;
    u.w_uniform = p->w_uniform;
    u.w0 = p->w0;
    u.es_flag = g->d_esflag;
    u.my_size = p->reframe ? p->reframe_size : (int)g->size;  // (arrivals come in the frame of the subdomain as it is after this step)
    u.emit_list = (p->emit_list_valid && p->emit_list_cap >= p->cap) ? p->d_emit_list : nullptr;
    if (!u.emit_list) p->emit_list_valid = false;  // arrivals the list does not know about
    p->sorted_exact = false;
    p->n_upper += incoming;
    p->added_since_snapshot += incoming;
    p->added_since_pub += incoming;
    return OPAL_OK;
}

// appends what the neighbours sent to the populations, one launch (k_link_tail); with zero_counts the send messages'
// leaver counts are reset for the next step (instead of link_begin).
//   sync          YeeGrid::synchronize runs in the same launch (its own workgroup)
//   adv           ... followed there by the field solve (small grids)
//   peer_exchange the launch makes the exchange itself over the link's peer-mapped mailboxes (the messages are NOT in
//                 the recv buffers yet: nothing has been called since the push); otherwise they have been moved already
//   pub           counters the re-binning policy wants published by this launch
int unpack_arrivals(opal_pop_t *const *pops, int npops, opal_link_t *link, const opal_grid_t *g, bool zero_counts,
                    const SyncArgs *sync = nullptr, const AdvArgs *adv = nullptr, bool peer_exchange = false,
                    const PublishBatch *pub = nullptr)
{
    if (g->left_bdy != OPAL_BDY_INTERNAL && g->right_bdy != OPAL_BDY_INTERNAL) {
        if (sync) {
            set_error("unpack_arrivals: a synchronize to run, but no neighbour to unpack from");
            return OPAL_ERR_ARG;
        }
        return OPAL_OK;
    }
    UnpackBatch b;
    memset(&b, 0, sizeof b);
    if (sync) {
        b.has_sync = 1;
        b.sync = *sync;
    }
    if (adv && sync && adv->edge_left <= ADV_EDGE && adv->edge_right <= ADV_EDGE) {
        b.has_adv = 1;
        b.adv = *adv;
    }
    if (pub) b.pub = *pub;
    const bool nb[2] = {g->left_bdy == OPAL_BDY_INTERNAL, g->right_bdy == OPAL_BDY_INTERNAL};
    hipStream_t st = nullptr;
    for (int i = 0; i < npops && b.n < 3; i++) {
        if (!pops[i]) continue;
        if (int rc = prepare_unpack(pops[i], link, g, b.a[b.n])) return rc;
        st = pops[i]->stream;
        b.n++;
    }
    if (!b.n) st = link->stream;  // nobody to unpack for: the kernel still files this exchange's bounds (link.cpp)
    if (peer_exchange) {
        HaloSrc hs{g->d_eb[g->cur], g->j(), g->lenp, (int)g->len};
        if (int rc = link_peer_next(link, nb[0], nb[1], false, &hs, b.peer)) return rc;
        link->xseq++;
        for (int side = 0; side < 2; side++) {
            const char *msg = b.peer.box[side] ? b.peer.box[side] + PEER_FLAG_BYTES : nullptr;
            for (int i = 0; i < b.n; i++) b.a[i].recv[side] = msg;
            const double *halo = msg ? reinterpret_cast<const double *>(msg + LINK_HEADER_BYTES) : nullptr;
            if (sync) (side == 0 ? b.sync.recv_left : b.sync.recv_right) = halo;
        }
    }
    if (zero_counts) {
        b.zero[0] = reinterpret_cast<unsigned *>(link_send(link, 0));
        b.zero[1] = reinterpret_cast<unsigned *>(link_send(link, 1));
    }
    // the exchange just made is number xseq - 1; a peer link files every LINK_PEER_EVERY-th one only (link.hpp)
    if (link->d_bounds && link->xseq && !(link_is_peer(link) && (link->xseq - 1) % LINK_PEER_EVERY)) {
        for (int side = 0; side < 2; side++) {
            b.send_head[side] = nb[side] ? reinterpret_cast<const unsigned *>(link_send(link, side)) : nullptr;
            // (a peer exchange made by this very launch leaves the neighbour's message in the mailbox)
            const char *got = peer_exchange ? (b.peer.box[side] ? b.peer.box[side] + PEER_FLAG_BYTES : nullptr) : link_recv(link, side);
            b.recv_head[side] = nb[side] && got ? reinterpret_cast<const unsigned *>(got) : nullptr;
        }
        b.bounds = &link->d_bounds[(link->xseq - 1) % LINK_RING];
        b.seq = link->xseq;
    }
    b.zone_at = link_h_zone(link->xseq ? link->xseq - 1 : 0u);
    // Where many particles stand near a cut (the bound the link is held to: a front, a hot dense plasma) a step's leavers can
    // be megabytes: sending and unpacking are then shared among TAIL_PARTS workgroups per species and side.  Two-stream
    // links (bound ~25 000 slots, ~30 leavers a step) keep the single workgroups.
    static const unsigned parts_at = getenv("OPAL_TAIL_PARTS_AT") ? (unsigned)atoi(getenv("OPAL_TAIL_PARTS_AT")) : 65536u;  // (tests)
    b.parts = (peer_exchange && std::max(link->plan_cap[0], link->plan_cap[1]) >= parts_at) ? TAIL_PARTS : 1;
    if (peer_exchange && link->burst_parts) b.parts = (int)link->burst_parts;
    b.first_slot = b.parts > 1 ? link->d_tail_words : nullptr;
    const unsigned nblk = (b.n ? 2 * b.n * b.parts : 1) + b.has_sync;
    if (b.has_adv) k_link_tail<ADV_TILE, true><<<nblk + grid_interior_tiles(b.adv), ADV_TILE, 0, st>>>(b);
    else k_link_tail<256, false><<<nblk, 256, 0, st>>>(b);
    OPAL_HIP(hipGetLastError());
    if (zero_counts) link->counts_clean = true;
    // (only once the reservations since the last snapshot add up to a few per cent of the population: a
    // snapshot is a small copy on the stream, not worth having every step)
    // (Not at all for a population under the adaptive re-binning policy: the counters that policy has published every other
    // step -- by the very kernel launched above, into mapped host memory -- include the slot count, and policy_step takes the
    // bound down with it.  The copy, its event and the packets round them cost the linked step ~2 us a step: round 4's trace.)
    for (int i = 0; i < npops; i++)
        if (pops[i] && pops[i]->added_since_snapshot > pops[i]->n_upper / 32 &&
            !(pops[i]->sort_interval < 0 && pops[i]->species != OPAL_PHOTON))
            if (int rc = opal::pop_snapshot_upper(pops[i], 0)) return rc;
    return OPAL_OK;
}

bool needs_link(const opal_grid_t *g)
{
    return (g->ngrids > 1 || g->self_link) && (g->left_bdy == OPAL_BDY_INTERNAL || g->right_bdy == OPAL_BDY_INTERNAL);
}

int pack_leaver_extras(opal_pop_t *p, opal_link_t *link, const opal_grid_t *g)
{
    if (link->nfields <= 6 || p->n_upper == 0) return OPAL_OK;
    PackExtrasArgs a;
    memset(&a, 0, sizeof a);
    a.send[0] = g->left_bdy == OPAL_BDY_INTERNAL ? link_send(link, 0) : nullptr;
    a.send[1] = g->right_bdy == OPAL_BDY_INTERNAL ? link_send(link, 1) : nullptr;
    a.slot = p->species;
    a.layout = link->layout;
    a.send_cap[0] = link->send_cap[0];
    a.send_cap[1] = link->send_cap[1];
    a.nextra = (unsigned)(link->nfields - 6);
    for (int q = 0; q < NEX; q++) a.ex[q] = p->extras ? p->d_ex[q] : nullptr;
    k_pack_extras<<<2, 256, 0, p->stream>>>(a);
    OPAL_HIP(hipGetLastError());
    return OPAL_OK;
}

// Population::advance's exchange (mod.rs:303-354) for one species on its own: push, then one
// whole-message exchange, then append the arrivals.
int advance_with_exchange(opal_pop_t *p, opal_link_t *link, opal_grid_t *g, double dt, int mode)
{
    if (!needs_link(g)) return run_push(p, link, g, dt, mode);
    if (!link) {
        set_error("%d subdomains: particle migration needs a link", g->ngrids);
        return OPAL_ERR_ARG;
    }
    if (int rc = link_begin(link, p->stream, 1u << p->species)) return rc;
    if (int rc = run_push(p, link, g, dt, mode)) return rc;
    if (int rc = pack_leaver_extras(p, link, g)) return rc;
    if (int rc = link_exchange_particles(link, g->left_bdy == OPAL_BDY_INTERNAL, g->right_bdy == OPAL_BDY_INTERNAL, p->stream)) return rc;
    return unpack_arrivals(&p, 1, link, g, true);
}

// Re-binning: counting sort by cell into the spare set of arrays, the chunk table rebuilt.
// wait = true: the host waits for the device and comes back knowing the slot count, the cell map and the chunk table
// (callers that need them: explicit opal_pop_sort, compaction for room, the overlapped step's edge / bulk split).
// wait = false (the policy's re-binnings): nothing is read back here.  The host keeps its upper bound of the slots in
// use and sizes the chunk table and the launches with it; the pushes take the number of sorted slots from the device
// (ctr->n_sorted); the true count arrives with a counter snapshot a step or two later (pop_tighten_upper).  A
// re-binning then costs its kernels and no bubble (measured in the laser-plasma interaction: 0.1-0.2 ms of idle device
// per re-binning, 140 re-binnings per 2000 steps).
int do_sort(opal_pop_t *p, const opal_grid_t *g, bool wait)
{
    if (p->n_upper == 0) return OPAL_OK;
    if (p->want_host_maps) wait = true;
    if (!p->have_alt) {
        if (int rc = alloc_set(p, p->cur ^ 1)) return rc;
        p->have_alt = true;
    }
    size_t ncells = g->size;
    if (p->count_cap < ncells + 2) {
        if (p->d_count) hipFree(p->d_count);
        OPAL_HIP(hipMalloc(&p->d_count, sizeof(unsigned) * (ncells + 2)));
        p->count_cap = ncells + 2;
    }
    double *d_prev_alt = nullptr;
    if (p->d_prev_x && p->prev_valid) OPAL_HIP(hipMalloc(&d_prev_alt, sizeof(double) * p->cap));
    prof_begin(KERNEL_SORT, p->stream);
    OPAL_HIP(hipMemsetAsync(p->d_count, 0, sizeof(unsigned) * (ncells + 2), p->stream));
    unsigned nblk = (unsigned)((p->n_upper + 256 * SORT_PPT - 1) / (256 * SORT_PPT));
    int k = p->cur, o = k ^ 1;
    k_sort_hist<<<nblk, 256, 0, p->stream>>>(p->d_cell[k], p->d_ctr, p->d_count);
    k_sort_scan<<<1, 1024, 0, p->stream>>>(p->d_count, (int)ncells, p->d_ctr);
    SortArgs sa;
    sa.cell = p->d_cell[k];
    sa.f[0] = p->d_x[k]; sa.f[1] = p->d_ux[k]; sa.f[2] = p->d_uy[k]; sa.f[3] = p->d_uz[k]; sa.f[4] = p->d_w[k];
    sa.ocell = p->d_cell[o];
    sa.of[0] = p->d_x[o]; sa.of[1] = p->d_ux[o]; sa.of[2] = p->d_uy[o]; sa.of[3] = p->d_uz[o]; sa.of[4] = p->d_w[o];
    sa.prev = d_prev_alt ? p->d_prev_x : nullptr;
    sa.oprev = d_prev_alt;
    for (int q = 0; q < NEX; q++) {
        sa.ex[q] = nullptr;
        sa.oex[q] = nullptr;
        if (p->extras && p->d_ex[q]) {
            if (!p->d_ex_alt[q]) OPAL_HIP(hipMalloc(&p->d_ex_alt[q], sizeof(double) * p->cap));
            sa.ex[q] = p->d_ex[q];
            sa.oex[q] = p->d_ex_alt[q];
        }
    }
    static const bool staged = !(getenv("OPAL_SORT_STAGED") && atoi(getenv("OPAL_SORT_STAGED")) == 0);  // (0: the A/B's other leg)
    if (staged) k_sort_scatter_staged<<<nblk, 256, 0, p->stream>>>(sa, p->d_ctr, p->d_count);
    else k_sort_scatter<<<nblk, 256, 0, p->stream>>>(sa, p->d_ctr, p->d_count);
    k_sort_finish<<<1, 1, 0, p->stream>>>(p->d_ctr);
    prof_end(KERNEL_SORT, p->stream);
    OPAL_HIP(hipGetLastError());
    if (wait)
        if (int rc = fetch_counters(p)) return rc;
    if (d_prev_alt) {
        hipFree(p->d_prev_x);
        p->d_prev_x = d_prev_alt;
    }
    for (int q = 0; q < NEX; q++)
        if (sa.oex[q]) std::swap(p->d_ex[q], p->d_ex_alt[q]);
    p->cur = o;
    p->count_cells = (unsigned)ncells;
    if (p->upper_pending) {  // a snapshot from before the compaction: no longer a bound worth having
        hipEventSynchronize(p->upper_event);
        p->upper_pending = false;
    }
    p->sorted_count_pending = false;
    if (wait) {
        p->n_upper = p->h_ctr->n_slots;
        // after the scatter the cursor of cell c stands one past its last particle: the host keeps that map
        if (p->n_cell_end != ncells) {
            free(p->h_cell_end);
            p->h_cell_end = (unsigned *)malloc(sizeof(unsigned) * ncells);
            p->n_cell_end = p->h_cell_end ? ncells : 0;
        }
        if (p->h_cell_end) OPAL_HIP(hipMemcpy(p->h_cell_end, p->d_count, sizeof(unsigned) * ncells, hipMemcpyDeviceToHost));
    } else {
        p->n_cell_end = 0;  // the host's cell map is of the previous order
    }
    // the next interval: end it ahead of the misses if the last one met a wall of them (policy_step)
    p->preempt_at = 0;
    if (p->sort_interval < 0 && p->last_onset > 0 && (p->last_trigger - p->last_onset) * 3 <= p->last_trigger)
        p->preempt_at = std::max(4, p->last_onset - 1);
    p->miss_onset = 0;
    p->last_onset = p->last_trigger = 0;
    p->last_interval = p->steps_since_sort;
    p->steps_since_sort = 0;
    p->emit_list_valid = false;  // slots moved
    p->sorted_exact = true;
    p->dead_cost = 0.0;
    p->ctr_pending = false;  // (a publication still on its way carries an older sequence number and is never read)
    // the chunk table: slot ranges of the push workgroups, cut every BLOCK x pop_max_ppt slots and every CHUNK_CELLS cells
    // (sized with the host's bound of the slot count; the kernel reads the true one: see k_chunk_table)
    {
        const unsigned N = (unsigned)p->n_upper, chunk = BLOCK * pop_max_ppt(p);
        const unsigned K1 = (N + chunk - 1) / chunk, K2 = (unsigned)((ncells + CHUNK_CELLS - 1) / CHUNK_CELLS);
        const unsigned M = N ? K1 + K2 : 0;
        if (M + 1 > p->tab_cap) {
            if (p->d_tab_slot) hipFree(p->d_tab_slot);
            free(p->h_tab_slot);
            p->d_tab_slot = nullptr; p->h_tab_slot = nullptr;
            p->tab_cap = (size_t)M + 1 + M / 4;
            OPAL_HIP(hipMalloc(&p->d_tab_slot, sizeof(unsigned) * p->tab_cap));
            p->h_tab_slot = (unsigned *)malloc(sizeof(unsigned) * p->tab_cap);
        }
        if (M) {
            k_chunk_table<<<(M + 255) / 256, 256, 0, p->stream>>>(p->d_count, (int)ncells, chunk, p->d_ctr, K1, K2, p->d_tab_slot);
            OPAL_HIP(hipGetLastError());
            if (wait) {
                if (p->h_tab_slot)
                    OPAL_HIP(hipMemcpyAsync(p->h_tab_slot, p->d_tab_slot, sizeof(unsigned) * (M + 1), hipMemcpyDeviceToHost, p->stream));
                OPAL_HIP(hipStreamSynchronize(p->stream));
            }
        }
        p->n_tab = M;
        p->n_sorted = wait ? N : 0;  // (a lower bound until the count arrives)
    }
    if (wait) {
        // uniform chunks (slots appended behind the table): at most about a quarter of the tile
        unsigned ne = std::max(1u, p->h_ctr->n_nonempty);
        double per_cell = (double)p->h_ctr->n_live / ne;
        int ppt = (int)((TILE / 4) * per_cell / BLOCK);
        p->ppt = std::max(1, std::min(pop_max_ppt(p), ppt));
    } else {
        p->ppt = std::max(1, std::min(pop_max_ppt(p), p->ppt));
        // the slot count the re-binning left, on its way to the host: it is both the new bound of the slots in use and
        // the number of sorted slots
        if (int rc = opal::pop_snapshot_upper(p, 0)) return rc;
        p->sorted_count_pending = true;
    }
    return OPAL_OK;
}

// Publications asked for while a PublishScope is alive go out together when it ends (one launch for the species of a
// step); without a scope each goes out at once.
struct PublishScope;
thread_local PublishScope *g_publish_scope = nullptr;
struct PublishScope {
    PublishBatch b{};
    hipStream_t stream = nullptr;
    PublishScope *outer = g_publish_scope;  // (scopes may nest: the innermost one collects)
    PublishScope() { g_publish_scope = this; }
    int flush()
    {
        if (b.n) {
            k_publish_counters<<<1, 64, 0, stream>>>(b);
            b.n = 0;
            OPAL_HIP(hipGetLastError());
        }
        return OPAL_OK;
    }
    ~PublishScope()
    {
        flush();
        g_publish_scope = outer;
    }
};

int ask_publication(opal_pop_t *p)
{
    if (p->ctr_pending) return OPAL_OK;
    p->pub_seq++;
    p->ctr_pending = true;
    p->added_since_pub = 0;  // (whatever is reserved from here on may or may not be in the slot count this publication carries)
    PublishBatch one{};
    PublishBatch &b = g_publish_scope ? g_publish_scope->b : one;
    if (g_publish_scope) {
        if (b.n == 4 || (b.n && g_publish_scope->stream != p->stream))
            if (int rc = g_publish_scope->flush()) return rc;
        g_publish_scope->stream = p->stream;
    }
    b.src[b.n] = p->d_ctr;
    b.dst[b.n] = p->d_pub;
    b.seq[b.n] = p->pub_seq;
    b.n++;
    if (!g_publish_scope) {
        k_publish_counters<<<1, 64, 0, p->stream>>>(one);
        OPAL_HIP(hipGetLastError());
    }
    return OPAL_OK;
}

int wait_published(const opal_pop_t *p)
{
    // the kernel that writes it was launched check_every steps ago: normally long done
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    // (acquire: the counters in front of the sequence number are read only after it has been seen)
    while (__atomic_load_n(&p->h_pub->seq, __ATOMIC_ACQUIRE) != p->pub_seq) {
        if ((++spins & 0xffff) == 0) {
            hipError_t q = hipStreamQuery(p->stream);  // surfaces a device fault instead of spinning on it
            if (q != hipSuccess && q != hipErrorNotReady) OPAL_HIP(q);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                set_error("the counters published for the re-binning policy never arrived");
                return OPAL_ERR_HIP;
            }
        }
    }
    return OPAL_OK;
}

// One step of the re-binning policy for `p`: counts the step, reads the counters published a check ago, asks for the
// next publication (into the PublishScope alive around the call, or by a launch of its own) and says whether the
// population should be re-binned now.  The caller re-bins (do_sort) -- after anything it still has to launch for this
// step (the linked step's tail kernel carries the publication and must see the slots where the push left them).
int policy_step_inner(opal_pop_t *p, bool &sort_now);
int policy_step(opal_pop_t *p, bool &sort_now)
{
    const int rc = policy_step_inner(p, sort_now);
    // in the step in which the cuts move the cells already count from the new cut while the grid is the old one (PushArgs:
    // out_shift): the re-binning, which goes by the grid's size, waits until the grid has followed (pops_sort_deferred)
    if (p->reframe) {
        p->sort_deferred = p->sort_deferred || sort_now;
        sort_now = false;
    }
    return rc;
}
int policy_step_inner(opal_pop_t *p, bool &sort_now)
{
    sort_now = false;
    p->steps_since_sort++;
    if (p->sort_interval == 0) return OPAL_OK;
    if (p->sort_interval > 0) {
        sort_now = p->steps_since_sort >= p->sort_interval;
        return OPAL_OK;
    }
    // adaptive ("rent or buy"): every few steps read the tile-miss counters; once the time lost to misses since the
    // last re-binning exceeds what a re-binning costs, re-bin.
#ifndef OPAL_CHECK_EVERY
#define OPAL_CHECK_EVERY 2  // measured: the laser-plasma interaction is 6 % faster at 2 than at 4; two-stream does not care
#endif
    constexpr int check_every = OPAL_CHECK_EVERY;
    // ns of kernel time per tile miss.  Measured by sweeping it: the two-stream bench is flat (misses are rare
    // there); the relativistic laser-plasma interaction runs at 0.494 ms/step at 100, 0.405 at 800 and, with the
    // chunk table, 0.36 at 2000 and 5000 against 0.385 at 800: a miss costs far more than its own global atomics,
    // because the workgroup it sits in finishes late, and misses arrive in bursts (a bunch outruns its tile), so
    // a high price just triggers the re-binning one check earlier
    // Two prices: a wave's pass through the slow path (its workgroup finishes that much later) and each particle in
    // it (15 global atomics that contend with their neighbours').  An isolated miss costs their sum, ~2 us as before;
    // 64 lanes missing together cost one pass and 64 small prices, not 64 passes.
#ifndef OPAL_PASS_NS
#define OPAL_PASS_NS 1900.0
#endif
#ifndef OPAL_MISS_NS
#define OPAL_MISS_NS 100.0
#endif
    constexpr double miss_ns = OPAL_MISS_NS, pass_ns = OPAL_PASS_NS;
    // Ahead of the misses.  Where they come as a wall -- a relativistic bunch outrunning its tiles, a dense ion front:
    // nothing for N steps, then thousands per step within three or four -- waiting for their cost to add up to a
    // re-binning's means paying both, and the counters the host sees are two to four steps old besides (the ion front of
    // the laser-plasma run took the push from 0.26 to 0.67 ms before the re-binning came).  If the previous interval
    // showed such a wall (the misses cost a re-binning within a third of the interval after they began), this interval
    // ends one step before the wall is due; an interval that ends without a miss seen moves the date two steps out.
    if (p->preempt_at > 0 && p->steps_since_sort >= p->preempt_at) {
        if (p->miss_onset == 0) {  // nothing seen yet: the wall may have moved further away -- probe later next time
            p->last_onset = p->preempt_at + 3;
            p->last_trigger = p->last_onset + 1;
        } else {
            p->last_onset = p->miss_onset;
            p->last_trigger = p->steps_since_sort;
        }
        sort_now = true;
        return OPAL_OK;
    }
    if (p->steps_since_sort % check_every != 0) return OPAL_OK;
    // The counters are published to mapped host memory by a one-wave kernel and read at the NEXT check, check_every
    // steps later: waiting for that older publication keeps the host at most ~2 check intervals ahead of the device
    // (so the policy sees recent counts) while the device still has a check interval of queued steps to run and
    // never idles.  No hipMemcpyAsync (a 64-byte copy hands the stream to the copy engine and back) and no event (a
    // recorded event is a barrier packet, ~5 us between two 5 us kernels): the kernel writes a sequence number behind
    // the counters and the host polls it; the populations checked in one step share one launch (PublishScope).
    bool fresh = false;
    if (p->ctr_pending) {
        if (int rc = wait_published(p)) return rc;
        fresh = true;
        p->h_ctr_seen = p->h_pub->c;
        p->ctr_pending = false;
    }
    if (fresh) {
        // the slot count of a check ago, and what has been reserved since that publication was asked for -- for arrivals
        // (prepare_unpack), for the photons an emission or an absorption pass may append (qed.hip, absorb.hip): a bound on
        // the slots in use now (a re-binning in between would have withdrawn the publication: sort_end)
        const size_t bound = (size_t)p->h_ctr_seen.n_slots + p->added_since_pub;
        if (bound < p->n_upper) p->n_upper = bound;
    }
    if (int rc = ask_publication(p)) return rc;
    if (!fresh) return OPAL_OK;
    const PopCounters seen = p->h_ctr_seen;  // snapshot taken when the copy was found complete
    // (a re-binning: ~100 B per particle at 3.6 TB/s -- 0.277 ms per 1e7 measured in round 3, 0.243 with the staged scatter of
    // round 4; prices of 0.018-0.028 ns per slot give the same laser-plasma run -- plus its launches and the host's part)
    double sort_ms = 0.028 * (double)seen.n_slots * 1e-6 + 0.05;
    if (p->species == OPAL_PHOTON && !p->extras) {
        // The field-free photon push has no tile to miss: order does not matter to it.  What a re-binning buys
        // is the slots that photons vacated at the open edges (or by migrating): each is still streamed, 36 B
        // per step, until it is squeezed out.  Rent or buy on that: accumulate the bytes spent on vacated slots
        // since the last re-binning (n_dead is sampled every check_every steps) and re-bin once they exceed
        // what the re-binning itself moves (~100 B per slot).
        p->dead_cost += (double)seen.n_dead * 36.0 * check_every;
        if (p->dead_cost > 100.0 * (double)seen.n_slots && seen.n_dead > 0) {
            p->dead_cost = 0.0;
            sort_now = true;
        }
        return OPAL_OK;
    }
    double lost_ms = ((double)seen.n_slow_pass * pass_ns + (double)seen.n_slow * miss_ns) * 1e-6;
    // (the counters seen were published a check ago, at steps_since_sort - check_every)
    if (seen.n_slow && p->miss_onset == 0) p->miss_onset = std::max(1, p->steps_since_sort - check_every);
    sort_now = lost_ms > sort_ms;
    if (sort_now) {
        p->last_onset = p->miss_onset ? p->miss_onset : p->steps_since_sort;
        p->last_trigger = p->steps_since_sort;
    }
    return OPAL_OK;
}

int maybe_sort(opal_pop_t *p, const opal_grid_t *g)
{
    bool sort_now;
    if (int rc = policy_step(p, sort_now)) return rc;
    return sort_now ? do_sort(p, g, false) : OPAL_OK;
}

}  // namespace

extern "C" {

int opal_pop_create(opal_pop_t **out, int species, size_t capacity, void *hip_stream)
{
    if (!out || (species != OPAL_ELECTRON && species != OPAL_ION && species != OPAL_PHOTON)) {
        set_error("opal_pop_create: bad arguments");
        return OPAL_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("opal_pop_create: no HIP device available (the HIP path has no CPU fallback)");
        return OPAL_ERR_HIP;
    }
    if (capacity >= (size_t)1 << 31) {
        set_error("opal_pop_create: capacity %zu exceeds 2^31 slots per subdomain", capacity);
        return OPAL_ERR_CAPACITY;
    }
    opal_pop_t *p = new opal_pop_t();
    memset((void *)p, 0, sizeof *p);
    p->species = species;
    p->cap = std::max<size_t>(capacity, 1024);
    p->stream = (hipStream_t)hip_stream;
    p->z_star = p->z = p->a = 1.0;  // ion.rs:93-95
    p->sort_interval = -1;
    p->keep_prev = true;
    p->ppt = OPAL_MAX_PPT;
    if (int rc = alloc_set(p, 0)) return rc;
    OPAL_HIP(hipMalloc(&p->d_ctr, sizeof(PopCounters)));
    OPAL_HIP(hipHostMalloc(&p->h_ctr, sizeof(PopCounters)));
    OPAL_HIP(hipHostMalloc(&p->h_pub, sizeof(PubCounters), hipHostMallocMapped | hipHostMallocCoherent));
    memset((void *)p->h_pub, 0, sizeof(PubCounters));
    OPAL_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&p->d_pub), p->h_pub, 0));
    OPAL_HIP(hipMemsetAsync(p->d_ctr, 0, sizeof(PopCounters), p->stream));
    p->partial_cap = 2048;
    OPAL_HIP(hipMalloc(&p->d_partial, sizeof(double) * (p->partial_cap + 8)));
    OPAL_HIP(hipStreamSynchronize(p->stream));
    *out = p;
    return OPAL_OK;
}

int opal_pop_free(opal_pop_t *p)
{
    if (!p) return OPAL_OK;
    free_set(p, 0);
    if (p->have_alt || p->d_cell[1]) free_set(p, 1);
    if (p->d_prev_x) hipFree(p->d_prev_x);
    for (int q = 0; q < NEX; q++) {
        if (p->d_ex[q]) hipFree(p->d_ex[q]);
        if (p->d_ex_alt[q]) hipFree(p->d_ex_alt[q]);
    }
    if (p->d_count) hipFree(p->d_count);
    if (p->d_hist) hipFree(p->d_hist);
    if (p->d_emit_list) hipFree(p->d_emit_list);
    if (p->d_tab_slot) hipFree(p->d_tab_slot);
    if (p->d_dbg) hipFree(p->d_dbg);
    if (p->d_abs_event) hipFree(p->d_abs_event);
    if (p->d_abs_partner) hipFree(p->d_abs_partner);
    if (p->d_abs_counts) hipFree(p->d_abs_counts);
    free(p->h_tab_slot);
    free(p->h_cell_end);
    if (p->upper_event) hipEventDestroy(p->upper_event);
    if (p->h_upper) hipHostFree(p->h_upper);
    hipFree(p->d_partial);
    hipFree(p->d_ctr);
    hipHostFree(p->h_ctr);
    hipHostFree(p->h_pub);
    if (p->fetch_event) hipEventDestroy(p->fetch_event);
    delete p;
    return OPAL_OK;
}

int opal_pop_set_charge_state(opal_pop_t *p, double z_star, double z, double a)
{
    if (int rc = check_pop(p)) return rc;
    if (p->species != OPAL_ION) return OPAL_OK;
    p->z_star = z_star; p->z = z; p->a = a;
    return OPAL_OK;
}

int opal_pop_upload_soa(opal_pop_t *p, size_t n, const int32_t *cell, const double *x, const double *ux,
                        const double *uy, const double *uz, const double *w, double dx, double dt)
{
    if (int rc = check_pop(p)) return rc;
    if (n > p->cap) {
        set_error("opal_pop_upload_soa: %zu particles exceed the capacity %zu", n, p->cap);
        return OPAL_ERR_CAPACITY;
    }
    int k = p->cur;
    OPAL_HIP(hipMemcpyAsync(p->d_cell[k], cell, sizeof(int) * n, hipMemcpyHostToDevice, p->stream));
    OPAL_HIP(hipMemcpyAsync(p->d_x[k], x, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    OPAL_HIP(hipMemcpyAsync(p->d_ux[k], ux, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    OPAL_HIP(hipMemcpyAsync(p->d_uy[k], uy, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    OPAL_HIP(hipMemcpyAsync(p->d_uz[k], uz, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    OPAL_HIP(hipMemcpyAsync(p->d_w[k], w, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    p->cold_yz = true;
    for (size_t i = 0; i < n && p->cold_yz; i++) p->cold_yz = uy[i] == 0.0 && uz[i] == 0.0;
    p->w_uniform = n > 0;
    p->w0 = n ? w[0] : 0.0;
    for (size_t i = 1; i < n && p->w_uniform; i++) p->w_uniform = w[i] == p->w0;
    p->h_ctr->n_slots = (unsigned)n;
    OPAL_HIP(hipMemsetAsync(p->d_ctr, 0, sizeof(PopCounters), p->stream));
    OPAL_HIP(hipMemcpyAsync(&p->d_ctr->n_slots, &p->h_ctr->n_slots, sizeof(unsigned), hipMemcpyHostToDevice, p->stream));
    p->n_upper = n;
    if (p->upper_pending) {  // a snapshot of the previous contents
        hipEventSynchronize(p->upper_event);
        p->upper_pending = false;
    }
    p->sorted_count_pending = false;
    p->count_cells = 0;
    p->miss_onset = p->last_onset = p->last_trigger = p->preempt_at = 0;
    p->added_since_snapshot = 0;
    p->steps_since_sort = 0;
    p->last_interval = 0;
    p->emit_list_valid = false;
    p->sorted_exact = false;
    p->dead_cost = 0.0;
    p->ctr_pending = false;  // (see do_sort)
    p->n_tab = 0;     // no re-binning yet: uniform chunks
    p->n_sorted = 0;
    p->dx_last = dx;
    {   // where each cell's particles are, if the caller's order is by cell (the loader's is)
        free(p->h_cell_end);
        p->h_cell_end = nullptr;
        p->n_cell_end = 0;
        bool sorted = n > 0;
        for (size_t i = 1; i < n && sorted; i++) sorted = cell[i] >= cell[i - 1];
        if (sorted && cell[0] >= 0) {
            size_t ncells = (size_t)cell[n - 1] + 1;
            p->h_cell_end = (unsigned *)calloc(ncells, sizeof(unsigned));
            if (p->h_cell_end) {
                for (size_t i = 0; i < n; i++) p->h_cell_end[cell[i]] = (unsigned)(i + 1);
                for (size_t c = 1; c < ncells; c++)
                    if (!p->h_cell_end[c]) p->h_cell_end[c] = p->h_cell_end[c - 1];  // empty cell
                p->n_cell_end = ncells;
                // uniform chunks until the first re-binning: about a quarter of a tile each, as do_sort sizes them
                size_t nonempty = 1;
                for (size_t i = 1; i < n; i++) nonempty += cell[i] != cell[i - 1];
                int ppt = (int)((TILE / 4) * ((double)n / (double)nonempty) / BLOCK);
                p->ppt = std::max(1, std::min(pop_max_ppt(p), ppt));
            }
        }
    }
    // T::create: prev_x (only kept if the caller later uses the separate deposit operator)
    if (n) {
        if (int rc = ensure_prev(p)) return rc;
        unsigned nb = (unsigned)((n + 255) / 256);
        if (p->species == OPAL_ELECTRON)
            k_init_prev<OPAL_ELECTRON><<<nb, 256, 0, p->stream>>>(p->d_cell[k], p->d_x[k], p->d_ux[k], p->d_uy[k], p->d_uz[k], p->d_prev_x, (unsigned)n, dt, dx);
        else if (p->species == OPAL_PHOTON)
            k_init_prev<OPAL_PHOTON><<<nb, 256, 0, p->stream>>>(p->d_cell[k], p->d_x[k], p->d_ux[k], p->d_uy[k], p->d_uz[k], p->d_prev_x, (unsigned)n, dt, dx);
        else
            k_init_prev<OPAL_ION><<<nb, 256, 0, p->stream>>>(p->d_cell[k], p->d_x[k], p->d_ux[k], p->d_uy[k], p->d_uz[k], p->d_prev_x, (unsigned)n, dt, dx);
        p->prev_valid = true;
        OPAL_HIP(hipGetLastError());
    }
    if (p->extras && n)
        if (int rc = reset_extras(p, 0, n)) return rc;
    OPAL_HIP(hipStreamSynchronize(p->stream));
    return OPAL_OK;
}

int opal_pop_len(const opal_pop_t *p, size_t *n)
{
    if (int rc = check_pop(p)) return rc;
    // live = slots - dead; count on the host from the cell array (diagnostic path)
    if (int rc = fetch_counters(p)) return rc;
    size_t slots = p->h_ctr->n_slots;
    std::vector<int> cell(slots);
    if (slots) OPAL_HIP(hipMemcpy(cell.data(), p->d_cell[p->cur], sizeof(int) * slots, hipMemcpyDeviceToHost));
    size_t live = 0;
    for (size_t i = 0; i < slots; i++) live += cell[i] != DEAD_CELL;
    *n = live;
    return OPAL_OK;
}

int opal_pop_download_soa(const opal_pop_t *p, int32_t *cell, double *x, double *prev_x, double *ux, double *uy,
                          double *uz, double *w)
{
    if (int rc = check_pop(p)) return rc;
    if (int rc = fetch_counters(p)) return rc;
    size_t slots = p->h_ctr->n_slots;
    if (!slots) return OPAL_OK;
    int k = p->cur;
    std::vector<int> hc(slots);
    OPAL_HIP(hipMemcpy(hc.data(), p->d_cell[k], sizeof(int) * slots, hipMemcpyDeviceToHost));
    std::vector<double> tmp(slots);
    auto pull = [&](const double *src, double *dst) -> int {
        if (!dst) return OPAL_OK;
        if (!src) {
            set_error("opal_pop_download_soa: attribute not resident (prev_x needs a prior advance)");
            return OPAL_ERR_ARG;
        }
        OPAL_HIP(hipMemcpy(tmp.data(), src, sizeof(double) * slots, hipMemcpyDeviceToHost));
        size_t o = 0;
        for (size_t i = 0; i < slots; i++)
            if (hc[i] != DEAD_CELL) dst[o++] = tmp[i];
        return OPAL_OK;
    };
    if (int rc = pull(p->d_x[k], x)) return rc;
    if (int rc = pull(p->prev_valid ? p->d_prev_x : nullptr, prev_x)) return rc;
    if (int rc = pull(p->d_ux[k], ux)) return rc;
    if (int rc = pull(p->d_uy[k], uy)) return rc;
    if (int rc = pull(p->d_uz[k], uz)) return rc;
    if (int rc = pull(p->d_w[k], w)) return rc;
    if (cell) {
        size_t o = 0;
        for (size_t i = 0; i < slots; i++)
            if (hc[i] != DEAD_CELL) cell[o++] = hc[i];
    }
    return OPAL_OK;
}

int opal_pop_track_fields(opal_pop_t *p, unsigned fields)
{
    if (int rc = check_pop(p)) return rc;
    if (p->species == OPAL_ION) return OPAL_OK;  // ion.rs has no chi / y / z; its `work` is not surfaced
    bool need[NEX] = {};
    need[EX_CHI] = fields & (OPAL_TRACK_CHI | OPAL_TRACK_EMISSION);  // emission reads the electron's chi
    need[EX_WORK] = (fields & OPAL_TRACK_WORK) && p->species == OPAL_ELECTRON;  // photon.rs:207-209: always 0
    need[EX_Y] = need[EX_Z] = fields & OPAL_TRACK_YZ;
    need[EX_TAU] = (fields & OPAL_TRACK_EMISSION) && p->species == OPAL_ELECTRON;
    need[EX_ID] = need[EX_TAU] || (fields & OPAL_TRACK_ID);
    need[EX_TAU_ABS] = need[EX_TAU_ST] = need[EX_BIRTH] = (fields & OPAL_TRACK_ABSORPTION) && p->species == OPAL_PHOTON;
    bool fresh[NEX], any_fresh = false, any = false;
    for (int q = 0; q < NEX; q++) {
        fresh[q] = need[q] && !p->d_ex[q];
        if (fresh[q]) OPAL_HIP(hipMalloc(&p->d_ex[q], sizeof(double) * p->cap));
        if (!need[q] && p->d_ex[q]) {
            OPAL_HIP(hipStreamSynchronize(p->stream));
            hipFree(p->d_ex[q]);
            p->d_ex[q] = nullptr;
            if (p->d_ex_alt[q]) hipFree(p->d_ex_alt[q]);
            p->d_ex_alt[q] = nullptr;
        }
        any_fresh |= fresh[q];
        any |= need[q];
    }
    p->extras = !any ? 0 : need[EX_TAU] ? 2 : 1;
    p->emit_list_valid = false;
    if (any_fresh) {
        // start the new arrays as T::create does for every slot (arrays tracked before keep their values)
        double *keep[NEX];
        for (int q = 0; q < NEX; q++) {
            keep[q] = p->d_ex[q];
            if (!fresh[q]) p->d_ex[q] = nullptr;
        }
        int rc = reset_extras(p, 0, p->cap);
        for (int q = 0; q < NEX; q++) p->d_ex[q] = keep[q];
        if (rc) return rc;
    }
    return OPAL_OK;
}

int opal_pop_track_extras(opal_pop_t *p, int on)
{
    if (int rc = check_pop(p)) return rc;
    if (on < 0 || on > 2) {
        set_error("opal_pop_track_extras: level must be 0, 1 or 2");
        return OPAL_ERR_ARG;
    }
    unsigned fields = 0;
    if (on) fields = OPAL_TRACK_CHI | OPAL_TRACK_WORK | OPAL_TRACK_YZ | (p->species == OPAL_PHOTON ? OPAL_TRACK_ID : 0u);
    if (on == 2) fields |= OPAL_TRACK_EMISSION;
    return opal_pop_track_fields(p, fields);
}

int opal_pop_keep_prev_x(opal_pop_t *p, int on)
{
    if (int rc = check_pop(p)) return rc;
    p->keep_prev = on != 0;
    if (!on) p->prev_valid = false;
    return OPAL_OK;
}

int opal_pop_download_extras(const opal_pop_t *p, double *chi, double *work, double *y, double *z)
{
    if (int rc = check_pop(p)) return rc;
    if (!p->extras) {
        set_error("opal_pop_download_extras: extras are not tracked");
        return OPAL_ERR_ARG;
    }
    if (int rc = fetch_counters(p)) return rc;
    size_t slots = p->h_ctr->n_slots;
    std::vector<int> hc(slots);
    std::vector<double> tmp(slots);
    if (slots) OPAL_HIP(hipMemcpy(hc.data(), p->d_cell[p->cur], sizeof(int) * slots, hipMemcpyDeviceToHost));
    const double *src[4] = {p->d_ex[EX_CHI], p->d_ex[EX_WORK], p->d_ex[EX_Y], p->d_ex[EX_Z]};
    double *dst[4] = {chi, work, y, z};
    for (int k = 0; k < 4; k++) {
        if (!dst[k] || !slots) continue;
        if (!src[k]) {  // a photon has no `work` (photon.rs:207-209)
            size_t live = 0;
            for (size_t i = 0; i < slots; i++) live += hc[i] != DEAD_CELL;
            std::fill(dst[k], dst[k] + live, 0.0);
            continue;
        }
        OPAL_HIP(hipMemcpy(tmp.data(), src[k], sizeof(double) * slots, hipMemcpyDeviceToHost));
        size_t o = 0;
        for (size_t i = 0; i < slots; i++)
            if (hc[i] != DEAD_CELL) dst[k][o++] = tmp[i];
    }
    return OPAL_OK;
}

// Optical depths against emission and the ids that key the emission random numbers, in slot order
// (dead slots skipped), for electrons tracked at level 2; for photons `id` is the parent electron's.
int opal_pop_download_emission_state(const opal_pop_t *p, double *tau, uint64_t *id)
{
    if (int rc = check_pop(p)) return rc;
    if ((tau && !p->d_ex[EX_TAU]) || (id && !p->d_ex[EX_ID])) {
        set_error("opal_pop_download_emission_state: not tracked (opal_pop_track_extras level 2 for electrons, >= 1 for photons)");
        return OPAL_ERR_ARG;
    }
    if (int rc = fetch_counters(p)) return rc;
    size_t slots = p->h_ctr->n_slots;
    if (!slots) return OPAL_OK;
    std::vector<int> hc(slots);
    std::vector<double> tmp(slots);
    OPAL_HIP(hipMemcpy(hc.data(), p->d_cell[p->cur], sizeof(int) * slots, hipMemcpyDeviceToHost));
    if (tau) {
        OPAL_HIP(hipMemcpy(tmp.data(), p->d_ex[EX_TAU], sizeof(double) * slots, hipMemcpyDeviceToHost));
        size_t o = 0;
        for (size_t i = 0; i < slots; i++)
            if (hc[i] != DEAD_CELL) tau[o++] = tmp[i];
    }
    if (id) {
        OPAL_HIP(hipMemcpy(tmp.data(), p->d_ex[EX_ID], sizeof(double) * slots, hipMemcpyDeviceToHost));
        size_t o = 0;
        for (size_t i = 0; i < slots; i++)
            if (hc[i] != DEAD_CELL) memcpy(&id[o++], &tmp[i], sizeof(uint64_t));
    }
    return OPAL_OK;
}

// Overwrites them for the first n slots (as uploaded: call right after opal_pop_upload_soa).
int opal_pop_upload_emission_state(opal_pop_t *p, size_t n, const double *tau, const uint64_t *id)
{
    if (int rc = check_pop(p)) return rc;
    if ((tau && !p->d_ex[EX_TAU]) || (id && !p->d_ex[EX_ID]) || n > p->cap) {
        set_error("opal_pop_upload_emission_state: not tracked (opal_pop_track_extras level 2), or n exceeds the capacity");
        return OPAL_ERR_ARG;
    }
    if (!n) return OPAL_OK;
    if (tau) OPAL_HIP(hipMemcpyAsync(p->d_ex[EX_TAU], tau, sizeof(double) * n, hipMemcpyHostToDevice, p->stream));
    if (id) OPAL_HIP(hipMemcpyAsync(p->d_ex[EX_ID], id, sizeof(uint64_t) * n, hipMemcpyHostToDevice, p->stream));
    p->emit_list_valid = false;  // optical depths changed behind the push's back
    OPAL_HIP(hipStreamSynchronize(p->stream));
    return OPAL_OK;
}

int opal_pop_counters(const opal_pop_t *p, uint32_t *out)
{
    if (int rc = check_pop(p)) return rc;
    if (int rc = fetch_counters(p)) return rc;
    out[0] = p->h_ctr->n_slots;
    out[1] = p->h_ctr->n_deleted;
    out[2] = p->h_ctr->n_slow;
    out[3] = (uint32_t)p->steps_since_sort;
    out[4] = (uint32_t)p->ppt;
    out[5] = p->h_ctr->n_live;
    out[6] = p->h_ctr->n_nonempty;
    out[7] = (uint32_t)p->n_upper;
    return OPAL_OK;
}

/* diagnostic: start (n > 0) or read and stop (out != null) recording the first tile misses of the pushes:
 * out[0] = number of misses since the start, then up to 4096 x (workgroup, slot, cell, tile base cell) */
int opal_set_elision(int on)
{
    g_elision = on != 0;
    return OPAL_OK;
}

int opal_pop_debug_misses(opal_pop_t *p, int start, int32_t *out)
{
    if (int rc = check_pop(p)) return rc;
    if (start) {
        if (!p->d_dbg) OPAL_HIP(hipMalloc(&p->d_dbg, sizeof(int) * (1 + 4 * 4096)));
        OPAL_HIP(hipMemsetAsync(p->d_dbg, 0, sizeof(int) * (1 + 4 * 4096), p->stream));
        return OPAL_OK;
    }
    if (!p->d_dbg || !out) {
        set_error("opal_pop_debug_misses: not recording");
        return OPAL_ERR_ARG;
    }
    OPAL_HIP(hipMemcpyAsync(out, p->d_dbg, sizeof(int) * (1 + 4 * 4096), hipMemcpyDeviceToHost, p->stream));
    OPAL_HIP(hipStreamSynchronize(p->stream));
    hipFree(p->d_dbg);
    p->d_dbg = nullptr;
    return OPAL_OK;
}

int opal_pop_set_sort_interval(opal_pop_t *p, int interval)
{
    if (int rc = check_pop(p)) return rc;
    p->sort_interval = interval;
    return OPAL_OK;
}

int opal_pop_sort(opal_pop_t *p, const opal_grid_t *grid)
{
    if (int rc = check_pop(p)) return rc;
    if (!grid) {
        set_error("opal_pop_sort: null grid");
        return OPAL_ERR_ARG;
    }
    return do_sort(p, grid);
}

int opal_pop_advance(opal_pop_t *p, opal_link_t *link, const opal_grid_t *grid, double dt)
{
    if (int rc = check_pop(p)) return rc;
    if (!grid) {
        set_error("opal_pop_advance: null grid");
        return OPAL_ERR_ARG;
    }
    const bool prev = p->keep_prev && p->species != OPAL_PHOTON;  // photons are never deposited (main.rs:254-258)
    if (prev)
        if (int rc = ensure_prev(p)) return rc;
    p->prev_valid = prev;
    if (!prev) p->keep_prev = false;
    if (int rc = advance_with_exchange(p, link, const_cast<opal_grid_t *>(grid), dt, MODE_ADVANCE)) return rc;
    if (int rc = maybe_sort(p, grid)) return rc;
    return OPAL_OK;
}

int opal_pop_advance_deposit(opal_pop_t *p, opal_link_t *link, opal_grid_t *grid, double dt)
{
    if (int rc = check_pop(p)) return rc;
    if (!grid) {
        set_error("opal_pop_advance_deposit: null grid");
        return OPAL_ERR_ARG;
    }
    p->prev_valid = false;
    if (int rc = advance_with_exchange(p, link, grid, dt, MODE_FUSED)) return rc;
    if (int rc = maybe_sort(p, grid)) return rc;
    return OPAL_OK;
}

int opal_grid_deposit(opal_grid_t *g, const opal_pop_t *pop, double dt)
{
    if (!g) {
        set_error("opal_grid_deposit: null grid");
        return OPAL_ERR_ARG;
    }
    if (int rc = check_pop(pop)) return rc;
    if (pop->n_upper == 0) return OPAL_OK;  // yee.rs:557-559
    if (!pop->prev_valid) {
        set_error("opal_grid_deposit: prev_x is not resident; call opal_pop_advance (not the fused pass) first");
        return OPAL_ERR_ARG;
    }
    return run_push(const_cast<opal_pop_t *>(pop), nullptr, g, dt, MODE_DEPOSIT);
}

int opal_pop_total_kinetic_energy(const opal_pop_t *p, double *energy)
{
    if (int rc = check_pop(p)) return rc;
    if (p->n_upper == 0) {
        *energy = 0.0;
        return OPAL_OK;
    }
    int k = p->cur;
    int nblk = (int)std::min<size_t>((p->n_upper + BLOCK - 1) / BLOCK, p->partial_cap);
    if (p->species == OPAL_ELECTRON)
        k_kinetic<OPAL_ELECTRON><<<nblk, BLOCK, 0, p->stream>>>(p->d_cell[k], p->d_ux[k], p->d_uy[k], p->d_uz[k], p->d_w[k], p->d_ctr, p->a, p->d_partial);
    else if (p->species == OPAL_PHOTON)
        k_kinetic<OPAL_PHOTON><<<nblk, BLOCK, 0, p->stream>>>(p->d_cell[k], p->d_ux[k], p->d_uy[k], p->d_uz[k], p->d_w[k], p->d_ctr, p->a, p->d_partial);
    else
        k_kinetic<OPAL_ION><<<nblk, BLOCK, 0, p->stream>>>(p->d_cell[k], p->d_ux[k], p->d_uy[k], p->d_uz[k], p->d_w[k], p->d_ctr, p->a, p->d_partial);
    k_fold<<<1, BLOCK, 0, p->stream>>>(p->d_partial, nblk, p->d_partial + p->partial_cap);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpyAsync(energy, p->d_partial + p->partial_cap, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    OPAL_HIP(hipStreamSynchronize(p->stream));
    return check_pop_err(const_cast<opal_pop_t *>(p));
}

}  // extern "C"

namespace opal {
int pop_check_err(opal_pop_t *p) { return check_pop_err(p); }
// pieces of the fused multi-subdomain step (sim.cpp): push without exchanging, then unpack after the
// single combined exchange
int pop_push_fused_noexchange(opal_pop_t *p, opal_link_t *link, opal_grid_t *g, double dt)
{
    p->prev_valid = false;
    if (int rc = run_push(p, link, g, dt, MODE_FUSED)) return rc;
    return link ? pack_leaver_extras(p, link, g) : OPAL_OK;
}
// Population::advance's push (separate-operator order) without its exchange: leavers are packed into the
// link's messages, which the caller ships once for all species
int pop_push_advance_noexchange(opal_pop_t *p, opal_link_t *link, opal_grid_t *g, double dt)
{
    const bool prev = p->keep_prev && p->species != OPAL_PHOTON;
    if (prev)
        if (int rc = ensure_prev(p)) return rc;
    p->prev_valid = prev;
    if (int rc = run_push(p, link, g, dt, MODE_ADVANCE)) return rc;
    return link ? pack_leaver_extras(p, link, g) : OPAL_OK;
}
// both species of a fused step in one launch; no exchange, no re-binning bookkeeping
int pops_push_fused_pair_noexchange(opal_pop_t *pe, opal_pop_t *pi, opal_link_t *link, opal_grid_t *g, double dt)
{
    pe->prev_valid = false;
    pi->prev_valid = false;
    if (int rc = run_push_pair(pe, pi, link, g, dt)) return rc;
    if (!link) return OPAL_OK;
    if (int rc = pack_leaver_extras(pe, link, g)) return rc;
    return pack_leaver_extras(pi, link, g);
}
// the reference's separate operators for both charged species of a lone subdomain, one launch each:
// Population::advance x 2 (no exchange, no re-binning bookkeeping), then YeeGrid::deposit x 2
int pops_advance_pair_noexchange(opal_pop_t *pe, opal_pop_t *pi, opal_grid_t *g, double dt)
{
    for (opal_pop_t *p : {pe, pi}) {
        const bool prev = p->keep_prev;
        if (prev)
            if (int rc = ensure_prev(p)) return rc;
        p->prev_valid = prev;
    }
    return run_push_pair(pe, pi, nullptr, g, dt, PART_ALL, MODE_ADVANCE);
}
int pops_deposit_pair(opal_pop_t *pe, opal_pop_t *pi, opal_grid_t *g, double dt)
{
    if (!pe->prev_valid || !pi->prev_valid) {
        set_error("deposit: prev_x is not resident; advance (not the fused pass) first");
        return OPAL_ERR_ARG;
    }
    return run_push_pair(pe, pi, nullptr, g, dt, PART_ALL, MODE_DEPOSIT);
}
// A fused step in two launches so that the neighbour exchange can overlap the second: true when this step of
// both species can be split into edge chunks (everything that can leave or touch the halo cells) and bulk
// chunks; then call pops_push_fused_pair_part with edges = true, start the exchange, and call it with false.
bool pops_plan_split(opal_pop_t *pe, opal_pop_t *pi, opal_link_t *link, const opal_grid_t *g)
{
    if (!pair_launch_possible(pe, pi, link)) return false;
    bool a = pop_plan_split(pe, g), b = pop_plan_split(pi, g);
    return a && b;
}
int pops_push_fused_pair_part(opal_pop_t *pe, opal_pop_t *pi, opal_link_t *link, opal_grid_t *g, double dt, bool edges,
                              hipStream_t on)
{
    pe->prev_valid = false;
    pi->prev_valid = false;
    return run_push_pair(pe, pi, link, g, dt, edges ? PART_EDGES : PART_BULK, MODE_FUSED, on);
}
int pop_maybe_sort(opal_pop_t *p, const opal_grid_t *g) { return maybe_sort(p, g); }
// The policy's step for a step's species, leaving the re-binnings it decides on to the caller (pops_sort_decided) and
// the publication of the counters to a kernel the caller is about to launch anyway (grid_sync_advance): `pub` comes
// back with n = 0 if nothing is due (or if the batch went out by a launch of its own: populations on other streams).
int pops_policy(opal_pop_t *const *pops, int npops, PublishBatch *pub, bool *sort_now)
{
    memset(pub, 0, sizeof *pub);
    PublishScope scope;
    for (int i = 0; i < npops; i++) {
        sort_now[i] = false;
        if (pops[i])
            if (int rc = policy_step(pops[i], sort_now[i])) return rc;
    }
    if (scope.b.n && pops[0] && scope.stream == pops[0]->stream) {
        *pub = scope.b;
        scope.b.n = 0;
    }
    return OPAL_OK;
}
int pops_sort_decided(opal_pop_t *const *pops, int npops, const bool *sort_now, const opal_grid_t *g)
{
    for (int i = 0; i < npops; i++)
        if (pops[i] && sort_now[i])
            if (int rc = do_sort(pops[i], g, false)) return rc;
    return OPAL_OK;
}
// the re-binning bookkeeping of a step's species together (their counters are published in one launch)
int pops_maybe_sort(opal_pop_t *const *pops, int npops, const opal_grid_t *g)
{
    PublishScope scope;
    for (int i = 0; i < npops; i++)
        if (pops[i])
            if (int rc = maybe_sort(pops[i], g)) return rc;
    return OPAL_OK;
}
// exact binning by cell now (photon absorption pairs every photon with the electrons of its cell): leaves
// p->d_count[c] = one past the last slot of cell c
int pop_sort_exact(opal_pop_t *p, const opal_grid_t *g)
{
    if (p->sorted_exact && p->d_count && p->count_cells == g->size) return OPAL_OK;  // nothing has moved since
    return do_sort(p, g);
}
// The host's bound on the slots in use (n_upper) when particles are appended by the device without the host
// knowing how many (photon emission): the true count is copied to pinned memory asynchronously after an
// append; once a copy has landed, n_upper drops to that count plus whatever has been reserved since.
int pop_tighten_upper(opal_pop_t *p)
{
    if (!p->upper_pending) return OPAL_OK;
    hipError_t q = hipEventQuery(p->upper_event);
    if (q == hipErrorNotReady) return OPAL_OK;
    OPAL_HIP(q);
    p->upper_pending = false;
    size_t bound = (size_t)*p->h_upper + p->added_since_snapshot;
    if (bound < p->n_upper) p->n_upper = bound;
    if (p->sorted_count_pending) {  // the count a re-binning left (do_sort, wait = false)
        p->n_sorted = *p->h_upper;
        p->sorted_count_pending = false;
    }
    return OPAL_OK;
}
int pop_snapshot_upper(opal_pop_t *p, size_t just_added)
{
    if (p->upper_pending) {  // an older snapshot is still in flight: it does not know about this append
        p->added_since_snapshot += just_added;
        return OPAL_OK;
    }
    if (!p->upper_event) {
        OPAL_HIP(hipEventCreateWithFlags(&p->upper_event, hipEventDisableTiming));
        OPAL_HIP(hipHostMalloc((void **)&p->h_upper, sizeof(unsigned)));
    }
    OPAL_HIP(hipMemcpyAsync(p->h_upper, &p->d_ctr->n_slots, sizeof(unsigned), hipMemcpyDeviceToHost, p->stream));
    OPAL_HIP(hipEventRecord(p->upper_event, p->stream));
    p->upper_pending = true;
    p->added_since_snapshot = 0;  // taken behind this append in stream order
    return OPAL_OK;
}
// room for `extra` more particles behind the slots in use
int pop_reserve(opal_pop_t *p, size_t extra)
{
    if (p->n_upper + extra <= p->cap) return OPAL_OK;
    size_t want = p->n_upper + extra;
    return grow(p, want + want / 2);
}
// the arrivals of every species of a fused linked step in one launch, then each one's re-binning policy;
// also resets the send messages' leaver counts, so the next step needs no link_begin
// the unpack alone: a step with photon emission runs emit_radiation between it and the re-binning policy, so
// that the emitter list the push left behind (slot indices) is still good when it is consumed
int pops_unpack(opal_pop_t *const *pops, int npops, opal_link_t *link, const opal_grid_t *g)
{
    return unpack_arrivals(pops, npops, link, g, true);
}
// ... `sync` (and `adv`: the field solve of a small grid) run in the same launch; with `peer_exchange` that launch also
// makes the exchange (peer-mapped mailboxes).  The policy's counters are published by it too.
int pops_unpack_and_maybe_sort(opal_pop_t *const *pops, int npops, opal_link_t *link, const opal_grid_t *g,
                               const SyncArgs *sync, const AdvArgs *adv, bool peer_exchange)
{
    bool sort_now[4] = {};
    PublishBatch pub{};
    {
        PublishScope scope;  // collects what the species ask to have published ...
        for (int i = 0; i < npops && i < 4; i++)
            if (pops[i])
                if (int rc = policy_step(pops[i], sort_now[i])) return rc;
        if (scope.b.n && scope.stream == (pops[0] ? pops[0]->stream : link->stream)) {
            pub = scope.b;  // ... and hands it to the tail kernel instead of launching
            scope.b.n = 0;
        }
    }
    if (int rc = unpack_arrivals(pops, npops, link, g, true, sync, adv, peer_exchange, &pub)) return rc;
    for (int i = 0; i < npops && i < 4; i++)
        if (pops[i] && sort_now[i])
            if (int rc = do_sort(pops[i], g, false)) return rc;
    return OPAL_OK;
}

// Diagnostic (OPAL_DEBUG_CELLS=1, sim.cpp): the first live slot whose cell lies outside [0, size), with where it stands
__global__ void k_check_cells(const int *__restrict__ cell, const PopCounters *ctr, int size, unsigned *out)
{
    const unsigned n = ctr->n_slots;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = cell[i];
        if (c != DEAD_CELL && (unsigned)c >= (unsigned)size)
            if (atomicCAS(out, 0u, 1u) == 0u) {
                out[1] = i; out[2] = (unsigned)c; out[3] = n; out[4] = ctr->n_sorted; out[5] = ctr->n_dead;
            }
    }
}
int pop_debug_check_cells(opal_pop_t *p, const opal_grid_t *g, const char *when, size_t step)
{
    if (!p || p->n_upper == 0) return OPAL_OK;
    unsigned *d = nullptr, h[8] = {0};
    OPAL_HIP(hipMalloc(&d, sizeof h));
    OPAL_HIP(hipMemsetAsync(d, 0, sizeof h, p->stream));
    k_check_cells<<<1024, 256, 0, p->stream>>>(p->d_cell[p->cur], p->d_ctr, (int)g->size, d);
    OPAL_HIP(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, p->stream));
    OPAL_HIP(hipStreamSynchronize(p->stream));
    hipFree(d);
    if (h[0]) {
        set_error("debug: species %d, %s step %zu: slot %u holds cell %d (subdomain of %zu cells); slots in use %u, sorted %u, vacated %u; "
                  "host: n_upper %zu, n_sorted %u, n_tab %u, ppt %d, steps since re-binning %d, reframe %d",
                  p->species, when, step, h[1], (int)h[2], g->size, h[3], h[4], h[5], p->n_upper, p->n_sorted, p->n_tab, p->ppt,
                  p->steps_since_sort, (int)p->reframe);
        return OPAL_ERR_ASSERT;
    }
    return OPAL_OK;
}

// ---- a re-cut of the decomposition (sim.cpp: sim_recut), the populations' part
// live particles of `p` (drains its stream)
int pop_live_count(opal_pop_t *p, size_t *n)
{
    if (int rc = fetch_counters(p)) return rc;
    *n = (size_t)p->h_ctr->n_slots - std::min(p->h_ctr->n_slots, p->h_ctr->n_dead);
    return OPAL_OK;
}
// Re-bins `p` exactly and waits: *cell_end[c] is then one past the last slot of cell c, c < g->size (null: no particles)
int pop_cell_ends(opal_pop_t *p, const opal_grid_t *g, const unsigned **cell_end)
{
    *cell_end = nullptr;
    if (p->upper_pending)
        if (int rc = pop_tighten_upper(p)) return rc;
    if (p->n_upper == 0) return OPAL_OK;
    if (int rc = do_sort(p, g, true)) return rc;
    if (p->n_cell_end == g->size) *cell_end = p->h_cell_end;
    return OPAL_OK;
}
// particles of `p` per cell of `g` (live slots; drains the stream): counts[c], c < g->size
int pop_cell_histogram(opal_pop_t *p, const opal_grid_t *g, std::vector<unsigned> &counts)
{
    const size_t ncells = g->size;
    counts.assign(ncells, 0u);
    if (p->upper_pending)
        if (int rc = pop_tighten_upper(p)) return rc;
    if (p->n_upper == 0) return OPAL_OK;
    if (p->hist_cap < ncells + 2) {
        if (p->d_hist) hipFree(p->d_hist);
        p->d_hist = nullptr;
        p->hist_cap = 0;
        OPAL_HIP(hipMalloc(&p->d_hist, sizeof(unsigned) * (ncells + ncells / 2 + 2)));
        p->hist_cap = ncells + ncells / 2 + 2;
    }
    unsigned *d_hist = p->d_hist;
    hipError_t e = hipMemsetAsync(d_hist, 0, sizeof(unsigned) * (ncells + 2), p->stream);
    if (e == hipSuccess) {
        const unsigned nblk = (unsigned)((p->n_upper + 256 * SORT_PPT - 1) / (256 * SORT_PPT));
        k_sort_hist<<<nblk, 256, 0, p->stream>>>(p->d_cell[p->cur], p->d_ctr, d_hist);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_hist, sizeof(unsigned) * ncells, hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    OPAL_HIP(e);
    return OPAL_OK;
}
// live particles of `p` as the re-binning policy last saw them published (at most a few steps old), without touching the
// stream; false: nothing published yet (or the population is not under the adaptive policy): pop_live_count
bool pop_live_estimate(const opal_pop_t *p, size_t *n)
{
    if (p->sort_interval >= 0 || p->pub_seq == 0 || p->h_ctr_seen.n_slots == 0) return false;
    *n = (size_t)p->h_ctr_seen.n_slots - std::min(p->h_ctr_seen.n_slots, p->h_ctr_seen.n_dead);
    return true;
}
// The step in which the cuts move, the particles' side (PushArgs: out_shift): from here to pops_reframe_end the pushes of
// `pops` store cells counted from the new left cut -- old cell `shift` -- and send what lies outside the `new_size` cells
// from there on; nothing is re-binned meanwhile.
void pops_reframe_begin(opal_pop_t *const *pops, int npops, int shift, size_t new_size)
{
    for (int i = 0; i < npops; i++)
        if (pops[i]) {
            pops[i]->reframe = true;
            pops[i]->reframe_shift = shift;
            pops[i]->reframe_size = (int)new_size;
        }
}
// the re-binnings the policy decided on while the cells counted from cuts the grid had not followed yet
int pops_sort_deferred(opal_pop_t *const *pops, int npops, const opal_grid_t *g)
{
    for (int i = 0; i < npops; i++)
        if (pops[i] && pops[i]->sort_deferred) {
            pops[i]->sort_deferred = false;
            if (int rc = do_sort(pops[i], g, false)) return rc;
        }
    return OPAL_OK;
}
void pops_reframe_end(opal_pop_t *const *pops, int npops)
{
    for (int i = 0; i < npops; i++)
        if (pops[i]) {
            pops[i]->reframe = false;
            pops[i]->n_cell_end = 0;   // the host's cell map and the re-binning's cell ends were of the old frame
            pops[i]->count_cells = 0;
            pops[i]->sorted_exact = false;
        }
}
// packs what the cells given away hold into the link's send messages (after link_begin) and moves the stayers' cell
// indices to the new frame; `size` is the subdomain's size BEFORE the re-cut
int pops_recut_pack(opal_pop_t *const *pops, int npops, opal_link_t *link, size_t size, int give_left, int give_right, int delta)
{
    for (int i = 0; i < npops; i++) {
        opal_pop_t *p = pops[i];
        if (!p || p->n_upper == 0) continue;
        if (p->species >= link->nslots) {
            set_error("this link's messages have no room for species %d (see opal_link_set_layout)", p->species);
            return OPAL_ERR_LINK;
        }
        RecutPackArgs a;
        memset(&a, 0, sizeof a);
        const int k = p->cur;
        a.cell = p->d_cell[k];
        a.f[0] = p->d_x[k]; a.f[1] = p->d_ux[k]; a.f[2] = p->d_uy[k]; a.f[3] = p->d_uz[k]; a.f[4] = p->d_w[k];
        a.prev = p->prev_valid ? p->d_prev_x : nullptr;
        for (int q = 0; q < NEX; q++) a.ex[q] = p->extras ? p->d_ex[q] : nullptr;
        a.n = (unsigned)p->n_upper;
        a.give_left = give_left;
        a.first_right = (int)size - give_right;
        a.delta = delta;
        for (int side = 0; side < 2; side++) {
            a.send[side] = link_send(link, side);
            a.cap[side] = link->send_cap[side];
        }
        a.layout = link->layout;
        a.slot = p->species;
        a.nextra = link->nfields > 6 ? (unsigned)(link->nfields - 6) : 0u;
        a.ctr = p->d_ctr;
        const unsigned nblk = (unsigned)std::min<size_t>((p->n_upper + 255) / 256, 8192);
        k_recut_pack<<<nblk, 256, 0, p->stream>>>(a);
        OPAL_HIP(hipGetLastError());
        p->sorted_exact = false;
        p->emit_list_valid = false;
        p->n_cell_end = 0;  // the host's cell map is of the old frame
    }
    return OPAL_OK;
}
// the arrivals (the exchange has been made), then every population in exact order on the re-cut grid `g`
int pops_recut_finish(opal_pop_t *const *pops, int npops, opal_link_t *link, const opal_grid_t *g)
{
    if (int rc = unpack_arrivals(pops, npops, link, g, true)) return rc;
    for (int i = 0; i < npops; i++) {
        opal_pop_t *p = pops[i];
        if (!p) continue;
        const int interval = p->last_interval;  // (what the policy knows of the population's temperature is not the re-cut's doing)
        if (int rc = do_sort(p, g, true)) return rc;
        p->last_interval = interval;
        if (int rc = check_pop_err(p)) return rc;
    }
    return OPAL_OK;
}
}
