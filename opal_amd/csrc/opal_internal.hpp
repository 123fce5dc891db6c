// Internal host-side definitions shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/opal_amd.h"

namespace opal {

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define OPAL_STR2(x) #x
#define OPAL_STR(x) OPAL_STR2(x)
#define OPAL_HIP(call)                                          \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return opal::hip_fail(e_, #call " at " __FILE__ ":" OPAL_STR(__LINE__)); \
    } while (0)

// Device-resident counters of a population (one 64-byte record).
struct PopCounters {
    unsigned int n_slots;      // slots in use (live + dead), device truth
    unsigned int n_deleted;    // particles removed at open edges since last reset
    unsigned int n_sent_left;  // leavers packed for the left / right neighbour (this exchange)
    unsigned int n_sent_right;
    unsigned int n_slow;       // particles that missed the LDS tile (pushed through the slow path) since last sort
    int err;                   // bit 0: dxi >= 1; bit 1: gather index; bit 2: deposit index; bit 3: leaver overflow
    unsigned int n_live;       // written by sort / count kernels
    unsigned int n_nonempty;   // non-empty cells at last sort
    unsigned int n_emit;       // electrons whose optical depth ran out in the last push (photon emission)
    unsigned int n_lost;       // photons that found no free slot (capacity is checked by the host first: stays 0)
    unsigned int n_dead;       // slots vacated (deleted or sent away) since the last re-binning
    unsigned int w_mixed;      // an arrival brought a weight other than the population's uniform one (see opal_pop::w_uniform)
    unsigned int hot_yz;       // some slot may hold a non-zero uy or uz (see opal_pop::cold_yz)
    unsigned int n_slow_pass;  // wave passes through the slow path since last sort (what the misses cost, see maybe_sort)
    unsigned int n_sorted;     // slots [0, n_sorted) are where the last re-binning put them (chunk table); behind: appended since
    unsigned int bad_cell;     // a cell index that failed the gather's range check (err bit 1), for the error message
};

// EX_TAU_ABS, EX_TAU_ST, EX_BIRTH: a photon's optical depths against absorption / stimulated emission and its birth
// time (photon.rs:28-30), carried only when photon absorption is on
struct PubCounters {
    PopCounters c;
    volatile unsigned seq;  // written after c, behind a system-scope fence
    unsigned pad[15];
};

// the re-binning policy's view of the counters: the 16 words of up to four populations go to mapped host memory, each
// followed by the sequence number the host waits for (written by whichever kernel of the step carries the batch)
struct PublishBatch {
    const PopCounters *src[4];
    PubCounters *dst[4];
    unsigned seq[4];
    int n;
    // a device word the same kernel clears once the counters have gone out: the electrons' emitter count, which the step's
    // emission has consumed by then (run_push would otherwise zero it with a launch of its own before every push)
    unsigned *zero;
};

enum { EX_CHI = 0, EX_WORK = 1, EX_Y = 2, EX_Z = 3, EX_TAU = 4, EX_ID = 5, EX_TAU_ABS = 6, EX_TAU_ST = 7, EX_BIRTH = 8, NEX = 9 };
enum { KERNEL_FUSED = 0, KERNEL_ADVANCE = 1, KERNEL_DEPOSIT = 2, KERNEL_FIELD = 3, KERNEL_SORT = 4, KERNEL_PHOTON = 5,
       KERNEL_RADIATE = 6, KERNEL_COUNT = 7 };  // (5: the photons' push; 6: k_radiate)
// which form of the fused / advance push a launch ran (bench.py prices a launch's algorithmic bytes by it):
// bit 0 electrostatic launch (uy, uz neither read nor written), bit 1 rho deposited, bit 2 / 3 the electrons' / ions'
// weight taken from the launch arguments instead of streamed; bit 4 a single-species launch (k_push)
enum { FORM_ES = 1, FORM_RHO = 2, FORM_W_ARG_E = 4, FORM_W_ARG_I = 8, FORM_SINGLE = 16, PUSH_FORMS = 32 };
void prof_form(unsigned form);
void prof_begin(int kernel, hipStream_t s);
void prof_end(int kernel, hipStream_t s);

}  // namespace opal

struct opal_grid {
    int id, ngrids, rank_to_left, rank_to_right;
    bool self_link;  // a lone periodic subdomain sends to itself through the link instead of wrapping in place
    int left_bdy, right_bdy;
    int L, R;  // left / right boundary sizes (cells)
    size_t size, len, lenp;
    double dx, xmin_global, offset;
    hipStream_t stream;
    double *d_eb[2];  // ping-pong field buffers, 6 components x lenp
    double *d_jb[2];  // jx jy jz rho, 4 x lenp, twice: the field solve zeroes the spare one for the next step
    int jcur;
    bool j_spare_zero;
    double *j() const { return d_jb[jcur]; }
    int cur;
    double *d_send;   // 2 x 80 doubles: what synchronize sends left / right
    double *d_damp;   // absorbing layer factors (1 - sigma), R entries (or null)
    double *d_scalar; // scratch for reductions
    int *d_err;
    double *h_pin;    // pinned host scratch (>= 11 * len doubles)
    // one word of mapped host memory: non-zero once a transverse field (Ey, Ez, B) has been uploaded or an
    // electrostatic push launch (particles.hip, ES) has found its belief wrong somewhere; the host reads it without waiting to choose the kernel of later steps
    int *h_esflag, *d_esflag;
    // false on the steps of an opal_sim_step call whose rho nobody can observe (all but the last): the paired fused
    // push then leaves the rho row at zero instead of depositing a diagnostic the next step's clear would erase
    bool want_rho = true;
    std::vector<double> damp_host;
    double cell_x(size_t c) const { return xmin_global + offset + (((double)c) - ((double)L)) * dx; }  // yee.rs:296
};

struct opal_pop {
    int species;
    size_t cap;
    size_t n_upper;  // host upper bound on slots in use
    hipStream_t stream;
    int *d_cell[2];
    double *d_x[2], *d_ux[2], *d_uy[2], *d_uz[2], *d_w[2];
    bool have_alt;
    int cur;
    double *d_prev_x;               // allocated on first unfused advance
    bool prev_valid;
    bool keep_prev;                 // store prev_x in the unfused advance (only a later deposit reads it)
    // optional per-particle extras, any of which may be absent (null):
    //   electrons  chi work y z (diagnostics, level 1)  + tau id (photon emission, level 2)
    //   photons    chi y z id
    // id is a 64-bit integer kept in a double's bits (only ever copied).
    double *d_ex[opal::NEX];
    double *d_ex_alt[opal::NEX];  // re-binning targets, allocated on first use
    unsigned int *d_emit_list;    // slots of this step's emitters (photon emission), emit_list_cap entries
    size_t emit_list_cap;
    bool emit_list_valid;         // the last push left the list of emitters (and d_ctr->n_emit) behind; nothing has
                                  // moved or re-armed a slot since (re-binning, uploads, an emission pass)
    bool sorted_exact;            // slots are exactly in cell order and d_count holds the cell ends (set by a
                                  // re-binning, cleared by anything that moves or adds particles)
    double dead_cost;             // bytes streamed for vacated slots since the last re-binning (photon policy)
    // where each cell's particles ended up at the last re-binning (or sorted upload): slot index one past the last
    // particle of cell c, for c < n_cell_end; host copy, used to tell which slots can reach a subdomain edge
    unsigned int *h_cell_end;
    size_t n_cell_end;
    // this step's split into edge and bulk workgroups (pop_plan_split): blocks [0, split_lo) and [split_hi, all) can
    // hold particles that leave or deposit into the halo cells; the blocks in between cannot
    unsigned split_lo, split_hi;
    hipEvent_t upper_event;       // asynchronous snapshots of n_slots that tighten n_upper (see pop_tighten_upper)
    unsigned int *h_upper;        // pinned
    bool upper_pending;
    size_t added_since_snapshot;
    unsigned long long id_base;   // ids of freshly uploaded particles are id_base + slot (set per subdomain)
    int extras;  // 0 off, 1 diagnostics, 2 + optical depth / id for photon emission
    double z_star, z, a;
    opal::PopCounters *d_ctr;
    opal::PopCounters *h_ctr;  // pinned
    // what the re-binning policy reads: the counters as k_publish_counters last wrote them into mapped host memory,
    // followed by the sequence number of that publication (no copy engine, no event: see maybe_sort)
    opal::PubCounters *h_pub, *d_pub;
    unsigned pub_seq;  // sequence number of the last publication asked for
    opal::PopCounters h_ctr_seen;
    hipEvent_t fetch_event;  // behind the synchronous counter read-backs
    bool ctr_pending;
    int sort_interval;         // 0 never, >0 fixed, <0 adaptive
    int steps_since_sort;
    // the adaptive policy's memory of the previous interval between re-binnings (policy_step): at how many steps the
    // first tile misses were seen, at how many the misses had cost a re-binning; and, from those, the step of this
    // interval at which to re-bin BEFORE the misses come (0: wait for them)
    int miss_onset, last_onset, last_trigger, preempt_at;
    int ppt;                   // particles per thread in the push kernels (uniform chunks: before the first
                               // re-binning, and for slots appended since the last one)
    // Chunk table, rebuilt by every re-binning: the push workgroups' slot ranges, cut both every 256 x 8 slots and
    // every CHUNK_CELLS cells, so that no workgroup's particles span more cells than its LDS tile can serve however
    // sparse the plasma is there.  tab_slot[i] .. tab_slot[i+1] are chunk i's slots; slots >= n_sorted (arrivals
    // appended since) are taken in uniform chunks behind the table.
    unsigned int *d_tab_slot;
    unsigned int *h_tab_slot;  // host copy of d_tab_slot (pop_plan_split)
    size_t tab_cap;
    // (n_sorted on the host: a LOWER bound of the device's ctr->n_sorted -- 0 from a re-binning that did not wait for the
    // device until the count arrives with the next counter snapshot; the pushes read the device's)
    unsigned int n_tab, n_sorted;
    bool sorted_count_pending; // the snapshot in flight (upper_event) carries the slot count a re-binning left: it is n_sorted
    unsigned int count_cells;  // d_count holds the cell ends of this many cells (set by a re-binning)
    size_t added_since_pub;    // room reserved for arrivals since the policy's last publication was asked for (policy_step)
    int last_interval;         // steps between the last two re-binnings (0: none yet): short = a hot population (particles.hip: ROTATE_BELOW)
    bool want_host_maps;       // keep h_cell_end / h_tab_slot current (the overlapped step's edge / bulk split reads them):
                               // re-binnings then wait for the device
    // the step in which the cuts move (sim.cpp: sim_recut): this step's pushes store cells counted from the NEW left cut (old
    // cell reframe_shift) and test leavers against the NEW size; arrivals are unpacked into that frame (particles.hip: PushArgs)
    bool reframe;
    int reframe_shift, reframe_size;
    bool sort_deferred;  // the policy asked for a re-binning during such a step: made once the grid has followed
    bool n_emit_clean;         // d_ctr->n_emit is zero: a kernel of the last step cleared it (PublishBatch::zero)
    int *d_dbg;                // opal_pop_debug_misses
    // Every particle uploaded so far carries the same weight w0 (a uniform plasma loaded with a fixed number per
    // cell): the pushes then take it from their arguments instead of streaming 8 bytes per particle.  Arrivals
    // are compared on the device (d_ctr->w_mixed, which the pushes read); anything else that writes weights into
    // this population clears w_uniform on the host.
    bool w_uniform;
    double w0;
    // Every uy and uz uploaded was zero and only electrostatic launches (particles.hip, ES) have pushed the
    // population since: they keep it so, or raise d_ctr->hot_yz and the grid's flag when they cannot.
    bool cold_yz;
    // opal_absorb's scratch (photon populations): per photon slot what happened and with which electron; event counts
    unsigned char *d_abs_event;
    unsigned int *d_abs_partner, *d_abs_counts;
    size_t abs_cap;
    unsigned int *d_hist;      // pop_cell_histogram's scratch (a re-cut's plan), hist_cap words
    size_t hist_cap;
    unsigned int *d_count;     // sort scratch: ncells + 2
    size_t count_cap;
    double *d_partial;
    size_t partial_cap;
    double dx_last;
};
