// Neighbour exchange between subdomains (stands in for the reference's MPI
// point-to-point traffic: yee.rs:351-409 for the halo, particle/mod.rs:303-354
// for migrating particles).  The library packs one message per direction; the
// transport (RCCL, gloo, ...) is the callback the host supplied.
//
// Message layout (bytes):
//   [0, 64)            header (uint32 words): [0, 3) leavers per species slot (0 electrons, 1 ions, 2 photons);
//                      [4, 7) particles per species within LINK_ZONE cells of the edge this message crosses,
//                      counted by the push that packed it
//   [64, 704)          halo: 8 cells x (rho jx jy jz Ex Ey Ez Bx By Bz) doubles  (yee.rs:353-354)
//   then row blocks of 64 particles, ordered (block b, species slot s) b-major:
//       int32  cell[64]           re-based cell index in the receiver's frame (mod.rs:291-293, 366-368)
//       double x[64] ux[64] uy[64] uz[64] w[64] prev_x[64]
//       double chi[64] work[64] y[64] z[64] tau[64] id[64]      links created with >= 12 fields
//       double tau_abs[64] tau_st[64] birth[64]                 links created with 15 fields (photon absorption)
// so that the first m blocks of every species are a PREFIX of the message: an exchange ships
// 704 + m S blk bytes per edge, m chosen per edge and step.
//
// How m follows the particles without the host waiting for the device.  A particle moves less than a cell per
// step, so the leavers of step k through an edge are among the particles that stood within LINK_ZONE cells of
// it -- on either side -- after step k - LINK_ZONE.  Every push counts those into its message's header; the counts
// travel with the message, and the unpack kernel copies both ends' counts of every exchange into mapped host memory.
// When exchange X is planned the host reads what the most recent exchange <= X - LINK_ZONE that counted each species
// left there (long since written: the host merely cannot run more than LINK_ZONE exchanges ahead of the device) and
// sizes each edge to the SUM of the two ends' counts plus LINK_SLACK (link_plan has the argument).  Both ends of an
// edge hold the same numbers, so they agree on the size without talking; a front that reaches a cut is seen LINK_ZONE
// cells before it arrives.  Transports that cannot take a size per edge (opal_exchange_fn) keep shipping all
// `particle_capacity` slots; so do links that carry stimulated emission.  A peer link's messages need no sizing (the
// sending kernel reads the leaver counts itself), but the same bound is kept for it as plan_cap: it is the room the
// populations reserve for a step's arrivals.
// A halo-only exchange (opal_grid_synchronize) moves the first 704 bytes.
// What a subdomain sends left arrives as its left neighbour's "recv_right".
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "link.hpp"

using namespace opal;

namespace opal {

char *link_send(opal_link_t *l, int side) { return side == 0 ? l->d_send_left : l->d_send_right; }
char *link_recv(opal_link_t *l, int side) { return side == 0 ? l->d_recv_left : l->d_recv_right; }

size_t link_prefix_bytes(const opal_link_t *l, unsigned slots)
{
    return LINK_HEAD_BYTES + (size_t)(slots / LINK_ROWS) * l->layout.nslots * l->layout.blk;
}

// Leaver counts of the send messages start a step at zero.  The kernel that unpacks a step's arrivals also
// re-zeroes them (they have been shipped by then) and marks the link clean, so only the first step -- or one
// after an aborted exchange -- pays for the two memsets.
int link_begin(opal_link_t *l, hipStream_t st, unsigned species_mask)
{
    if (!l->counts_clean) {
        OPAL_HIP(hipMemsetAsync(l->d_send_left, 0, LINK_HEADER_BYTES, st));
        OPAL_HIP(hipMemsetAsync(l->d_send_right, 0, LINK_HEADER_BYTES, st));
    }
    l->counts_clean = false;  // a push is about to count leavers into them
    return link_plan(l, species_mask);
}

// How many slots per species the exchange about to be made (number xseq) ships on each edge, for the species in
// `species_mask` (bit s = species slot s is pushed, and counted, before this exchange).
//
// The bound.  A particle moves less than a cell per push.  A leaver of species s through an edge at exchange X was,
// LINK_ZONE pushes of s earlier at the most, within LINK_ZONE cells of that edge ON EITHER SIDE of it (it may have
// crossed into this subdomain meanwhile and be on its way back): it was counted by this subdomain's push (`mine`) or by
// the neighbour's (`theirs`) -- the pushes count the particles that stay within the zone AND the leavers through that
// edge (a leaver of exchange e arrives after the receiver's push has counted, and stands within a cell of the cut).
// So need(s) = mine + theirs of the most recent exchange e_s <= X - LINK_ZONE that counted s -- between e_s and X the species is pushed at most LINK_ZONE times, whatever mix of per-species and combined
// exchanges the caller makes (the ring remembers which species each exchange counted).  Photons born since the count
// (photon emission: links with >= 12 fields) come from electrons that were within the same LINK_ZONE cells, one photon
// per electron and step at most: need(photons) += LINK_ZONE x need(electrons).  Stimulated emission (15 fields) has
// no such bound (several photons may copy themselves off one electron): those links ship every slot.  No usable entry
// in the ring (start-up; more than LINK_RING - LINK_ZONE exchanges since the species was last counted): every slot.
// A peer link (whose messages need no sizing: plan_cap only, the room populations reserve) counts LINK_PEER_ZONE cells
// and files every LINK_PEER_EVERY-th exchange: the same argument with an entry up to LINK_PEER_ZONE exchanges old (a
// caller that exchanges species by species over such a link finds entries for the species of those exchanges only; the
// others reserve the link's capacity).
size_t link_room_for(const opal_link_t *l, size_t slots)
{
    size_t want = slots + slots / 4, cap2 = 1024;
    while (cap2 < want) cap2 *= 2;
    if (l->peer) {
        // An edge's two mailboxes are one IPC-shared allocation, and one of 2 GiB or more could not be mapped by
        // the neighbour (hipIpcOpenMemHandle never returned for 2.5 GB where 1.26 GB had taken milliseconds:
        // round 4, two processes on one MI355X).  An edge stops growing below that; a bound beyond it -- the bound
        // counts what COULD cross, newborn photons of every electron near the cut included -- is then not
        // covered by room, and a real overflow would be the error it always was.
        const size_t per_slot = (size_t)l->layout.nslots * l->layout.blk / LINK_ROWS;
        const size_t most = ((((size_t)1 << 31) - 65536) / 2 - LINK_HEAD_BYTES) / per_slot;
        while (cap2 > most) cap2 /= 2;
    }
    return cap2;
}

int link_plan(opal_link_t *l, unsigned species_mask)
{
    unsigned full[2] = {(unsigned)l->cap_side[0], (unsigned)l->cap_side[1]};
    l->send_cap[0] = l->plan_cap[0] = full[0];
    l->send_cap[1] = l->plan_cap[1] = full[1];
    const unsigned X = l->xseq;
    l->ring_mask[X % LINK_RING] = species_mask;
    // A transport that is told a size per edge ships the bound; a peer link ships what leaves (its kernels read the
    // counts), but the populations still need the bound -- it is what they reserve room for before each step's arrivals,
    // and reserving the link's whole capacity every step would have a link sized for a pile-up (128 cells' worth of
    // particles) blow the host's bound on the slots in use up by millions per step.
    const bool sizes_messages = l->bounded && l->exchange2 && !link_is_peer(l);
    // (a link that carries stimulated emission -- 15 fields -- ships every slot: several photons may copy themselves off one
    // electron, so the count below is no bound for it.  It is still the best estimate of what stands near the cut, and an
    // edge of such a link grows by it like any other instead of staying at its creation size through a pile-up.)
    const bool every_slot = l->nfields >= 15;
    if ((!sizes_messages && !link_is_peer(l) && !(every_slot && l->growable)) || !l->h_bounds) return OPAL_OK;
    if (X < (unsigned)LINK_ZONE && !l->hold_on) return OPAL_OK;
    unsigned need[2] = {0, 0};
    // (a peer link files every LINK_PEER_EVERY-th exchange only and counts LINK_PEER_ZONE cells: see link.hpp)
    const bool peer = link_is_peer(l);
    // (how old an entry may be, in exchanges -- the counted zone itself is zone_scale times as wide while cuts drift: link.hpp)
    const unsigned zone_cells = peer ? (unsigned)LINK_PEER_ZONE : (unsigned)LINK_ZONE;
    auto entry_of = [&](int sp, const LinkBounds **out, bool before_the_hold) -> int {  // most recent usable entry that counted species sp
        *out = nullptr;
        unsigned e = X - LINK_ZONE;
        if (peer) e -= e % LINK_PEER_EVERY;
        for (;; e -= peer ? LINK_PEER_EVERY : 1) {
            if (l->hold_on && e < l->hold_from && !before_the_hold) break;  // counted near where the cuts were before they moved
            // overwritten since (or about to be).  (An old entry is as good as a recent one for a called transport: the
            // exchanges after it that did not count the species did not push it either, so at most LINK_ZONE pushes of
            // it lie between the entry and X.  A peer link skips entries whatever they counted, so age itself bounds.)
            if (X - e >= (unsigned)LINK_RING || (peer && X - e > zone_cells)) break;
            if (l->ring_mask[e % LINK_RING] & (1u << sp)) {
                const LinkBounds *b = &l->h_bounds[e % LINK_RING];
                // Written by the unpack kernel of exchange e, at least LINK_ZONE exchanges ago: normally there already.
                // The wait must not have a timing-dependent way out -- a subdomain that gave up and shipped another
                // size than its neighbour expects would hang the transport -- so it only ends in the entry or an error.
                const auto t0 = std::chrono::steady_clock::now();
                unsigned long spins = 0;
                while (__atomic_load_n(&b->seq, __ATOMIC_ACQUIRE) != e + 1) {
                    if ((++spins & 0xfffff) == 0 &&
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                        set_error("link: the particle bounds of exchange %u never arrived (every whole-message exchange must "
                                  "be followed by the unpack of its arrivals on the link's stream)", e);
                        return OPAL_ERR_LINK;
                    }
                }
                *out = b;
                return OPAL_OK;
            }
            if (e < (peer ? (unsigned)LINK_PEER_EVERY : 1u)) break;
        }
        return OPAL_OK;
    };
    const LinkBounds *be = nullptr;
    const bool qed = l->nfields >= 12 && l->nslots >= 3;
    const bool held = l->hold_on && X - l->hold_from < zone_cells;  // (X >= hold_from: xseq only grows)
    // the cuts moved fewer than zone_cells exchanges ago: what stood near them then (link_hold) -- for a side whose cut did
    // move.  The cut of the other side is where it was: what was counted there, before the hold or after, stands (a rank far
    // from the front would otherwise reserve, and grow its edges to, the histogram's generous figure at every re-cut of
    // somebody else's cuts -- and re-bin three times as often for the room: the 6-rank rehearsal of configs[3]).
    const bool side_held[2] = {held && l->hold_side[0], held && l->hold_side[1]};
    bool unknown[2] = {false, false};
    for (int side = 0; side < 2; side++)
        if (side_held[side]) need[side] = l->hold_need[side];
    if (!(side_held[0] && side_held[1])) {
        if (qed && (species_mask & 4u))
            if (int rc = entry_of(0, &be, held)) return rc;
        for (int sp = 0; sp < l->nslots; sp++) {
            if (!(species_mask & (1u << sp))) continue;
            const LinkBounds *b;
            if (int rc = entry_of(sp, &b, held)) return rc;
            if (!b || (sp == 2 && qed && !be)) {  // no history to go by: every slot
                if (!held) return OPAL_OK;
                unknown[0] = !side_held[0];
                unknown[1] = !side_held[1];
                break;
            }
            for (int side = 0; side < 2; side++) {
                if (side_held[side]) continue;
                unsigned n = b->mine[side][sp] + b->theirs[side][sp];
                if (sp == 2 && qed) n += zone_cells * (be->mine[side][0] + be->theirs[side][0]);  // (one photon per electron and exchange)
                need[side] = need[side] > n ? need[side] : n;
            }
        }
    }
    unsigned slots[2];
    size_t grow_to[2] = {0, 0};
    for (int side = 0; side < 2; side++) {
        slots[side] = unknown[side] ? full[side] : (need[side] + LINK_SLACK + LINK_ROWS - 1) / LINK_ROWS * LINK_ROWS;
        if (slots[side] > full[side] && l->growable && need[side]) {
            // More particles stand within reach of this cut than its messages have room for: the edge grows before the
            // push that could overflow it is enqueued (the reference's receive_vec takes messages of any size,
            // particle/mod.rs:308-354).  The other end of the edge reads the same numbers for the same exchange and
            // grows to the same capacity; a pile-up that moves on leaves the room behind (no shrinking: the memory is
            // what the densest moment of the run needed at this cut).
            const size_t cap2 = link_room_for(l, slots[side]);
            if (cap2 > full[side]) grow_to[side] = cap2;
        }
    }
    if (grow_to[0] || grow_to[1]) {
        if (getenv("OPAL_LINK_DEBUG"))
            fprintf(stderr, "[link] rank %d exchange %u: need %u / %u, room %u / %u -> grow to %zu / %zu\n", l->rank, X, need[0],
                    need[1], full[0], full[1], grow_to[0], grow_to[1]);
        if (int rc = link_grow(l, grow_to)) return rc;
        if (getenv("OPAL_LINK_DEBUG")) fprintf(stderr, "[link] rank %d exchange %u: grown\n", l->rank, X);
        full[0] = (unsigned)l->cap_side[0];
        full[1] = (unsigned)l->cap_side[1];
    }
    for (int side = 0; side < 2; side++) {
        l->plan_cap[side] = (slots[side] < full[side] && !every_slot) ? slots[side] : full[side];
        l->send_cap[side] = (sizes_messages && !every_slot) ? l->plan_cap[side] : full[side];
    }
    return OPAL_OK;
}

void link_hold(opal_link_t *l, const unsigned need[2], const bool moved[2])
{
    l->hold_on = true;
    l->hold_from = l->xseq;
    for (int side = 0; side < 2; side++) {
        l->hold_need[side] = need[side];
        l->hold_side[side] = moved && !getenv("OPAL_HOLD_BOTH_SIDES") ? moved[side] : true;  // (the knob: a diagnostic)
    }
}

int link_zone_cells(const opal_link_t *l) { return (link_is_peer(l) ? LINK_PEER_ZONE : LINK_ZONE) * (l->zone_scale > 1 ? (int)l->zone_scale : 1); }

int link_grow(opal_link_t *l, const size_t new_cap_in[2])
{
    size_t new_cap[2];
    bool any = false;
    for (int side = 0; side < 2; side++) {
        new_cap[side] = (new_cap_in[side] + 63) / 64 * 64;
        if (new_cap[side] <= l->cap_side[side]) new_cap[side] = 0;
        any = any || new_cap[side];
    }
    if (!any) return OPAL_OK;
    if (l->external) {
        set_error("link: an edge needs room for %zu particles per species, but the link's buffers belong to the caller "
                  "(opal_link_set_buffers)", new_cap[0] > new_cap[1] ? new_cap[0] : new_cap[1]);
        return OPAL_ERR_CAPACITY;
    }
    if (l->peer && !link_is_peer(l)) {
        // (the mailboxes of an unconnected peer link keep their creation stride, and opal_link_peer_connect takes the
        // neighbours' offsets from it: buffers grown here would outrun them)
        set_error("link: a peer link grows only once it is connected (opal_link_peer_connect)");
        return OPAL_ERR_LINK;
    }
    OPAL_HIP(hipDeviceSynchronize());  // nothing in flight reads or writes the buffers about to go
    void *state[2] = {nullptr, nullptr};
    // whatever fails from here on, the mailboxes allocated for edges that did not finish growing are let go
    auto body = [&]() -> int {
        if (l->peer)  // every record is on its way before anybody waits for one (peer.hip)
            for (int side = 0; side < 2; side++)
                if (new_cap[side])
                    if (int rc = link_peer_grow_begin(l, side, new_cap[side], &state[side])) return rc;
        for (int side = 0; side < 2; side++) {
            if (!new_cap[side]) continue;
            const size_t nb = link_prefix_bytes(l, (unsigned)new_cap[side]);
            char **dev[2] = {side == 0 ? &l->d_send_left : &l->d_send_right, side == 0 ? &l->d_recv_left : &l->d_recv_right};
            for (auto d : dev) {
                char *fresh = nullptr;
                OPAL_HIP(hipMalloc((void **)&fresh, nb));
                hipError_t e = hipMemset(fresh, 0, nb);
                // (the header of a send message carries the zone counts of the exchange before: kept)
                if (e == hipSuccess) e = hipMemcpy(fresh, *d, LINK_HEAD_BYTES, hipMemcpyDeviceToDevice);
                if (e != hipSuccess) {
                    hipFree(fresh);
                    OPAL_HIP(e);
                }
                OPAL_HIP(hipFree(*d));
                *d = fresh;
            }
            if (!l->device_buffers) {
                char **host[2] = {side == 0 ? &l->h_send_left : &l->h_send_right, side == 0 ? &l->h_recv_left : &l->h_recv_right};
                for (auto h : host) {
                    OPAL_HIP(hipHostFree(*h));
                    *h = nullptr;
                    OPAL_HIP(hipHostMalloc((void **)h, nb));
                    memset(*h, 0, nb);
                }
            }
            if (l->peer) {
                void *st = state[side];
                state[side] = nullptr;  // (link_peer_grow_finish owns it from here, success or not)
                if (int rc = link_peer_grow_finish(l, side, st)) return rc;
            }
            l->cap_side[side] = new_cap[side];
            l->nbytes_side[side] = nb;
            l->grows++;
        }
        return OPAL_OK;
    };
    const int rc = body();
    for (int side = 0; side < 2; side++)
        if (state[side]) link_peer_grow_abort(l, state[side]);
    l->nbytes = l->nbytes_side[0] > l->nbytes_side[1] ? l->nbytes_side[0] : l->nbytes_side[1];
    return rc;
}

// whether exchanges may be issued on `st` rather than on the stream the link was created with: host-staged
// transports are synchronous (the library brackets them with copies on `st`), native ones follow set_stream
bool link_can_use_stream(const opal_link_t *l, hipStream_t st)
{
    return st == l->stream || !l->device_buffers || l->set_stream;
}

// moves the first n_left / n_right bytes of the send buffers to the neighbours and fills the recv buffers
static int link_move(opal_link_t *l, size_t n_left, size_t n_right, bool to_left, bool to_right, hipStream_t st)
{
    if (l->set_stream) l->set_stream(l->user, st);
    if (n_left > l->nbytes_side[0] || n_right > l->nbytes_side[1]) {
        set_error("link exchange: %zu / %zu bytes exceed the edges' message sizes %zu / %zu", n_left, n_right,
                  l->nbytes_side[0], l->nbytes_side[1]);
        return OPAL_ERR_ARG;
    }
    if (!l->device_buffers) {
        if (to_left) OPAL_HIP(hipMemcpyAsync(l->h_send_left, l->d_send_left, n_left, hipMemcpyDeviceToHost, st));
        if (to_right) OPAL_HIP(hipMemcpyAsync(l->h_send_right, l->d_send_right, n_right, hipMemcpyDeviceToHost, st));
        OPAL_HIP(hipStreamSynchronize(st));
    }
    char *sl = l->device_buffers ? l->d_send_left : l->h_send_left;
    char *sr = l->device_buffers ? l->d_send_right : l->h_send_right;
    char *rl = l->device_buffers ? l->d_recv_left : l->h_recv_left;
    char *rr = l->device_buffers ? l->d_recv_right : l->h_recv_right;
    int rc;
    if (l->exchange2)
        rc = l->exchange2(l->user, to_left ? sl : nullptr, to_right ? sr : nullptr, to_left ? rl : nullptr,
                          to_right ? rr : nullptr, n_left, n_right);
    else
        rc = l->exchange(l->user, to_left ? sl : nullptr, to_right ? sr : nullptr, to_left ? rl : nullptr,
                         to_right ? rr : nullptr, n_left > n_right ? n_left : n_right);
    if (rc) {
        set_error("exchange callback failed with %d", rc);
        return OPAL_ERR_LINK;
    }
    if (!l->device_buffers) {
        if (to_left) OPAL_HIP(hipMemcpyAsync(l->d_recv_left, l->h_recv_left, n_left, hipMemcpyHostToDevice, st));
        if (to_right) OPAL_HIP(hipMemcpyAsync(l->d_recv_right, l->h_recv_right, n_right, hipMemcpyHostToDevice, st));
    }
    return OPAL_OK;
}

int link_exchange_halo(opal_link_t *l, bool to_left, bool to_right, hipStream_t st)
{
    return link_move(l, LINK_HEAD_BYTES, LINK_HEAD_BYTES, to_left, to_right, st);
}

int link_exchange_particles(opal_link_t *l, bool to_left, bool to_right, hipStream_t st)
{
    // without a per-edge transport both edges (and every subdomain) must use one size: the full one
    const bool per_edge = l->exchange2 != nullptr;
    const size_t n_left = per_edge ? link_prefix_bytes(l, l->send_cap[0]) : l->nbytes_side[0];
    const size_t n_right = per_edge ? link_prefix_bytes(l, l->send_cap[1]) : l->nbytes_side[1];
    l->last_bytes[0] = to_left ? n_left : 0;
    l->last_bytes[1] = to_right ? n_right : 0;
    int rc = link_move(l, n_left, n_right, to_left, to_right, st);
    l->xseq++;
    return rc;
}

}  // namespace opal

extern "C" {

int opal_link_create(opal_link_t **out, int rank, int size, int device_buffers, size_t particle_capacity,
                     opal_exchange_fn exchange, void *user, void *hip_stream)
{
    return opal_link_create_layout(out, rank, size, device_buffers, particle_capacity, 2, 6, exchange, user, hip_stream);
}

int opal_link_create_layout(opal_link_t **out, int rank, int size, int device_buffers, size_t particle_capacity,
                            int species_slots, int fields_per_particle, opal_exchange_fn exchange, void *user,
                            void *hip_stream)
{
    if (!out || !exchange || size < 1 || rank < 0 || rank >= size || species_slots < 2 || species_slots > 3
        || (fields_per_particle != 6 && fields_per_particle != 12 && fields_per_particle != 15)) {
        set_error("opal_link_create: bad arguments");
        return OPAL_ERR_ARG;
    }
    if (particle_capacity == 0) particle_capacity = 4096;
    particle_capacity = (particle_capacity + 63) / 64 * 64;
    opal_link_t *l = new opal_link_t();
    memset((void *)l, 0, sizeof *l);
    l->rank = rank;
    l->size = size;
    l->device_buffers = device_buffers != 0;
    l->cap = particle_capacity;
    l->exchange = exchange;
    l->user = user;
    l->stream = (hipStream_t)hip_stream;
    l->nslots = species_slots;
    l->nfields = fields_per_particle;
    l->layout.nslots = (unsigned)species_slots;
    l->layout.blk = LINK_ROWS * (unsigned)(sizeof(int32_t) + (size_t)fields_per_particle * sizeof(double));
    l->nbytes = link_prefix_bytes(l, (unsigned)particle_capacity);
    l->cap_side[0] = l->cap_side[1] = particle_capacity;
    l->nbytes_side[0] = l->nbytes_side[1] = l->nbytes;
    l->send_cap[0] = l->send_cap[1] = (unsigned)particle_capacity;
    l->plan_cap[0] = l->plan_cap[1] = (unsigned)particle_capacity;
    l->bounded = true;  // takes effect once a per-edge transport is set (opal_link_set_exchange2, native links)
    OPAL_HIP(hipHostMalloc((void **)&l->h_bounds, sizeof(LinkBounds) * LINK_RING, hipHostMallocMapped | hipHostMallocCoherent));
    memset((void *)l->h_bounds, 0, sizeof(LinkBounds) * LINK_RING);
    OPAL_HIP(hipHostGetDevicePointer((void **)&l->d_bounds, (void *)l->h_bounds, 0));
    OPAL_HIP(hipMalloc((void **)&l->d_tail_words, 8 * sizeof(unsigned long long)));
    OPAL_HIP(hipMemset(l->d_tail_words, 0, 8 * sizeof(unsigned long long)));
    char **dev[4] = {&l->d_send_left, &l->d_send_right, &l->d_recv_left, &l->d_recv_right};
    for (auto d : dev) {
        OPAL_HIP(hipMalloc((void **)d, l->nbytes));
        OPAL_HIP(hipMemset(*d, 0, l->nbytes));
    }
    if (!l->device_buffers) {
        char **host[4] = {&l->h_send_left, &l->h_send_right, &l->h_recv_left, &l->h_recv_right};
        for (auto h : host) {
            OPAL_HIP(hipHostMalloc((void **)h, l->nbytes));
            memset(*h, 0, l->nbytes);
        }
    }
    *out = l;
    return OPAL_OK;
}

int opal_link_free(opal_link_t *l)
{
    if (!l) return OPAL_OK;
    if (l->user_free) l->user_free(l->user);
    link_peer_free(l);
    if (!(l->external && l->device_buffers)) {
        hipFree(l->d_send_left); hipFree(l->d_send_right); hipFree(l->d_recv_left); hipFree(l->d_recv_right);
    }
    if (!l->device_buffers && !l->external) {
        hipHostFree(l->h_send_left); hipHostFree(l->h_send_right); hipHostFree(l->h_recv_left); hipHostFree(l->h_recv_right);
    }
    if (l->h_bounds) hipHostFree((void *)l->h_bounds);
    if (l->d_tail_words) hipFree(l->d_tail_words);
    delete l;
    return OPAL_OK;
}

int opal_link_set_exchange2(opal_link_t *l, opal_exchange2_fn exchange2)
{
    if (!l) {
        set_error("opal_link_set_exchange2: null link");
        return OPAL_ERR_ARG;
    }
    l->exchange2 = exchange2;
    return OPAL_OK;
}

int opal_link_set_growth(opal_link_t *l, int on)
{
    if (!l) {
        set_error("opal_link_set_growth: null link");
        return OPAL_ERR_ARG;
    }
    if (on && l->external) {
        set_error("opal_link_set_growth: the link's buffers belong to the caller (opal_link_set_buffers)");
        return OPAL_ERR_ARG;
    }
    l->growable = on != 0;
    return OPAL_OK;
}

int opal_link_growth(const opal_link_t *l, size_t *cap_left, size_t *cap_right, unsigned *grows)
{
    if (!l) {
        set_error("opal_link_growth: null link");
        return OPAL_ERR_ARG;
    }
    if (cap_left) *cap_left = l->cap_side[0];
    if (cap_right) *cap_right = l->cap_side[1];
    if (grows) *grows = l->grows;
    return OPAL_OK;
}

int opal_link_set_bounded(opal_link_t *l, int on)
{
    if (!l) {
        set_error("opal_link_set_bounded: null link");
        return OPAL_ERR_ARG;
    }
    l->bounded = on != 0;
    return OPAL_OK;
}

int opal_link_last_exchange_bytes(const opal_link_t *l, size_t *left, size_t *right)
{
    if (!l) {
        set_error("opal_link_last_exchange_bytes: null link");
        return OPAL_ERR_ARG;
    }
    if (left) *left = l->last_bytes[0];
    if (right) *right = l->last_bytes[1];
    return OPAL_OK;
}

size_t opal_link_message_bytes(const opal_link_t *l) { return l ? l->nbytes : 0; }
size_t opal_link_halo_bytes(const opal_link_t *l) { return l ? LINK_HEAD_BYTES : 0; }
size_t opal_link_particle_capacity(const opal_link_t *l) { return l ? l->cap : 0; }

// Replaces the buffers the transport sees by caller-owned ones of at least opal_link_message_bytes()
// (device memory when device_buffers != 0, host memory otherwise), e.g. torch tensors, so that the
// transport can name them in its own terms.  Order: send_left, send_right, recv_left, recv_right.
int opal_link_set_buffers(opal_link_t *l, void *send_left, void *send_right, void *recv_left, void *recv_right)
{
    if (!l || !send_left || !send_right || !recv_left || !recv_right) {
        set_error("opal_link_set_buffers: bad arguments");
        return OPAL_ERR_ARG;
    }
    if (l->external) {
        set_error("opal_link_set_buffers: buffers already replaced");
        return OPAL_ERR_ARG;
    }
    if (l->device_buffers) {
        hipFree(l->d_send_left); hipFree(l->d_send_right); hipFree(l->d_recv_left); hipFree(l->d_recv_right);
        l->d_send_left = (char *)send_left; l->d_send_right = (char *)send_right;
        l->d_recv_left = (char *)recv_left; l->d_recv_right = (char *)recv_right;
    } else {
        hipHostFree(l->h_send_left); hipHostFree(l->h_send_right); hipHostFree(l->h_recv_left); hipHostFree(l->h_recv_right);
        l->h_send_left = (char *)send_left; l->h_send_right = (char *)send_right;
        l->h_recv_left = (char *)recv_left; l->h_recv_right = (char *)recv_right;
    }
    l->external = true;
    l->counts_clean = false;
    return OPAL_OK;
}

void *opal_link_buffer(opal_link_t *l, int which)
{
    if (!l) return nullptr;
    if (l->device_buffers) {
        char *b[4] = {l->d_send_left, l->d_send_right, l->d_recv_left, l->d_recv_right};
        return which >= 0 && which < 4 ? b[which] : nullptr;
    }
    char *b[4] = {l->h_send_left, l->h_send_right, l->h_recv_left, l->h_recv_right};
    return which >= 0 && which < 4 ? b[which] : nullptr;
}

}  // extern "C"
