// Internal view of opal_link_t (see link.cpp for the message layout).
#pragma once
#include "opal_internal.hpp"

namespace opal {
constexpr size_t LINK_HEADER_BYTES = 64;
constexpr size_t LINK_HALO_BYTES = 80 * sizeof(double);
constexpr size_t LINK_HEAD_BYTES = LINK_HEADER_BYTES + LINK_HALO_BYTES;  // 704
// header words (uint32): [0, 3) leavers per species slot; [4, 7) and [8, 11) particles per species within LINK_ZONE cells
// of the edge the message crosses after this step's push (an upper bound on the leavers LINK_ZONE steps from now) -- two
// sets, used by exchange number parity: the push before exchange X counts into set X & 1, the tail of exchange X reads
// that set (any of its workgroups, at any time: nobody writes it then) and zeroes the other one for the next push
constexpr int LINK_H_COUNT = 0, LINK_H_ZONE = 4;
constexpr int link_h_zone(unsigned exchange) { return LINK_H_ZONE + 4 * (int)(exchange & 1u); }
// peer links (peer.hip) use the bound only for the room the populations reserve, and pay for every write to host memory
// at the end of the tail kernel (~1 us): they count a wider zone and file it with the host every LINK_PEER_EVERY-th
// exchange only -- an entry is then at most LINK_ZONE + LINK_PEER_EVERY - 1 <= LINK_PEER_ZONE exchanges old when it is used
constexpr int LINK_PEER_ZONE = 8, LINK_PEER_EVERY = 4;
constexpr int LINK_ZONE = 3;        // cells; also the lag (in exchanges) with which message sizes follow the bound
constexpr unsigned LINK_ROWS = 64;  // particles per row block
constexpr unsigned LINK_SLACK = 64; // slots beyond the bound (nothing relies on it: the bound itself is rigorous, link_plan)
constexpr int LINK_RING = 8;        // exchanges whose bounds the host keeps

// Row-blocked body: block (b, s) = particles [64 b, 64 b + 64) of species slot s: int32 cell[64] then `nfields`
// columns of double[64]; blocks are ordered b-major, so "the first m blocks of every species" is a prefix.
struct LinkLayout {
    unsigned nslots, blk;  // species slots; bytes per block = 64 * (4 + 8 * nfields)
};
__host__ __device__ inline size_t link_block_off(LinkLayout L, int slot, unsigned i)
{
    return LINK_HEAD_BYTES + ((size_t)(i >> 6) * L.nslots + (unsigned)slot) * L.blk;
}
__host__ __device__ inline int *link_cell(char *msg, LinkLayout L, int slot, unsigned i)
{
    return reinterpret_cast<int *>(msg + link_block_off(L, slot, i)) + (i & 63u);
}
__host__ __device__ inline double *link_field(char *msg, LinkLayout L, int slot, int f, unsigned i)
{
    return reinterpret_cast<double *>(msg + link_block_off(L, slot, i) + 256 + (size_t)f * 512) + (i & 63u);
}
__host__ __device__ inline const int *link_cell(const char *msg, LinkLayout L, int slot, unsigned i)
{
    return reinterpret_cast<const int *>(msg + link_block_off(L, slot, i)) + (i & 63u);
}
__host__ __device__ inline const double *link_field(const char *msg, LinkLayout L, int slot, int f, unsigned i)
{
    return reinterpret_cast<const double *>(msg + link_block_off(L, slot, i) + 256 + (size_t)f * 512) + (i & 63u);
}

// what one exchange leaves for the host (mapped pinned memory, written by the unpack kernel): the zone counts of
// this subdomain's two outgoing messages and of the two it received, and the exchange they belong to
struct LinkBounds {
    unsigned mine[2][3];    // [side][species slot]: my particles near my left / right edge
    unsigned theirs[2][3];  // [side]: the left / right neighbour's particles near the edge we share
    unsigned pad[3];
    volatile unsigned seq;  // exchange number + 1, written last
};
}  // namespace opal

struct opal_link {
    int rank, size;
    bool device_buffers;
    bool external;  // buffers the transport sees belong to the caller
    bool counts_clean;  // the send messages' leaver counts are known to be zero (see link_begin)
    size_t cap;     // particles per species per direction the link was created with room for (the smaller of cap_side)
    size_t cap_side[2];   // ... and what the buffers (and, peer links, the mailboxes) of the left / right edge have room for NOW:
                          // an edge grows when the bound on what can cross it exceeds its room (link_plan, link_grow)
    size_t nbytes_side[2];  // bytes of a message of that edge with every slot in use
    bool growable;        // edges grow instead of overflowing (default: peer links; opal_link_set_growth)
    unsigned grows;       // how often an edge of this link has grown (diagnostic)
    int nslots;     // species blocks per message: 2 (electrons, ions) or 3 (+ photons)
    int nfields;    // doubles per particle: 6 (x ux uy uz w prev_x), 12 (+ chi work y z tau id) or 15 (+ tau_abs tau_st birth)
    opal::LinkLayout layout;
    size_t nbytes;  // a message with all `cap` slots (the larger edge's: max of nbytes_side)
    opal_exchange_fn exchange;
    opal_exchange2_fn exchange2;  // optional: per-edge sizes (see link_exchange_particles)
    void *user;
    void (*user_free)(void *);  // releases `user` with the link (native transports), or null
    // optional: tells a stream-ordered transport which stream the next exchanges are ordered on; null = it only
    // knows the stream it was created with (e.g. a host-language callback tied to its framework's stream)
    void (*set_stream)(void *user, hipStream_t st);
    hipStream_t stream;
    char *d_send_left, *d_send_right, *d_recv_left, *d_recv_right;  // device side (kernels pack / unpack here)
    char *h_send_left, *h_send_right, *h_recv_left, *h_recv_right;  // host staging when device_buffers == 0
    // message sizes that follow the particles (link.cpp): bounds of the last LINK_RING exchanges, written by the device
    opal::LinkBounds *h_bounds, *d_bounds;  // the same mapped pinned memory, host and device view
    unsigned long long *d_tail_words;       // 8 device words: where the workgroups that share an unpack agree on its first slot
    unsigned xseq;           // whole-message exchanges so far
    unsigned ring_mask[opal::LINK_RING];  // which species' pushes counted into exchange e's headers, at e % LINK_RING (link_plan)
    unsigned send_cap[2];    // slots (a multiple of 64) the next exchange ships per species: left / right edge
    unsigned plan_cap[2];    // bound on the particles per species that can cross each edge in the next exchange (<= send_cap):
                             // what the populations reserve room for (a peer link ships whatever leaves, up to `cap`)
    bool bounded;            // sizes follow the bounds (needs exchange2); else every exchange ships all `cap` slots
    size_t last_bytes[2];    // what the last whole-message exchange moved per edge (diagnostic)
    // After the cuts have moved (sim.cpp: sim_recut) the ring's entries count particles near where the cuts WERE.  Until
    // exchanges made with the new cuts have filed theirs, the bound is what stood within the zone of each new cut when it
    // was made (hold_need, from the histogram the re-cut was planned with): exchange hold_from + j comes j + 1 pushes after
    // the re-cut, so that count covers j < zone; entries of exchanges before hold_from are never used again.
    bool hold_on;
    unsigned hold_from, hold_need[2];
    bool hold_side[2];  // the cut on that side moved: a side whose cut stayed goes on by what was counted there (before or after)
    unsigned zone_scale;   // 0 / 1: the zone as it is; 2: twice as wide -- while cuts drift a cell at a time (sim.cpp) a leaver of
                           // exchange X was, k exchanges earlier, within 2 k cells of where the cut stood THEN
    unsigned burst_parts;  // != 0: the next tail kernel shares every (species, side) among this many workgroups (the step in
                           // which a re-cut hands hundreds of thousands of particles over: sim.cpp; back to 0 afterwards)
    // peer-mapped mailboxes (peer.hip, peer_device.hpp): the kernels of the step move the messages themselves
    struct opal_peer *peer;  // null for transports that are called (RCCL, torch.distributed, host callbacks)
};

namespace opal {
char *link_send(opal_link_t *l, int side);
char *link_recv(opal_link_t *l, int side);
// species_mask: bit s = species slot s (0 electrons, 1 ions, 2 photons) is pushed before the coming exchange
int link_begin(opal_link_t *l, hipStream_t st, unsigned species_mask);
// decides how many slots per species the coming exchange ships on each edge (send_cap), from the bounds the device
// reported at least LINK_ZONE exchanges ago; call before the push that packs the leavers
int link_plan(opal_link_t *l, unsigned species_mask);
size_t link_prefix_bytes(const opal_link_t *l, unsigned slots);
// halo-only exchange (the first LINK_HEAD_BYTES of each message)
int link_exchange_halo(opal_link_t *l, bool to_left, bool to_right, hipStream_t st);
// whole messages, each edge as long as link_plan decided
int link_exchange_particles(opal_link_t *l, bool to_left, bool to_right, hipStream_t st);
bool link_can_use_stream(const opal_link_t *l, hipStream_t st);
// peer.hip
struct PeerArgs;
struct HaloSrc;
bool link_is_peer(const opal_link_t *l);
// the arguments with which a kernel of the step makes the NEXT exchange itself (advances the link's sequence); halo
// = where the halo is packed from, or null when the send messages already hold it
int link_peer_next(opal_link_t *l, bool to_left, bool to_right, bool head_only, const HaloSrc *halo, PeerArgs &a);
void link_peer_free(opal_link_t *l);
// Makes room for new_cap[side] particles per species on the left / right edge: new send / recv buffers, and on a peer link
// new mailboxes, whose handle the two ends of an edge hand each other through a control line in the OLD mailboxes
// (peer.hip).  Both ends of an edge call it for the same exchange with the same capacity (link_plan decides from numbers
// both hold).  Drains the device.
int link_grow(opal_link_t *l, const size_t new_cap[2]);  // (0: that edge stays as it is)
// the capacity an edge that must hold `slots` particles per species grows to (a power of two with a quarter to spare; a
// peer link's stays below what its neighbour can map)
size_t link_room_for(const opal_link_t *l, size_t slots);
int link_peer_grow_begin(opal_link_t *l, int side, size_t new_cap, void **state);
int link_peer_grow_finish(opal_link_t *l, int side, void *state);  // (releases `state` whatever it returns)
void link_peer_grow_abort(opal_link_t *l, void *state);            // a growth begun that will not be finished
// OPAL_ERR_LINK if a wait for a neighbour's message has ever timed out on this link (peer links; OPAL_OK for the others).
// Synchronises with the link's stream: called where the host waits for the device anyway.
int link_peer_check(opal_link_t *l);
// the cuts have just moved: need[side] particles per species stand within the zone of the left / right cut now (see hold_on)
// moved[side]: whether that cut changed at all (null: both did)
void link_hold(opal_link_t *l, const unsigned need[2], const bool moved[2] = nullptr);
// cells either side of a cut within which the pushes count particles into the message headers (what link_plan's bound covers)
int link_zone_cells(const opal_link_t *l);
}  // namespace opal
