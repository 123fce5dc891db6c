// Host side of the peer-mapped mailboxes (peer_device.hpp) and the stand-alone exchange kernel that lets every
// existing exchange point (opal_grid_synchronize's halo-only exchange, the separate-operator step's whole
// messages) run over them unchanged; the fused linked step does the same inside its own tail kernel
// (particles.hip: k_link_tail).
#include <unistd.h>
#include "peer_device.hpp"

#include <chrono>
#include <cstring>
#include <vector>

using namespace opal;

struct opal_peer {
    // my mailboxes for the messages FROM the left / right neighbour: a control line (PEER_CTL_BYTES), then parity 0 | 1,
    // each stride[side] bytes.  At creation both sides live in one allocation (base0: one IPC handle to hand round); an
    // edge that has grown has an allocation of its own.
    char *box_mem[2] = {nullptr, nullptr};
    size_t stride[2] = {0, 0};      // flag line + message, rounded up to 256
    char *base0 = nullptr;          // the allocation of creation
    bool own[2] = {false, false};   // box_mem[side] is an allocation of its own (grown)
    char *peer_base[2] = {nullptr, nullptr};  // the neighbour's mailboxes for MY messages (its control line first)
    size_t total0 = 0;                        // bytes of base0
    void *mapped[2] = {nullptr, nullptr};     // hipIpcOpenMemHandle mappings to close (null: the neighbour is in this process)
    bool self[2] = {false, false};            // the neighbour on that side is this very link
    unsigned *d_err = nullptr;
    unsigned long long *d_wait = nullptr;  // [0] ticks spent waiting for the neighbours' messages, [1] waits (peer_wait)
    unsigned seq = 0;          // exchanges made so far
    unsigned gen[2] = {0, 0};  // growths of each edge so far
    // mailboxes and mappings an edge has outgrown: let go with the link, not during the run (the neighbour may still hold
    // a mapping of mine, and I one of its: nothing is unmapped or freed while the other end could be looking)
    std::vector<char *> old_boxes;
    std::vector<void *> old_maps;
    bool connected = false;
    int alloc_kind = 0;        // 0 uncached, 1 fine-grained, 2 plain device memory
};

// what travels through an edge's control line when it grows
struct PeerCtl {
    unsigned gen, cap;
    char handle[OPAL_PEER_HANDLE_BYTES];
};

namespace {

// The exchange as a launch of its own (the unfused paths: opal_grid_synchronize's halo-only exchange, the separate-operator
// and emission steps' whole messages): send both messages, wait for the neighbours', copy what arrived into the link's recv
// buffers, where the unpack and synchronize kernels read them.
// Round 4: PEER_MOVE_BLOCKS workgroups share the payload (a single one moved a colliding-beams step's 9 MB of photons at
// ~10 GB/s: 2 ms per step at four subdomains).  Header and halo go out from workgroup 0; every workgroup stores its share of
// the row blocks (system-scope write-through stores), fences, and takes a ticket; whoever draws the last ticket knows that
// every share is out and publishes the sequence numbers.  No workgroup waits for another one of its own launch, so the
// launch needs no co-residency; all of them then poll their own mailboxes for the neighbours' numbers (bounded).
constexpr unsigned PEER_MOVE_BLOCKS = 64;
__global__ __launch_bounds__(1024) void k_peer_move(PeerArgs a, char *recv_left, char *recv_right, unsigned *ticket)
{
    constexpr unsigned HEAD_WORDS = (unsigned)(LINK_HEAD_BYTES / 8);
    const unsigned nthreads = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) {
        for (int side = 0; side < 2; side++) {
            const char *msg = a.send[side];
            if (!msg || threadIdx.x >= HEAD_WORDS) continue;
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.dst[side] + PEER_FLAG_BYTES) + threadIdx.x,
                               reinterpret_cast<const unsigned long long *>(msg)[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for (int side = 0; side < 2; side++) {
        const char *msg = a.send[side];
        if (!msg) continue;
        const unsigned words = peer_prefix_words(msg, a, side);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(msg);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.dst[side] + PEER_FLAG_BYTES);
        for (unsigned i = HEAD_WORDS + gtid; i < words; i += nthreads)
            __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();  // (every wave waits for its stores before it arrives)
    if (threadIdx.x == 0) {
        __threadfence_system();
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {  // every workgroup's share is out
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch: stream order)
            for (int side = 0; side < 2; side++)
                if (a.send[side])
                    __hip_atomic_store(reinterpret_cast<unsigned *>(a.dst[side]), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    peer_wait(a);
    for (int side = 0; side < 2; side++) {
        if (!a.box[side]) continue;
        const char *msg = a.box[side] + PEER_FLAG_BYTES;
        const unsigned words = peer_prefix_words(msg, a, side);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(msg);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(side == 0 ? recv_left : recv_right);
        for (unsigned i = gtid; i < words; i += nthreads) dst[i] = src[i];
    }
}

int peer_exchange2(void *user, const void *send_left, const void *send_right, void *recv_left, void *recv_right,
                   size_t n_left, size_t n_right)
{
    opal_link_t *l = static_cast<opal_link_t *>(user);
    PeerArgs a;
    const bool head_only = (send_left ? n_left : n_right) == LINK_HEAD_BYTES;
    if (link_peer_next(l, send_left != nullptr, send_right != nullptr, head_only, nullptr, a)) return 1;
    a.send[0] = static_cast<const char *>(send_left);
    a.send[1] = static_cast<const char *>(send_right);
    // (a halo-only exchange is 88 words: one workgroup)
    // (Not more workgroups for larger messages: every workgroup of the launch ends in a wait for the neighbours' messages, and
    // workgroups that wait must never keep the neighbours' senders off the chip -- round 5: 257 workgroups of the tail kernel per
    // rank, four ranks on one card, every rank's publishing workgroup behind its own waiting ones: 20 s, the timeout.  64 x 1024
    // threads of six ranks fit a card at once.)
    k_peer_move<<<head_only ? 1 : PEER_MOVE_BLOCKS, 1024, 0, l->stream>>>(a, static_cast<char *>(recv_left), static_cast<char *>(recv_right),
                                                                           l->peer->d_err + 1);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
// An edge's growth record into the control line of the neighbour's mailboxes: written the way the messages are (system-scope
// stores from a kernel over the mapping the link already uses), not by a host copy into another device's memory.
__global__ void k_peer_ctl_write(unsigned long long *dst, PeerCtl rec)
{
    static_assert(sizeof(PeerCtl) % 8 == 0 && sizeof(PeerCtl) <= PEER_CTL_BYTES, "control record");
    constexpr unsigned W = sizeof(PeerCtl) / 8;
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&rec);
    // everything but the first word (generation, capacity), then the fence, then that word: a reader that sees the
    // generation sees the handle
    if (threadIdx.x >= 1 && threadIdx.x < W) __hip_atomic_store(dst + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(dst, src[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int peer_exchange(void *user, const void *send_left, const void *send_right, void *recv_left, void *recv_right, size_t nbytes)
{
    return peer_exchange2(user, send_left, send_right, recv_left, recv_right, nbytes, nbytes);
}

}  // namespace

extern "C" int opal_link_peer_status(opal_link_t *l, int *timeouts, int *alloc_kind, uint32_t *exchanges);

namespace opal {

bool link_is_peer(const opal_link_t *l) { return l && l->peer && l->peer->connected; }

int link_peer_next(opal_link_t *l, bool to_left, bool to_right, bool head_only, const HaloSrc *halo, PeerArgs &a)
{
    opal_peer *p = l->peer;
    if (!p || !p->connected) {
        set_error("peer link: not connected (opal_link_peer_connect)");
        return OPAL_ERR_LINK;
    }
    memset(&a, 0, sizeof a);
    const unsigned par = p->seq & 1u;
    p->seq++;
    a.enabled = 1;
    a.seq = p->seq;
    a.layout = l->layout;
    a.cap[0] = (unsigned)l->cap_side[0];
    a.cap[1] = (unsigned)l->cap_side[1];
    a.head_only = head_only;
    a.err = p->d_err;
    a.wait = p->d_wait;
    if (halo) {
        a.pack_halo = 1;
        a.halo = *halo;
    }
    // what I send left lands in the left neighbour's from-right mailbox, and the other way round
    if (to_left) {
        a.send[0] = l->d_send_left;
        a.dst[0] = p->peer_base[0] + PEER_CTL_BYTES + par * p->stride[0];
        a.box[0] = p->box_mem[0] + PEER_CTL_BYTES + par * p->stride[0];
    }
    if (to_right) {
        a.send[1] = l->d_send_right;
        a.dst[1] = p->peer_base[1] + PEER_CTL_BYTES + par * p->stride[1];
        a.box[1] = p->box_mem[1] + PEER_CTL_BYTES + par * p->stride[1];
    }
    return OPAL_OK;
}

int link_peer_check(opal_link_t *l)
{
    if (!l || !l->peer || !l->peer->d_err) return OPAL_OK;
    return opal_link_peer_status(l, nullptr, nullptr, nullptr);
}

void link_peer_free(opal_link_t *l)
{
    opal_peer *p = l->peer;
    if (!p) return;
    for (int s = 0; s < 2; s++)
        if (p->mapped[s] && (s == 0 || p->mapped[1] != p->mapped[0])) hipIpcCloseMemHandle(p->mapped[s]);
    for (void *m : p->old_maps) hipIpcCloseMemHandle(m);
    for (int s = 0; s < 2; s++)
        if (p->own[s] && p->box_mem[s]) hipFree(p->box_mem[s]);
    for (char *b : p->old_boxes) hipFree(b);
    if (p->base0) hipFree(p->base0);
    if (p->d_err) hipFree(p->d_err);
    if (p->d_wait) hipFree(p->d_wait);
    delete p;
    l->peer = nullptr;
}

}  // namespace opal

// memory both ends see coherently: uncached, else fine-grained; OPAL_PEER_ALLOC=plain|fine|uncached overrides
static hipError_t peer_alloc(opal_peer *p, char **mem, size_t bytes)
{
    const char *want = getenv("OPAL_PEER_ALLOC");
    hipError_t e = hipErrorUnknown;
    if (!want || !strcmp(want, "uncached")) {
        e = hipExtMallocWithFlags((void **)mem, bytes, hipDeviceMallocUncached);
        p->alloc_kind = 0;
    }
    if (e != hipSuccess && (!want || !strcmp(want, "fine"))) {
        e = hipExtMallocWithFlags((void **)mem, bytes, hipDeviceMallocFinegrained);
        p->alloc_kind = 1;
    }
    if (e != hipSuccess && want && !strcmp(want, "plain")) {
        e = hipMalloc((void **)mem, bytes);
        p->alloc_kind = 2;
    }
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}

static size_t peer_stride(const opal_link_t *l, size_t cap)
{
    return (PEER_FLAG_BYTES + link_prefix_bytes(l, (unsigned)cap) + 255) / 256 * 256;
}

// The export of a fresh allocation has been seen to fail once ("invalid argument", one of four processes sharing a card,
// right after each had freed an exported allocation): tried again, then with another allocation, before the caller is told
// to use another transport.
static int peer_export(opal_peer *p, char **mem, size_t bytes, void *handle64)
{
    static_assert(sizeof(hipIpcMemHandle_t) == OPAL_PEER_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t h;
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 4; attempt++) {
        e = hipIpcGetMemHandle(&h, *mem);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        usleep(20000u << attempt);
        if (attempt >= 1 && attempt < 3) {  // (a fresh allocation after the last attempt would never be exported)
            char *fresh = nullptr;
            if (peer_alloc(p, &fresh, bytes) == hipSuccess) {  // (while the old one is still held: another address)
                hipFree(*mem);
                *mem = fresh;
                OPAL_HIP(hipMemset(*mem, 0, bytes));
                OPAL_HIP(hipDeviceSynchronize());
            }
        }
    }
    if (e != hipSuccess) return hip_fail(e, "hipIpcGetMemHandle(mailboxes), 4 attempts");
    memcpy(handle64, &h, sizeof h);
    return OPAL_OK;
}

namespace opal {

// Edge `side` gets mailboxes with room for `new_cap` particles per species.  Both ends of the edge are here for the same
// exchange (link_plan decides from numbers both hold) and have drained their streams: my tail kernel of the exchange
// before has consumed the neighbour's last message, so the neighbour has written its last word into my old mailboxes, and
// I mine into its.  Each end allocates, exports, writes {generation, capacity, handle} into the control line of the other's
// OLD mailboxes (a plain copy through the mapping it already holds) -- link_peer_grow_begin --, then waits for the other's
// record in its own control line, maps the new mailboxes and lets the old ones go -- link_peer_grow_finish.  Two phases
// because a subdomain may grow both its edges for the same exchange: every record is written before anybody waits for one
// (round a periodic ring, "left first, then right" would have every subdomain wait for its left neighbour's right edge).
// No host channel is needed, and nobody but the two ends of an edge takes part.
struct PeerGrowth {
    char *fresh = nullptr;
    size_t stride = 0, cap = 0;
    unsigned gen = 0;
};

int link_peer_grow_begin(opal_link_t *l, int side, size_t new_cap, void **state)
{
    *state = nullptr;
    opal_peer *p = l->peer;
    if (!p) return OPAL_OK;
    if (!p->connected) {  // (link_grow refuses before it gets here)
        set_error("peer link: not connected (opal_link_peer_connect)");
        return OPAL_ERR_LINK;
    }
    OPAL_HIP(hipDeviceSynchronize());
    PeerGrowth *gr = new PeerGrowth();
    gr->stride = peer_stride(l, new_cap);
    gr->cap = new_cap;
    const size_t bytes = PEER_CTL_BYTES + 2 * gr->stride;
    hipError_t e = peer_alloc(p, &gr->fresh, bytes);
    if (e != hipSuccess) {
        delete gr;
        return hip_fail(e, "allocating larger peer mailboxes");
    }
    *state = gr;  // (from here the caller lets it go if anything fails: link_peer_grow_abort)
    OPAL_HIP(hipMemset(gr->fresh, 0, bytes));
    OPAL_HIP(hipDeviceSynchronize());
    gr->gen = ++p->gen[side];
    if (p->self[side]) return OPAL_OK;
    PeerCtl mine;
    memset(&mine, 0, sizeof mine);
    mine.gen = gr->gen;
    mine.cap = (unsigned)new_cap;
    if (int rc = peer_export(p, &gr->fresh, bytes, mine.handle)) return rc;
    k_peer_ctl_write<<<1, 64, 0, l->stream>>>(reinterpret_cast<unsigned long long *>(p->peer_base[side]), mine);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipStreamSynchronize(l->stream));
    if (getenv("OPAL_LINK_DEBUG"))
        fprintf(stderr, "[link] rank %d side %d: record of growth %u (%zu slots, %zu bytes) written\n", l->rank, side, gr->gen, new_cap, bytes);
    return OPAL_OK;
}

int link_peer_grow_finish(opal_link_t *l, int side, void *state)
{
    opal_peer *p = l->peer;
    PeerGrowth *gr = static_cast<PeerGrowth *>(state);
    if (!p || !gr) return OPAL_OK;
    auto retire = [&](int s) {  // my old mailboxes of side s (the allocation of creation goes with the link)
        if (p->own[s]) p->old_boxes.push_back(p->box_mem[s]);
        p->own[s] = true;
    };
    if (p->self[side]) {
        // a lone periodic subdomain is its own neighbour: what it sends through this side lands in the mailboxes of its
        // OTHER side, which link_plan grows in the same call (the two edges read the same numbers)
        retire(side);
        p->box_mem[side] = gr->fresh;
        p->stride[side] = gr->stride;
        p->peer_base[1 - side] = gr->fresh;
        delete gr;
        return OPAL_OK;
    }
    PeerCtl theirs;
    const auto t0 = std::chrono::steady_clock::now();
    if (getenv("OPAL_LINK_DEBUG")) fprintf(stderr, "[link] rank %d side %d: waiting for the neighbour's record %u\n", l->rank, side, gr->gen);
    auto give_up = [&](int rc) {  // (the fresh mailboxes were exported but never used: nobody holds a mapping of them)
        link_peer_grow_abort(l, gr);
        return rc;
    };
    for (;;) {
        hipError_t e = hipMemcpy(&theirs, p->box_mem[side], sizeof theirs, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return give_up(hip_fail(e, "reading the neighbour's growth record"));
        if (theirs.gen == gr->gen) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
            set_error("peer link: the neighbour on side %d never sent the handle of its larger mailboxes (growth %u to %zu "
                      "particles per species)", side, gr->gen, gr->cap);
            return give_up(OPAL_ERR_LINK);
        }
        usleep(200);
    }
    if (theirs.cap != (unsigned)gr->cap) {
        set_error("peer link: the two ends of edge %d disagree on its new capacity (%zu here, %u there)", side, gr->cap, theirs.cap);
        return give_up(OPAL_ERR_LINK);
    }
    hipIpcMemHandle_t h;
    memcpy(&h, theirs.handle, sizeof h);
    void *map = nullptr;
    if (getenv("OPAL_LINK_DEBUG")) fprintf(stderr, "[link] rank %d side %d: record %u here, mapping\n", l->rank, side, gr->gen);
    {
        hipError_t e = hipIpcOpenMemHandle(&map, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return give_up(hip_fail(e, "hipIpcOpenMemHandle(the neighbour's larger mailboxes)"));
    }
    // the mapping of the neighbour's old mailboxes is shared by both sides when both neighbours are one subdomain
    void *old = p->mapped[side];
    p->mapped[side] = map;
    if (old && old != p->mapped[1 - side]) p->old_maps.push_back(old);
    p->peer_base[side] = static_cast<char *>(map);
    retire(side);
    p->box_mem[side] = gr->fresh;
    p->stride[side] = gr->stride;
    delete gr;
    return OPAL_OK;
}

void link_peer_grow_abort(opal_link_t *l, void *state)
{
    PeerGrowth *gr = static_cast<PeerGrowth *>(state);
    if (!gr) return;
    if (gr->fresh) hipFree(gr->fresh);
    delete gr;
}

}  // namespace opal

extern "C" {

int opal_link_create_peer(opal_link_t **out, int rank, int size, size_t particle_capacity, int species_slots,
                          int fields_per_particle, void *hip_stream)
{
    if (!out) {
        set_error("opal_link_create_peer: null output");
        return OPAL_ERR_ARG;
    }
    int rc = opal_link_create_layout(out, rank, size, 1, particle_capacity, species_slots, fields_per_particle, peer_exchange,
                                     nullptr, hip_stream);
    if (rc) return rc;
    opal_link_t *l = *out;
    l->user = l;
    l->exchange2 = peer_exchange2;
    l->bounded = false;  // no message size to plan: the sending kernel ships what the leavers occupy (link_plan still
                         // keeps plan_cap, the room the populations reserve for arrivals)
    l->growable = !getenv("OPAL_LINK_NO_GROWTH");  // an edge grows when more particles stand near it than it has room for
    opal_peer *p = new opal_peer();
    l->peer = p;
    p->stride[0] = p->stride[1] = peer_stride(l, l->cap);
    const size_t region = PEER_CTL_BYTES + 2 * p->stride[0];
    p->total0 = 2 * region;
    if (p->total0 >= ((size_t)1 << 31)) {
        // (an IPC-shared allocation of 2 GiB or more could not be mapped by the neighbour: link.cpp, link_plan)
        opal_link_free(l);
        *out = nullptr;
        set_error("opal_link_create_peer: room for %zu particles per species makes mailboxes of %zu bytes; IPC-shared allocations "
                  "must stay below 2 GiB (create the link smaller: its edges grow where the particles require it)",
                  particle_capacity, p->total0);
        return OPAL_ERR_ARG;
    }
    hipError_t e = peer_alloc(p, &p->base0, p->total0);
    if (e != hipSuccess) {
        p->base0 = nullptr;
        opal_link_free(l);
        *out = nullptr;
        return hip_fail(e, "allocating the peer mailboxes");
    }
    p->box_mem[0] = p->base0;
    p->box_mem[1] = p->base0 + region;
    e = hipMemset(p->base0, 0, p->total0);
    // [0] error word, [1] k_peer_move's ticket, [2..5] the first wait that gave up: side, the sequence number it waited for, the one it saw, its workgroup
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_err, 8 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(p->d_err, 0, 8 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_wait, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(p->d_wait, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {  // nothing of a half-built link survives a failure: the caller falls back to another transport
        opal_link_free(l);
        *out = nullptr;
        return hip_fail(e, "setting up the peer mailboxes");
    }
    return OPAL_OK;
}

int opal_link_peer_export(opal_link_t *l, void *handle64)
{
    if (!l || !l->peer || !handle64) {
        set_error("opal_link_peer_export: not a link made by opal_link_create_peer");
        return OPAL_ERR_ARG;
    }
    opal_peer *p = l->peer;
    char *before = p->base0;
    int rc = peer_export(p, &p->base0, p->total0, handle64);
    if (p->base0 != before) {  // (exported from a fresh allocation)
        p->box_mem[0] = p->base0;
        p->box_mem[1] = p->base0 + p->total0 / 2;
    }
    return rc;
}

int opal_link_peer_connect(opal_link_t *l, const void *left_handle64, const void *right_handle64)
{
    if (!l || !l->peer) {
        set_error("opal_link_peer_connect: not a link made by opal_link_create_peer");
        return OPAL_ERR_ARG;
    }
    opal_peer *p = l->peer;
    if (p->connected) {
        set_error("opal_link_peer_connect: already connected");
        return OPAL_ERR_ARG;
    }
    const void *hs[2] = {left_handle64, right_handle64};
    char *base[2];
    for (int s = 0; s < 2; s++) {
        if (!hs[s]) {
            base[s] = p->base0;  // the neighbour on that side is this very link (a lone periodic subdomain)
            p->self[s] = true;
        } else if (s == 1 && hs[0] && !memcmp(hs[0], hs[1], OPAL_PEER_HANDLE_BYTES)) {
            p->mapped[1] = p->mapped[0];  // two subdomains: both neighbours are the same one, mapped once
            base[1] = base[0];
        } else {
            hipIpcMemHandle_t h;
            memcpy(&h, hs[s], sizeof h);
            OPAL_HIP(hipIpcOpenMemHandle(&p->mapped[s], h, hipIpcMemLazyEnablePeerAccess));
            base[s] = static_cast<char *>(p->mapped[s]);
        }
    }
    // (every subdomain is created with the same capacity: the neighbours' regions are as long as mine)
    p->peer_base[0] = base[0] + p->total0 / 2;  // the left neighbour's from-right mailboxes
    p->peer_base[1] = base[1];                  // the right neighbour's from-left mailboxes
    p->connected = true;
    return OPAL_OK;
}

int opal_link_self_test(opal_link_t *l, int to_left, int to_right, int n)
{
    if (!l) {
        set_error("opal_link_self_test: null link");
        return OPAL_ERR_ARG;
    }
    if (!l->device_buffers) {  // (host-staged transports: completion is all there is to see)
        for (int k = 0; k < n; k++)
            if (int rc = link_exchange_halo(l, to_left != 0, to_right != 0, l->stream)) return rc;
        OPAL_HIP(hipStreamSynchronize(l->stream));
        return OPAL_OK;
    }
    // Every round ships a halo whose 80 doubles name their sender, the side they left through and the round, and checks
    // that what arrived is what the neighbour on that side must have sent in THIS round: a message that arrives late,
    // stale (the mailbox of two rounds ago) or torn fails here, before a simulation depends on it.
    constexpr int ND = (int)(LINK_HALO_BYTES / sizeof(double));
    auto pattern = [](int rank, int side, int round, int i) { return 1.0e6 * (rank + 1) + 1.0e4 * (side + 1) + 1.0e2 * round + i; };
    const int left = (l->rank - 1 + l->size) % l->size, right = (l->rank + 1) % l->size;
    double buf[ND];
    // ... n rounds one at a time, then 4 n exchanges back to back with nothing between them (the cadence of a run: the
    // two mailboxes of a direction are reused without acknowledgement, peer_device.hpp), then one more checked round
    for (int k = 0; k <= n; k++) {
        if (k == n)
            for (int q = 0; q < 4 * n; q++)
                if (int rc = link_exchange_halo(l, to_left != 0, to_right != 0, l->stream)) return rc;
        for (int side = 0; side < 2; side++) {
            for (int i = 0; i < ND; i++) buf[i] = pattern(l->rank, side, k, i);
            OPAL_HIP(hipMemcpyAsync(link_send(l, side) + LINK_HEADER_BYTES, buf, sizeof buf, hipMemcpyHostToDevice, l->stream));
        }
        OPAL_HIP(hipStreamSynchronize(l->stream));
        if (int rc = link_exchange_halo(l, to_left != 0, to_right != 0, l->stream)) return rc;
        for (int side = 0; side < 2; side++) {
            if (!(side == 0 ? to_left : to_right)) continue;
            OPAL_HIP(hipMemcpyAsync(buf, link_recv(l, side) + LINK_HEADER_BYTES, sizeof buf, hipMemcpyDeviceToHost, l->stream));
            OPAL_HIP(hipStreamSynchronize(l->stream));
            // from the left neighbour comes what it sent to the RIGHT, and the other way round
            const int from = side == 0 ? left : right, their_side = side == 0 ? 1 : 0;
            for (int i = 0; i < ND; i++)
                if (buf[i] != pattern(from, their_side, k, i)) {
                    set_error("link self-test: round %d, halo word %d from subdomain %d reads %.17g, not %.17g (a late, stale or "
                              "torn message; mailbox memory kind %d: 0 uncached, 1 fine-grained, 2 plain)", k, i, from, buf[i],
                              pattern(from, their_side, k, i), l->peer ? l->peer->alloc_kind : -1);
                    return OPAL_ERR_LINK;
                }
        }
    }
    // (the send buffers' halo regions are rewritten by every step's pack; nothing to restore)
    return l->peer ? opal_link_peer_status(l, nullptr, nullptr, nullptr) : OPAL_OK;
}

// link_plan against its specification, on a link that exists in host memory only (no device is touched: callable, and
// called by the CPU-side tests, without a GPU).  Three exchange patterns are played: every exchange counts every species
// (the driver's steps); exchanges alternate between electrons and ions (the separate operators called species by species);
// and a peer link, which files every LINK_PEER_EVERY-th exchange only.  After each plan the "device" files the exchange's
// zone counts the way the tail kernel would.  Expected, from the argument in link.cpp: the entry of the most recent
// exchange <= X - LINK_ZONE that counted the species (peer: the most recent filed one, at most LINK_PEER_ZONE old),
// mine + theirs, maximum over the species of the exchange, plus LINK_SLACK, in whole row blocks, capped at the capacity.
int opal_link_plan_selftest(void)
{
    auto fail = [](const char *what, int pattern, unsigned X, unsigned got, unsigned want) {
        set_error("link_plan self-test, pattern %d, exchange %u: %s is %u, expected %u", pattern, X, what, got, want);
        return OPAL_ERR_LINK;
    };
    for (int pattern = 0; pattern < 3; pattern++) {
        const bool peer = pattern == 2;
        opal_link_t l;
        memset((void *)&l, 0, sizeof l);
        LinkBounds ring[LINK_RING];
        memset((void *)ring, 0, sizeof ring);
        opal_peer pp;
        pp.connected = true;
        l.cap = 1u << 20;
        l.cap_side[0] = l.cap_side[1] = l.cap;
        l.nslots = 3;
        l.nfields = 6;
        l.bounded = !peer;
        l.exchange2 = reinterpret_cast<opal_exchange2_fn>(&ring);  // (never called: non-null is all link_plan asks)
        l.h_bounds = ring;
        l.peer = peer ? &pp : nullptr;
        unsigned mask_of[64], filed[64];
        auto count = [](unsigned e, int side, int sp, int whose) { return 1000u + 37u * e + 11u * (unsigned)side + 5u * (unsigned)sp + 400u * (unsigned)whose; };
        for (unsigned X = 0; X < 48; X++) {
            const unsigned mask = pattern == 1 ? (X % 2 ? 2u : 1u) : 3u;
            mask_of[X] = mask;
            l.xseq = X;
            if (X == 21) {
                const unsigned need21[2] = {5000u, 7000u};
                const bool moved21[2] = {true, false};  // (the right cut stays where it is: its side goes on by the entries)
                link_hold(&l, need21, moved21);
            }
            if (int rc = link_plan(&l, mask)) return rc;
            // the specification
            unsigned want[2] = {(unsigned)l.cap, (unsigned)l.cap};
            if (X >= (unsigned)LINK_ZONE) {
                unsigned need[2] = {0, 0};
                bool all = true;
                for (int sp = 0; sp < 2 && all; sp++) {
                    if (!(mask & (1u << sp))) continue;
                    int found = -1;
                    for (int e = (int)X - LINK_ZONE; e >= 0 && (int)X - e < LINK_RING; e--) {
                        if (peer && (int)X - e > LINK_PEER_ZONE) break;
                        if (filed[e] && (mask_of[e] & (1u << sp))) { found = e; break; }
                    }
                    if (found < 0) { all = false; break; }
                    for (int side = 0; side < 2; side++) {
                        unsigned n = count((unsigned)found, side, sp, 0) + count((unsigned)found, side, sp, 1);
                        if (n > need[side]) need[side] = n;
                    }
                }
                if (all)
                    for (int side = 0; side < 2; side++) {
                        unsigned slots = (need[side] + LINK_SLACK + LINK_ROWS - 1) / LINK_ROWS * LINK_ROWS;
                        want[side] = slots < (unsigned)l.cap ? slots : (unsigned)l.cap;
                    }
            }
            // ... and after the cuts have moved (link_hold at exchange 21, sim.cpp: sim_recut): what stood near the new cuts
            // then for the next zone exchanges -- on the side whose cut moved --, and never again an entry of an exchange before the move
            const unsigned held_at = 21, held_need[2] = {5000u, 7000u};
            const unsigned zone_cells = peer ? (unsigned)LINK_PEER_ZONE : (unsigned)LINK_ZONE;
            if (X >= held_at) {
                if (X - held_at < zone_cells) {
                    want[0] = (held_need[0] + LINK_SLACK + LINK_ROWS - 1) / LINK_ROWS * LINK_ROWS;  // (want[1]: as if nothing had happened)
                } else {
                    unsigned need[2] = {0, 0};
                    bool all = true;
                    for (int sp = 0; sp < 2 && all; sp++) {
                        if (!(mask & (1u << sp))) continue;
                        int found = -1;
                        for (int e = (int)X - LINK_ZONE; e >= (int)held_at && (int)X - e < LINK_RING; e--) {
                            if (peer && (int)X - e > LINK_PEER_ZONE) break;
                            if (filed[e] && (mask_of[e] & (1u << sp))) { found = e; break; }
                        }
                        if (found < 0) { all = false; break; }
                        for (int side = 0; side < 2; side++) {
                            unsigned n = count((unsigned)found, side, sp, 0) + count((unsigned)found, side, sp, 1);
                            if (n > need[side]) need[side] = n;
                        }
                    }
                    for (int side = 0; side < 2; side++) {
                        unsigned slots = (need[side] + LINK_SLACK + LINK_ROWS - 1) / LINK_ROWS * LINK_ROWS;
                        want[side] = all && slots < (unsigned)l.cap ? slots : (unsigned)l.cap;
                    }
                }
            }
            for (int side = 0; side < 2; side++) {
                if (l.plan_cap[side] != want[side]) return fail("plan_cap", pattern, X, l.plan_cap[side], want[side]);
                const unsigned ship = peer ? (unsigned)l.cap : want[side];  // a peer link ships what leaves, up to the capacity
                if (l.send_cap[side] != ship) return fail("send_cap", pattern, X, l.send_cap[side], ship);
            }
            if (X >= 8 && want[0] == (unsigned)l.cap && !(X >= held_at && X < held_at + 2 * LINK_RING))
                return fail("plan_cap (no entry found where one exists)", pattern, X, want[0], 0);
            // the tail kernel of exchange X files its counts (a peer link: every LINK_PEER_EVERY-th exchange)
            filed[X] = !peer || X % LINK_PEER_EVERY == 0;
            if (filed[X]) {
                LinkBounds *b = &ring[X % LINK_RING];
                for (int side = 0; side < 2; side++)
                    for (int sp = 0; sp < 3; sp++) {
                        b->mine[side][sp] = (mask & (1u << sp)) ? count(X, side, sp, 0) : 0u;
                        b->theirs[side][sp] = (mask & (1u << sp)) ? count(X, side, sp, 1) : 0u;
                    }
                b->seq = X + 1;
            }
        }
    }
    return OPAL_OK;
}

int opal_link_peer_wait(opal_link_t *l, double *microseconds, uint64_t *waits, int reset)
{
    if (!l || !l->peer || !l->peer->d_wait) {
        set_error("opal_link_peer_wait: not a link made by opal_link_create_peer");
        return OPAL_ERR_ARG;
    }
    unsigned long long w[2] = {0, 0};
    OPAL_HIP(hipMemcpyAsync(w, l->peer->d_wait, sizeof w, hipMemcpyDeviceToHost, l->stream));
    if (reset) OPAL_HIP(hipMemsetAsync(l->peer->d_wait, 0, sizeof w, l->stream));
    OPAL_HIP(hipStreamSynchronize(l->stream));
    if (microseconds) *microseconds = (double)w[0] / 100.0;  // wall_clock64() ticks at 100 MHz
    if (waits) *waits = w[1];
    return OPAL_OK;
}

int opal_link_peer_status(opal_link_t *l, int *timeouts, int *alloc_kind, uint32_t *exchanges)
{
    if (!l || !l->peer) {
        set_error("opal_link_peer_status: not a link made by opal_link_create_peer");
        return OPAL_ERR_ARG;
    }
    unsigned errw[8] = {0};
    OPAL_HIP(hipMemcpyAsync(errw, l->peer->d_err, sizeof errw, hipMemcpyDeviceToHost, l->stream));
    OPAL_HIP(hipStreamSynchronize(l->stream));
    const unsigned err = errw[0];
    if (timeouts) *timeouts = (int)(err & 1u);
    if (alloc_kind) *alloc_kind = l->peer->alloc_kind;
    if (exchanges) *exchanges = l->peer->seq;
    if (err) {
        set_error("peer link: a neighbour's message did not arrive within %.0f s (subdomain %d: the %s mailbox held sequence number %u where "
                  "%u was waited for, workgroup %u; %u exchanges enqueued so far)", (double)PEER_TIMEOUT_TICKS / 1e8, l->rank,
                  errw[2] ? "right" : "left", errw[4], errw[3], errw[5], l->peer->seq);
        return OPAL_ERR_LINK;
    }
    return OPAL_OK;
}

}  // extern "C"
