// Peer-mapped mailboxes: the neighbour exchange done BY the kernels of the step, without a transport in between.
//
// Replaces, on one node, the reference's blocking MPI point-to-point traffic (yee.rs:351-409 for the halo,
// particle/mod.rs:303-354 for migrating particles) and this library's own RCCL send/recv pair: every subdomain owns
// four mailboxes in device memory that its neighbours can address (hipIpcOpenMemHandle across processes, the plain
// pointer inside one) -- from the left and from the right neighbour, two of each, used alternately.  A step's kernel
// writes its message (64-byte header, 640-byte halo, the row blocks the leavers occupy) straight into the
// neighbour's mailbox over xGMI, publishes it with a sequence number stored behind a system-scope release, and
// then waits for the sequence number in its own mailbox.  No launch, no host call and no size negotiation stands
// between a push and the arrivals it exchanges: the message is as long as the leavers it carries.
//
// Why two mailboxes per direction suffice without acknowledgements: exchange k uses mailbox k & 1.  A neighbour
// writes exchange k + 2 into the same mailbox only after it has finished its own exchange k + 1, which needed my
// message k + 1, which I send after my unpack of exchange k has completed (stream order).  So the mailbox is
// never overwritten before it has been read.
//
// Memory model.  The mailboxes are allocated uncached / fine-grained (peer.hip); payload words are written with
// system-scope relaxed atomic stores (write-through), then the workgroup's barrier, then ONE lane stores the
// sequence number with a system-scope release.  The receiver's lane 0 polls it with system-scope relaxed loads,
// issues one system-scope acquire fence, and releases the rest of its workgroup through a barrier; payload loads follow.
// Every wait is bounded (PEER_TIMEOUT_TICKS of the 100 MHz wall clock): a neighbour that never sends makes the
// kernel raise an error word and carry on, so that every wave reaches its exit.
#pragma once
#include "link.hpp"

namespace opal {
constexpr size_t PEER_FLAG_BYTES = 64;          // the sequence number's line, in front of the message
constexpr size_t PEER_CTL_BYTES = 256;          // in front of an edge's two mailboxes: the line through which the two ends of
                                                // the edge hand each other the handle of LARGER mailboxes (peer.hip: link_peer_grow)
constexpr long long PEER_TIMEOUT_TICKS = 2000000000LL;  // 20 s of wall_clock64() (100 MHz)
constexpr long long PEER_TICK_MAX = 1000000LL;          // the most one pass of a polling loop is counted for (10 ms)

#ifdef __HIPCC__
// The time a wave has spent polling, summed pass by pass.  NOT `now - start`: with six processes sharing one card (round 5's
// rehearsals) the waits of two subdomains on the same neighbour gave up at the same exchange -- the mailbox still held the
// message of two exchanges before -- in runs that, by the host's clock, never stalled (they reached the next drain sooner than
// runs without the error): the difference of two readings of the counter had jumped by more than the 20 s, whether because
// the waves were off the chip in between (other processes' turn) or because the counter itself is not one clock for
// everybody.  One pass of a polling loop takes microseconds; a longer gap is not time spent waiting here and counts for
// PEER_TICK_MAX at the most, so that a wait only gives up after 20 s of its own polling (>= 2000 anomalies otherwise).
struct SpinClock {
    long long last, spent;
    __device__ SpinClock() : last(wall_clock64()), spent(0) {}
    __device__ bool expired()
    {
        const long long now = wall_clock64(), d = now - last;
        last = now;
        spent += d < 0 ? 0 : d > PEER_TICK_MAX ? PEER_TICK_MAX : d;
        return spent > PEER_TIMEOUT_TICKS;
    }
};
#endif

// what a packed halo is read from (grid rows), when the sending kernel packs it itself
struct HaloSrc {
    const double *eb, *jr;
    size_t lenp;
    int len;
};

struct PeerArgs {
    int enabled;
    const char *send[2];   // my outgoing messages (device buffers the push packed the leavers into); null: no neighbour there
    char *dst[2];          // the neighbour's mailbox for this exchange: flag line, then the message
    const char *box[2];    // my mailboxes for this exchange: [0] from the left neighbour, [1] from the right one
    unsigned seq;          // this exchange's sequence number (>= 1, monotonic)
    LinkLayout layout;
    unsigned cap[2];       // slots per species a message of the left / right edge has room for
    int head_only;         // halo-only exchange (yee.rs:358-409 on its own): header + halo, no leavers
    int pack_halo;         // the halo is taken from the grid rows (HaloSrc) instead of the send message
    HaloSrc halo;
    unsigned *err;         // device word: bit 0 = a neighbour's message never came; the word behind it: the senders' ticket
    unsigned long long *wait;  // device words: [0] ticks (100 MHz) workgroup 0 has spent in peer_wait so far, [1] its waits
};

#ifdef __HIPCC__
__device__ inline double halo_value(const HaloSrc &h, int side, int r)
{
    // yee.rs:353-354: send_left = cell[0..8], send_right = cell[len-8..len]; per cell rho j[3] E[3] B[3]
    const int k = r / 10, comp = r % 10;
    const int cellidx = side == 0 ? k : h.len - 8 + k;
    if (comp == 0) return h.jr[3 * h.lenp + cellidx];
    if (comp < 4) return h.jr[(size_t)(comp - 1) * h.lenp + cellidx];
    return h.eb[(size_t)(comp - 4) * h.lenp + cellidx];
}

// bytes of a message that carry something: header + halo + the row blocks its leavers reach into
__device__ inline unsigned peer_prefix_words(const char *msg, const PeerArgs &a, int side)
{
    if (a.head_only) return (unsigned)(LINK_HEAD_BYTES / 8);
    const unsigned *head = reinterpret_cast<const unsigned *>(msg);
    unsigned m = 0;
    for (unsigned s = 0; s < a.layout.nslots; s++) {
        unsigned c = head[LINK_H_COUNT + s];
        c = c < a.cap[side] ? c : a.cap[side];
        m = m > c ? m : c;
    }
    const unsigned blocks = (m + LINK_ROWS - 1) / LINK_ROWS;
    return (unsigned)((LINK_HEAD_BYTES + (size_t)blocks * a.layout.nslots * a.layout.blk) / 8);
}

// Called by every thread of ONE workgroup: writes both outgoing messages into the neighbours' mailboxes and
// publishes them.  Contains barriers.
__device__ inline void peer_send(const PeerArgs &a)
{
    constexpr unsigned HEAD_WORDS = (unsigned)(LINK_HEAD_BYTES / 8), HEADER_WORDS = (unsigned)(LINK_HEADER_BYTES / 8);
    // header and halo first: their loads depend on nothing (the rows' extent is read from the header)
    for (int side = 0; side < 2; side++) {
        const char *msg = a.send[side];
        if (!msg || threadIdx.x >= HEAD_WORDS) continue;
        const unsigned i = threadIdx.x;
        unsigned long long v;
        if (a.pack_halo && i >= HEADER_WORDS)
            v = (unsigned long long)__double_as_longlong(halo_value(a.halo, side, (int)(i - HEADER_WORDS)));
        else
            v = reinterpret_cast<const unsigned long long *>(msg)[i];
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.dst[side] + PEER_FLAG_BYTES) + i, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int side = 0; side < 2; side++) {
        const char *msg = a.send[side];
        if (!msg) continue;
        const unsigned words = peer_prefix_words(msg, a, side);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(msg);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.dst[side] + PEER_FLAG_BYTES);
        for (unsigned i = HEAD_WORDS + threadIdx.x; i < words; i += blockDim.x)
            __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();  // (every wave waits for its stores before it arrives)
    if (threadIdx.x == 0)
        for (int side = 0; side < 2; side++)
            if (a.send[side])  // release: everything stored above is visible before the sequence number is
                __hip_atomic_store(reinterpret_cast<unsigned *>(a.dst[side]), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same send shared among the workgroups of a launch (the tail of a linked step at a cut a front is crossing: megabytes of
// leavers per step, which one workgroup moves at a few tens of GB/s).  peer_send_head: header and halo of both messages, and
// the row blocks of the species slots in `rows_of` (bit s = slot s: those no other workgroup takes).  peer_send_rows: the row
// blocks of ONE species slot of ONE message, every `of`-th block starting at block `part`.  peer_publish: called by every
// sending workgroup after its stores (contains a barrier); the one that draws the last of `senders` tickets knows every
// share is out and stores the sequence numbers.  No workgroup waits for another of its own launch.
__device__ inline void peer_send_rows(const PeerArgs &a, int side, unsigned slot, unsigned part, unsigned of)
{
    const char *msg = a.send[side];
    if (!msg || a.head_only) return;
    unsigned c = reinterpret_cast<const unsigned *>(msg)[LINK_H_COUNT + slot];
    c = c < a.cap[side] ? c : a.cap[side];
    const unsigned blocks = (c + LINK_ROWS - 1) / LINK_ROWS, wpb = a.layout.blk / 8;
    for (unsigned b = part; b < blocks; b += of) {
        // 16 bytes per lane, written through to memory as the 8-byte system-scope stores of the header are (sc0 sc1: the
        // compiler has no 16-byte atomic store to ask for, and a plain or non-temporal store may stay in this GPU's L2, which a
        // neighbour that reads its mailbox uncached never sees -- round 5: six ranks, garbage cells).  A row block is a
        // multiple of 16 bytes at a multiple of 16; the publisher's system-scope fence orders them before the sequence number.
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const size_t off = LINK_HEAD_BYTES + ((size_t)b * a.layout.nslots + slot) * a.layout.blk;
        const u32x4 *src = reinterpret_cast<const u32x4 *>(msg + off);
        u32x4 *dst = reinterpret_cast<u32x4 *>(a.dst[side] + PEER_FLAG_BYTES + off);
        for (unsigned i = threadIdx.x; i < wpb / 2; i += blockDim.x) {
            const u32x4 v = src[i];
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(v) : "memory");
        }
        // (the compiler does not count stores it did not emit: every wave waits for its own here, before the publisher's barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
__device__ inline void peer_send_head(const PeerArgs &a, unsigned rows_of)
{
    constexpr unsigned HEAD_WORDS = (unsigned)(LINK_HEAD_BYTES / 8), HEADER_WORDS = (unsigned)(LINK_HEADER_BYTES / 8);
    for (int side = 0; side < 2; side++) {
        const char *msg = a.send[side];
        if (!msg || threadIdx.x >= HEAD_WORDS) continue;
        const unsigned i = threadIdx.x;
        unsigned long long v;
        if (a.pack_halo && i >= HEADER_WORDS)
            v = (unsigned long long)__double_as_longlong(halo_value(a.halo, side, (int)(i - HEADER_WORDS)));
        else
            v = reinterpret_cast<const unsigned long long *>(msg)[i];
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.dst[side] + PEER_FLAG_BYTES) + i, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int side = 0; side < 2; side++)
        for (unsigned s = 0; s < a.layout.nslots; s++)
            if (rows_of & (1u << s)) peer_send_rows(a, side, s, 0, 1);
}
// (returns true in thread 0 of the workgroup that drew the last ticket: every sender has read the send messages by then)
__device__ inline bool peer_publish(const PeerArgs &a, unsigned *ticket, unsigned senders)
{
    bool last = false;
    __syncthreads();  // (every wave waits for its stores before it arrives)
    if (threadIdx.x == 0) {
        __threadfence_system();
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == senders - 1) {
            last = true;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch: stream order)
            for (int side = 0; side < 2; side++)
                if (a.send[side])
                    __hip_atomic_store(reinterpret_cast<unsigned *>(a.dst[side]), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    return last;
}

// Called by every thread of a workgroup that is about to read what the neighbours sent.  Contains a barrier.
__device__ inline void peer_wait(const PeerArgs &a)
{
    if (threadIdx.x == 0) {
        SpinClock clock;
        for (int side = 0; side < 2; side++) {
            if (!a.box[side]) continue;
            const unsigned *flag = reinterpret_cast<const unsigned *>(a.box[side]);
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
                if (clock.expired()) {
                    if (!(atomicOr(a.err, 1u) & 1u)) {  // the first to give up says what it saw (opal_link_peer_status)
                        a.err[2] = (unsigned)side;
                        a.err[3] = a.seq;
                        a.err[4] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        a.err[5] = blockIdx.x;
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system scope: what the neighbours stored before their sequence numbers
        // How long this subdomain stood waiting for its neighbours, kept by ONE workgroup of every launch that waits (the
        // first: it always takes part).  What the number means on real hardware: my push was done and my message out, and
        // the neighbours' were not yet here -- link latency if it is the same few microseconds on every subdomain, imbalance
        // or clock skew if some subdomains wait long and others not at all (bench.py: per_rank.tail_wait_us_per_step).
        if (blockIdx.x == 0 && a.wait) {
            clock.expired();
            a.wait[0] += (unsigned long long)clock.spent;
            a.wait[1] += 1ull;
        }
    }
    __syncthreads();
}
#endif
}  // namespace opal
