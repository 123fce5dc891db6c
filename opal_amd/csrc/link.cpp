// Neighbour exchange between subdomains (stands in for the reference's MPI
// point-to-point traffic: yee.rs:351-409 for the halo, particle/mod.rs:303-354
// for migrating particles).  The library packs one fixed-layout message per
// direction; the transport (RCCL, gloo, ...) is the callback the host supplied.
//
// Message layout (bytes), P = particle_capacity of the link:
//   [0, 64)            header: uint32 count[S] (species slot 0 = electrons, 1 = ions, 2 = photons), padding
//   [64, 704)          halo: 8 cells x (rho jx jy jz Ex Ey Ez Bx By Bz) doubles  (yee.rs:353-354)
//   then per species slot s = 0 .. S-1 (each block 8-byte aligned):
//       int32  cell[P]            re-based cell index in the receiver's frame (mod.rs:291-293, 366-368)
//       double x[P] ux[P] uy[P] uz[P] w[P] prev_x[P]
//       double chi[P] work[P] y[P] z[P] tau[P] id[P]     only when the link was created with 12 fields
//       double tau_abs[P] tau_st[P] birth[P]              only with 15 fields (photon absorption)
// S = 2 and 6 fields by default; runs with photon emission use S = 3 and 12 fields (the reference ships
// whole structs, so optical depths and ids must travel with an electron that changes subdomain).
// A halo-only exchange (opal_grid_synchronize) moves the first 704 bytes; anything with particles moves
// the whole message -- all P slots of every column, because the receiver cannot know the counts before they
// arrive -- unless the transport offers an exact-size exchange (rccl_link.cpp: head first, then only the
// occupied part of every column, at the price of one host synchronisation per exchange).
// What a subdomain sends left arrives as its left neighbour's "recv_right".
#include <hip/hip_runtime_api.h>

#include <cstring>

#include "link.hpp"

using namespace opal;

namespace opal {

size_t link_species_offset(const opal_link_t *l, int slot) { return LINK_HEAD_BYTES + (size_t)slot * l->species_bytes; }

char *link_send(opal_link_t *l, int side) { return side == 0 ? l->d_send_left : l->d_send_right; }
char *link_recv(opal_link_t *l, int side) { return side == 0 ? l->d_recv_left : l->d_recv_right; }

// Leaver counts of the send messages start a step at zero.  The kernel that unpacks a step's arrivals also
// re-zeroes them (they have been shipped by then) and marks the link clean, so only the first step -- or one
// after an aborted exchange -- pays for the two memsets.
int link_begin(opal_link_t *l, hipStream_t st)
{
    if (!l->counts_clean) {
        OPAL_HIP(hipMemsetAsync(l->d_send_left, 0, LINK_HEADER_BYTES, st));
        OPAL_HIP(hipMemsetAsync(l->d_send_right, 0, LINK_HEADER_BYTES, st));
    }
    l->counts_clean = false;  // a push is about to count leavers into them
    return OPAL_OK;
}

// Moves `nbytes` of each send buffer to the neighbours and fills the recv buffers.
// whether exchanges may be issued on `st` rather than on the stream the link was created with: host-staged
// transports are synchronous (the library brackets them with copies on `st`), native ones follow set_stream
bool link_can_use_stream(const opal_link_t *l, hipStream_t st)
{
    return st == l->stream || !l->device_buffers || l->set_stream;
}

int link_exchange(opal_link_t *l, size_t nbytes, bool to_left, bool to_right, hipStream_t st)
{
    if (l->set_stream) l->set_stream(l->user, st);
    if (nbytes > l->nbytes) {
        set_error("link_exchange: %zu bytes exceed the link's message size %zu", nbytes, l->nbytes);
        return OPAL_ERR_ARG;
    }
    if (l->exchange_exact && nbytes == l->nbytes) {
        if (l->exchange_exact(l, to_left, to_right, st)) return OPAL_ERR_LINK;  // the callee has set the error text
        return OPAL_OK;
    }
    if (!l->device_buffers) {
        if (to_left) OPAL_HIP(hipMemcpyAsync(l->h_send_left, l->d_send_left, nbytes, hipMemcpyDeviceToHost, st));
        if (to_right) OPAL_HIP(hipMemcpyAsync(l->h_send_right, l->d_send_right, nbytes, hipMemcpyDeviceToHost, st));
        OPAL_HIP(hipStreamSynchronize(st));
    }
    char *sl = l->device_buffers ? l->d_send_left : l->h_send_left;
    char *sr = l->device_buffers ? l->d_send_right : l->h_send_right;
    char *rl = l->device_buffers ? l->d_recv_left : l->h_recv_left;
    char *rr = l->device_buffers ? l->d_recv_right : l->h_recv_right;
    int rc = l->exchange(l->user, to_left ? sl : nullptr, to_right ? sr : nullptr, to_left ? rl : nullptr,
                         to_right ? rr : nullptr, nbytes);
    if (rc) {
        set_error("exchange callback failed with %d", rc);
        return OPAL_ERR_LINK;
    }
    if (!l->device_buffers) {
        if (to_left) OPAL_HIP(hipMemcpyAsync(l->d_recv_left, l->h_recv_left, nbytes, hipMemcpyHostToDevice, st));
        if (to_right) OPAL_HIP(hipMemcpyAsync(l->d_recv_right, l->h_recv_right, nbytes, hipMemcpyHostToDevice, st));
    }
    return OPAL_OK;
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
    l->species_bytes = particle_capacity * (sizeof(int32_t) + (size_t)fields_per_particle * sizeof(double));
    l->nbytes = LINK_HEAD_BYTES + (size_t)species_slots * l->species_bytes;
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
    if (!(l->external && l->device_buffers)) {
        hipFree(l->d_send_left); hipFree(l->d_send_right); hipFree(l->d_recv_left); hipFree(l->d_recv_right);
    }
    if (!l->device_buffers && !l->external) {
        hipHostFree(l->h_send_left); hipHostFree(l->h_send_right); hipHostFree(l->h_recv_left); hipHostFree(l->h_recv_right);
    }
    delete l;
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
