// Internal view of opal_link_t (see link.cpp for the message layout).
#pragma once
#include "opal_internal.hpp"

namespace opal {
constexpr size_t LINK_HEADER_BYTES = 64;
constexpr size_t LINK_HALO_BYTES = 80 * sizeof(double);
constexpr size_t LINK_HEAD_BYTES = LINK_HEADER_BYTES + LINK_HALO_BYTES;  // 704
}  // namespace opal

struct opal_link {
    int rank, size;
    bool device_buffers;
    bool external;  // buffers the transport sees belong to the caller
    bool counts_clean;  // the send messages' leaver counts are known to be zero (see link_begin)
    size_t cap;     // particles per species per direction
    int nslots;     // species blocks per message: 2 (electrons, ions) or 3 (+ photons)
    int nfields;    // doubles per particle: 6 (x ux uy uz w prev_x) or 12 (+ chi work y z tau id)
    size_t species_bytes, nbytes;
    opal_exchange_fn exchange;
    void *user;
    void (*user_free)(void *);  // releases `user` with the link (native transports), or null
    // optional: ships a whole message but only the leavers it holds (see rccl_link.cpp); null = fixed size
    int (*exchange_exact)(opal_link *l, bool to_left, bool to_right, hipStream_t st);
    // optional: tells a stream-ordered transport which stream the next exchanges are ordered on; null = it only
    // knows the stream it was created with (e.g. a host-language callback tied to its framework's stream)
    void (*set_stream)(void *user, hipStream_t st);
    hipStream_t stream;
    char *d_send_left, *d_send_right, *d_recv_left, *d_recv_right;  // device side (kernels pack / unpack here)
    char *h_send_left, *h_send_right, *h_recv_left, *h_recv_right;  // host staging when device_buffers == 0
};

namespace opal {
size_t link_species_offset(const opal_link_t *l, int slot);
char *link_send(opal_link_t *l, int side);
char *link_recv(opal_link_t *l, int side);
int link_begin(opal_link_t *l, hipStream_t st);
int link_exchange(opal_link_t *l, size_t nbytes, bool to_left, bool to_right, hipStream_t st);
bool link_can_use_stream(const opal_link_t *l, hipStream_t st);
size_t link_body_bytes(const opal_link_t *l, const unsigned *counts);
int link_repack(opal_link_t *l, char *const strided[2], char *const compact[2], bool to_compact, hipStream_t st);
}  // namespace opal
