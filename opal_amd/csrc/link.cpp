// Neighbour exchange between subdomains (stands in for the reference's MPI
// point-to-point traffic: yee.rs:351-409 for the halo, particle/mod.rs:303-354
// for migrating particles).  The library packs fixed-size messages; the
// transport (RCCL, gloo, ...) is the callback the host supplied.
#include <hip/hip_runtime_api.h>

#include <cstring>

#include "opal_internal.hpp"

using namespace opal;

namespace opal {
int grid_pack_halo(opal_grid_t *g);
}

struct opal_link {
    int rank, size;
    bool device_buffers;
    size_t particle_capacity;
    opal_exchange_fn exchange;
    void *user;
    hipStream_t stream;
    size_t nbytes;
    char *send_left, *send_right, *recv_left, *recv_right;      // what the callback sees
    char *d_send_left, *d_send_right, *d_recv_left, *d_recv_right;  // device staging when host buffers are used
};

static const size_t HALO_BYTES = 80 * sizeof(double);

extern "C" {

int opal_link_create(opal_link_t **out, int rank, int size, int device_buffers, size_t particle_capacity,
                     opal_exchange_fn exchange, void *user, void *hip_stream)
{
    if (!out || !exchange || size < 1 || rank < 0 || rank >= size) {
        set_error("opal_link_create: bad arguments");
        return OPAL_ERR_ARG;
    }
    opal_link_t *l = new opal_link_t();
    memset((void *)l, 0, sizeof *l);
    l->rank = rank;
    l->size = size;
    l->device_buffers = device_buffers != 0;
    l->particle_capacity = particle_capacity;
    l->exchange = exchange;
    l->user = user;
    l->stream = (hipStream_t)hip_stream;
    l->nbytes = HALO_BYTES;
    char **dev[4] = {&l->d_send_left, &l->d_send_right, &l->d_recv_left, &l->d_recv_right};
    for (auto d : dev) {
        OPAL_HIP(hipMalloc((void **)d, l->nbytes));
        OPAL_HIP(hipMemset(*d, 0, l->nbytes));
    }
    if (l->device_buffers) {
        l->send_left = l->d_send_left; l->send_right = l->d_send_right;
        l->recv_left = l->d_recv_left; l->recv_right = l->d_recv_right;
    } else {
        char **host[4] = {&l->send_left, &l->send_right, &l->recv_left, &l->recv_right};
        for (auto h : host) OPAL_HIP(hipHostMalloc((void **)h, l->nbytes));
    }
    *out = l;
    return OPAL_OK;
}

int opal_link_free(opal_link_t *l)
{
    if (!l) return OPAL_OK;
    hipFree(l->d_send_left); hipFree(l->d_send_right); hipFree(l->d_recv_left); hipFree(l->d_recv_right);
    if (!l->device_buffers) {
        hipHostFree(l->send_left); hipHostFree(l->send_right); hipHostFree(l->recv_left); hipHostFree(l->recv_right);
    }
    delete l;
    return OPAL_OK;
}

size_t opal_link_message_bytes(const opal_link_t *l) { return l ? l->nbytes : 0; }

// yee.rs:358-409: this subdomain's first/last 8 cells go to its neighbours; what comes back is
// handed to the overlay kernel.  Returns device pointers to the received halos.
int opal_link_exchange_halo(opal_link_t *l, opal_grid_t *g, const double **recv_left, const double **recv_right)
{
    bool to_left = g->left_bdy == OPAL_BDY_INTERNAL, to_right = g->right_bdy == OPAL_BDY_INTERNAL;
    OPAL_HIP(hipMemcpyAsync(l->d_send_left, g->d_send, HALO_BYTES, hipMemcpyDeviceToDevice, g->stream));
    OPAL_HIP(hipMemcpyAsync(l->d_send_right, g->d_send + 80, HALO_BYTES, hipMemcpyDeviceToDevice, g->stream));
    if (!l->device_buffers) {
        OPAL_HIP(hipMemcpyAsync(l->send_left, l->d_send_left, HALO_BYTES, hipMemcpyDeviceToHost, g->stream));
        OPAL_HIP(hipMemcpyAsync(l->send_right, l->d_send_right, HALO_BYTES, hipMemcpyDeviceToHost, g->stream));
        OPAL_HIP(hipStreamSynchronize(g->stream));
    }
    int rc = l->exchange(l->user, to_left ? l->send_left : nullptr, to_right ? l->send_right : nullptr,
                         to_left ? l->recv_left : nullptr, to_right ? l->recv_right : nullptr, HALO_BYTES);
    if (rc) {
        set_error("exchange callback failed with %d", rc);
        return OPAL_ERR_LINK;
    }
    if (!l->device_buffers) {
        if (to_left) OPAL_HIP(hipMemcpyAsync(l->d_recv_left, l->recv_left, HALO_BYTES, hipMemcpyHostToDevice, g->stream));
        if (to_right) OPAL_HIP(hipMemcpyAsync(l->d_recv_right, l->recv_right, HALO_BYTES, hipMemcpyHostToDevice, g->stream));
    }
    *recv_left = (const double *)l->d_recv_left;
    *recv_right = (const double *)l->d_recv_right;
    return OPAL_OK;
}

}  // extern "C"
