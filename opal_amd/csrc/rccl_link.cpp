// Native neighbour transport: an opal_link_t whose exchange is four RCCL point-to-point calls in one group
// on the simulation's own HIP stream.  Compared with a host-language callback (opal_amd/exchange.py's
// TorchLink) nothing leaves the stream -- no second stream, no event hand-offs, no interpreter in the step
// loop -- which is worth ~50 us of a ~330 us step.  Each subdomain talks to its left and right neighbour
// only (a direct xGMI link per pair); there is no collective.
//
// RCCL is not linked: the library is opened at run time from the path the host passes (normally the copy
// PyTorch ships and has already loaded), so that libopal_amd.so itself has no RCCL dependency and a single
// RCCL instance lives in the process.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "link.hpp"

using namespace opal;

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int load_api(const char *path, RcclApi &api)
{
    const char *p = (path && *path) ? path : "librccl.so";
    api.handle = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) {
        set_error("cannot open RCCL at %s: %s", p, dlerror());
        return OPAL_ERR_LINK;
    }
    auto sym = [&](const char *name) { return dlsym(api.handle, name); };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send ||
        !api.Recv || !api.GetErrorString) {
        set_error("%s does not export the RCCL point-to-point API", p);
        return OPAL_ERR_LINK;
    }
    return OPAL_OK;
}

struct RcclState {
    RcclApi api;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int left = 0, right = 0;
};

#define OPAL_NCCL(st, call)                                                                \
    do {                                                                                   \
        ncclResult_t r_ = (call);                                                          \
        if (r_ != ncclSuccess) {                                                           \
            set_error("RCCL error %d (%s) in %s", (int)r_, (st)->api.GetErrorString(r_), #call); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

// send_left -> left neighbour's recv_right, send_right -> right neighbour's recv_left; the left edge moves
// n_left bytes each way, the right edge n_right (both ends of an edge pass the same number: link.cpp).  With two
// periodic subdomains (or one, sending to itself) both neighbours are the same peer: RCCL matches sends and
// receives per peer in issue order, so sends go (left, right) and receives (right, left).
int rccl_exchange2(void *user, const void *send_left, const void *send_right, void *recv_left, void *recv_right,
                   size_t n_left, size_t n_right)
{
    RcclState *st = static_cast<RcclState *>(user);
    OPAL_NCCL(st, st->api.GroupStart());
    if (send_left) OPAL_NCCL(st, st->api.Send(send_left, n_left, ncclUint8, st->left, st->comm, st->stream));
    if (send_right) OPAL_NCCL(st, st->api.Send(send_right, n_right, ncclUint8, st->right, st->comm, st->stream));
    if (recv_right) OPAL_NCCL(st, st->api.Recv(recv_right, n_right, ncclUint8, st->right, st->comm, st->stream));
    if (recv_left) OPAL_NCCL(st, st->api.Recv(recv_left, n_left, ncclUint8, st->left, st->comm, st->stream));
    OPAL_NCCL(st, st->api.GroupEnd());
    return 0;
}
int rccl_exchange(void *user, const void *send_left, const void *send_right, void *recv_left, void *recv_right,
                  size_t nbytes)
{
    return rccl_exchange2(user, send_left, send_right, recv_left, recv_right, nbytes, nbytes);
}

void rccl_free(void *user)
{
    RcclState *st = static_cast<RcclState *>(user);
    if (!st) return;
    if (st->comm) st->api.CommDestroy(st->comm);
    delete st;
}

}  // namespace

extern "C" {

int opal_rccl_unique_id(const char *librccl_path, void *id128)
{
    if (!id128) {
        set_error("opal_rccl_unique_id: null output");
        return OPAL_ERR_ARG;
    }
    RcclApi api;
    if (int rc = load_api(librccl_path, api)) return rc;
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) {
        set_error("ncclGetUniqueId failed: %s", api.GetErrorString(r));
        return OPAL_ERR_LINK;
    }
    static_assert(sizeof(ncclUniqueId) == OPAL_RCCL_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof id);
    return OPAL_OK;
}

int opal_link_create_rccl(opal_link_t **out, int rank, int size, size_t particle_capacity, int species_slots,
                          int fields_per_particle, const char *librccl_path, const void *unique_id, void *hip_stream)
{
    if (!out || !unique_id || size < 1 || rank < 0 || rank >= size) {
        set_error("opal_link_create_rccl: bad arguments");
        return OPAL_ERR_ARG;
    }
    RcclState *st = new RcclState();
    if (int rc = load_api(librccl_path, st->api)) {
        delete st;
        return rc;
    }
    st->stream = (hipStream_t)hip_stream;
    st->left = (rank - 1 + size) % size;
    st->right = (rank + 1) % size;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = st->api.CommInitRank(&st->comm, size, id, rank);  // collective over all subdomains
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank failed: %s", st->api.GetErrorString(r));
        delete st;
        return OPAL_ERR_LINK;
    }
    int rc = opal_link_create_layout(out, rank, size, 1, particle_capacity, species_slots, fields_per_particle,
                                     rccl_exchange, st, hip_stream);
    if (rc) {
        rccl_free(st);
        return rc;
    }
    (*out)->user_free = rccl_free;
    (*out)->exchange2 = rccl_exchange2;  // one size per edge: messages follow the particles (link.cpp)
    (*out)->set_stream = [](void *user, hipStream_t s) { static_cast<RcclState *>(user)->stream = s; };
    // the buffers are the library's own and RCCL is handed their addresses at every call: an edge can grow (link_grow) when
    // more particles stand near it than it has room for, exactly as a peer link's does
    (*out)->growable = !getenv("OPAL_LINK_NO_GROWTH");
    return OPAL_OK;
}

int opal_link_rccl_ranks(const opal_link_t *l, int *out)
{
    if (!l || !out || l->user_free != rccl_free) {
        set_error("opal_link_rccl_ranks: not a link created with opal_link_create_rccl");
        return OPAL_ERR_ARG;
    }
    const RcclState *st = static_cast<const RcclState *>(l->user);
    if (!st->api.CommCount) {
        set_error("this RCCL does not export ncclCommCount");
        return OPAL_ERR_LINK;
    }
    ncclResult_t r = st->api.CommCount(st->comm, out);
    if (r != ncclSuccess) {
        set_error("ncclCommCount failed: %s", st->api.GetErrorString(r));
        return OPAL_ERR_LINK;
    }
    return OPAL_OK;
}

}  // extern "C"
