// Compaction of a link message for the exact-size exchange (rccl_link.cpp): the push kernels fill each
// species' columns with a fixed stride of `particle_capacity` slots; before shipping, the occupied part of
// every column is copied into one contiguous body
//   for each species slot s with n_s > 0:  int32 cell[n_s rounded up to even]  double col_0[n_s] ... col_{F-1}[n_s]
// whose layout follows from the leaver counts in the message head alone, so the receiver can expand it back
// into strided columns (which is what the unpack kernel reads).
#include "link.hpp"

using namespace opal;

namespace {

struct PackArgs {
    char *strided[2];
    char *compact[2];
    size_t species_off[3];
    size_t cap;
    int nslots, nfields;
};

// one workgroup per (side, species slot); to_compact: strided -> compact, else the reverse
__global__ void k_link_repack(PackArgs a, int to_compact)
{
    const int side = blockIdx.x / a.nslots, slot = blockIdx.x % a.nslots;
    char *msg = a.strided[side];
    if (!msg) return;
    const unsigned *counts = reinterpret_cast<const unsigned *>(msg);  // the head travels (and arrives) first
    size_t body = 0;
    for (int s = 0; s < slot; s++) {
        size_t m = min((size_t)counts[s], a.cap);
        body += ((m + 1) & ~(size_t)1) * sizeof(int) + (size_t)a.nfields * m * sizeof(double);
    }
    const size_t n = min((size_t)counts[slot], a.cap);
    if (!n) return;
    char *blk = msg + a.species_off[slot];
    char *dst = a.compact[side] + body;
    int *cell_s = reinterpret_cast<int *>(blk), *cell_c = reinterpret_cast<int *>(dst);
    double *col_s = reinterpret_cast<double *>(blk + sizeof(int) * a.cap);
    double *col_c = reinterpret_cast<double *>(dst + ((n + 1) & ~(size_t)1) * sizeof(int));
    for (size_t k = threadIdx.x; k < n; k += blockDim.x) {
        if (to_compact) cell_c[k] = cell_s[k];
        else cell_s[k] = cell_c[k];
    }
    for (int f = 0; f < a.nfields; f++)
        for (size_t k = threadIdx.x; k < n; k += blockDim.x) {
            if (to_compact) col_c[(size_t)f * n + k] = col_s[(size_t)f * a.cap + k];
            else col_s[(size_t)f * a.cap + k] = col_c[(size_t)f * n + k];
        }
}

}  // namespace

namespace opal {

// bytes of the compact body for the given leaver counts
size_t link_body_bytes(const opal_link_t *l, const unsigned *counts)
{
    size_t body = 0;
    for (int s = 0; s < l->nslots; s++) {
        size_t m = counts[s] < l->cap ? counts[s] : l->cap;
        body += ((m + 1) & ~(size_t)1) * sizeof(int) + (size_t)l->nfields * m * sizeof(double);
    }
    return body;
}

// strided[side] <-> compact[side] for the sides whose pointers are non-null
int link_repack(opal_link_t *l, char *const strided[2], char *const compact[2], bool to_compact, hipStream_t st)
{
    PackArgs a;
    for (int k = 0; k < 2; k++) {
        a.strided[k] = strided[k];
        a.compact[k] = compact[k];
    }
    for (int s = 0; s < 3; s++) a.species_off[s] = link_species_offset(l, s);
    a.cap = l->cap;
    a.nslots = l->nslots;
    a.nfields = l->nfields;
    k_link_repack<<<2 * l->nslots, 256, 0, st>>>(a, to_compact ? 1 : 0);
    OPAL_HIP(hipGetLastError());
    return OPAL_OK;
}

}  // namespace opal
