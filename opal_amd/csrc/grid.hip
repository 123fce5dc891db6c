// YeeGrid on the GPU: SoA field/current arrays resident in HBM, halo pack +
// boundary-condition kernel, and the fused B(dt/2)-E(dt)-B(dt/2) stencil.
//
// HBM layout (doubles): two ping-pong field buffers d_eb[k][6][lenp] holding
// Ex Ey Ez Bx By Bz, and d_j[4][lenp] holding jx jy jz rho; lenp = len rounded
// up to 64 so every component row starts on a 512-byte boundary.  The field
// solve reads d_eb[cur] and writes d_eb[cur^1]: workgroups recompute a 2-cell
// halo in LDS instead of exchanging it, which is only race-free out of place.
// Roofline: HBM-bound, 9 reads + 6 writes of 8 B per cell per step (120 B/cell);
// at the BASELINE sizes (4 K - 33 K cells) it is launch-latency bound and costs
// < 1 % of a step.
#include "link.hpp"
#include "opal_internal.hpp"
#include "pic_device.hpp"
#include "grid_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace opal;

namespace {


// yee.rs:353-354: send_left = cell[0..8], send_right = cell[len-8..len]
__global__ void k_pack_halo(const double *__restrict__ eb, const double *__restrict__ jr, size_t lenp, int len,
                            double *__restrict__ send_left, double *__restrict__ send_right)
{
    int t = threadIdx.x;
    if (t >= 2 * HALO_DOUBLES) return;
    int side = t / HALO_DOUBLES, r = t % HALO_DOUBLES;
    int k = r / HALO_COMPS, comp = r % HALO_COMPS;
    int cellidx = side == 0 ? k : len - HALO_CELLS + k;
    double v;
    if (comp == 0) v = jr[3 * lenp + cellidx];                 // rho
    else if (comp < 4) v = jr[(size_t)(comp - 1) * lenp + cellidx];  // j
    else v = eb[(size_t)(comp - 4) * lenp + cellidx];          // E, B
    (side == 0 ? send_left : send_right)[r] = v;
}

__global__ void k_synchronize(SyncArgs a) { synchronize_body(a); }

constexpr int FT = 256;       // cells written per workgroup

// YeeGrid::advance (grid_device.hpp: advance_tile), one tile of FT cells per workgroup
__global__ __launch_bounds__(FT) void k_advance(AdvArgs a)
{
    __shared__ double lds[5 * (FT + 2 * ADV_HALO)];
    advance_tile<FT>(a, blockIdx.x * FT, a.len, threadIdx.x, lds);
}

__global__ void k_publish_only(PublishBatch pub) { publish_counters(pub); }

constexpr int ADV_LDS = 2 * 6 * (ADV_EDGE + 2 * ADV_HALO);  // doubles: the synchronize workgroup's two edge tiles (six components)
static_assert(ADV_LDS >= 5 * (ADV_TILE + 2 * ADV_HALO), "an interior tile fits the same array");

// synchronize + advance in ONE launch: workgroup 0 runs synchronize and then advances the cells it touched (and their
// stencil neighbours); the other workgroups advance the tiles in between, which depend on nothing synchronize writes
__global__ __launch_bounds__(ADV_TILE) void k_sync_advance(SyncArgs sa, AdvArgs a, PublishBatch pub)
{
    __shared__ double lds[ADV_LDS];
    // (the policy's counters leave from the last workgroup: off the synchronize workgroup's longer chain where there is another)
    if ((pub.n || pub.zero) && blockIdx.x == gridDim.x - 1) publish_counters(pub);
    if (blockIdx.x == 0) {
        sync_advance_edges(sa, a, lds);
    } else {
        advance_interior(a, (int)blockIdx.x - 1, lds);
    }
}

// yee.rs:787-809, interior cells only; one workgroup, fixed summation tree.
__global__ void k_field_energy(const double *__restrict__ eb, size_t lenp, int L, int size, double dx,
                               double *__restrict__ out)
{
    __shared__ double se[256], sb[256];
    int t = threadIdx.x;
    double e2 = 0.0, b2 = 0.0;
    for (int i = L + t; i < L + size; i += blockDim.x) {
        double ex = eb[0 * lenp + i], ey = eb[1 * lenp + i], ez = eb[2 * lenp + i];
        double bx = eb[3 * lenp + i], by = eb[4 * lenp + i], bz = eb[5 * lenp + i];
        e2 = e2 + ex * ex + ey * ey + ez * ez;
        b2 = b2 + bx * bx + by * by + bz * bz;
    }
    se[t] = e2;
    sb[t] = b2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) {
            se[t] += se[t + w];
            sb[t] += sb[t + w];
        }
        __syncthreads();
    }
    if (t == 0) out[0] = 0.5 * (VACUUM_PERMITTIVITY * se[0] + sb[0] / VACUUM_PERMEABILITY) * dx;
}

struct GlobalFields {
    const double *eb;
    size_t lenp;
    __device__ double operator()(int comp, int j) const { return eb[(size_t)comp * lenp + j]; }
};

// fields_at (yee.rs:499-529) for a batch of (cell, offset) pairs
__global__ void k_fields_at(const double *__restrict__ eb, size_t lenp, int L, int size, int n,
                            const int *__restrict__ c, const double *__restrict__ xi, double *__restrict__ E,
                            double *__restrict__ B, int *err)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if ((unsigned)c[i] >= (unsigned)size) {  // yee.rs:501-502
        atomicOr(err, 2);
        return;
    }
    GlobalFields f{eb, lenp};
    V3 e, b;
    gather(f, c[i] + L, xi[i], e, b);
    E[3 * i] = e.x; E[3 * i + 1] = e.y; E[3 * i + 2] = e.z;
    B[3 * i] = b.x; B[3 * i + 1] = b.y; B[3 * i + 2] = b.z;
}

int check_grid(const opal_grid_t *g)
{
    if (!g) {
        set_error("null grid handle");
        return OPAL_ERR_ARG;
    }
    return OPAL_OK;
}

}  // namespace

namespace opal {
int grid_apply_sync(opal_grid_t *g, const double *recv_left, const double *recv_right, double ly, double lz);
// used by particles.hip and link.cpp
int grid_pack_halo(opal_grid_t *g, double *send_left, double *send_right, hipStream_t st)
{
    k_pack_halo<<<1, 192, 0, st>>>(g->d_eb[g->cur], g->j(), g->lenp, (int)g->len, send_left, send_right);
    OPAL_HIP(hipGetLastError());
    return OPAL_OK;
}

// the arguments of synchronize's kernel, for a caller that runs it inside a launch of its own (grid_device.hpp)
void grid_sync_args(const opal_grid_t *g, const double *recv_left, const double *recv_right, double ly, double lz, SyncArgs &a)
{
    a.eb = g->d_eb[g->cur];
    a.jr = g->j();
    a.lenp = g->lenp;
    a.len = (int)g->len;
    a.L = g->L;
    a.R = g->R;
    a.recv_left = g->left_bdy == OPAL_BDY_INTERNAL ? recv_left : nullptr;
    a.recv_right = g->right_bdy == OPAL_BDY_INTERNAL ? recv_right : nullptr;
    a.self_periodic = g->ngrids == 1 && !g->self_link && g->left_bdy == OPAL_BDY_INTERNAL && g->right_bdy == OPAL_BDY_INTERNAL;
    a.left_bdy = g->left_bdy;
    a.right_bdy = g->right_bdy;
    a.laser_y = ly;
    a.laser_z = lz;
    a.damp = g->d_damp;
}

// The arguments of the field solve for a caller that runs it inside a launch that also runs synchronize, and the
// bookkeeping that follows such a launch (the ping-pong field buffers swap, the spare current block is zero again).
// The edges: synchronize writes cells [0, 8) and the last max(8, R) (ghost overlay, laser cell 2 or so, absorbing /
// conducting layer); a tile's stencil reaches 2 cells beyond what it writes.
void grid_advance_args(const opal_grid_t *g, double dt, AdvArgs &a)
{
    a.src = g->d_eb[g->cur];
    a.dst = g->d_eb[g->cur ^ 1];
    a.jr = g->j();
    a.jr_next = g->d_jb[g->jcur ^ 1];
    a.lenp = g->lenp;
    a.len = (int)g->len;
    a.dt = dt;
    a.dx = g->dx;
    const int len = (int)g->len;
    const int touched = std::max(std::max(g->R, 2 * GHOST_SIZE), g->L) + ADV_HALO;
    a.edge_left = std::min(64, len / 2);
    a.edge_right = (touched + 63) / 64 * 64;
    if (len - a.edge_left <= ADV_EDGE || a.edge_right > ADV_EDGE) a.edge_right = len - a.edge_left;  // no interior
}
int grid_interior_tiles(const AdvArgs &a)
{
    const int n = a.len - a.edge_left - a.edge_right;
    return n > 0 ? (n + ADV_TILE - 1) / ADV_TILE : 0;
}
void grid_advance_done(opal_grid_t *g)
{
    g->j_spare_zero = true;
    g->cur ^= 1;
}

// synchronize + advance in one launch
int grid_sync_advance(opal_grid_t *g, const double *recv_left, const double *recv_right, double ly, double lz, double dt,
                      const PublishBatch *pub)
{
    SyncArgs sa;
    grid_sync_args(g, recv_left, recv_right, ly, lz, sa);
    AdvArgs a;
    grid_advance_args(g, dt, a);
    if (a.edge_left > ADV_EDGE || a.edge_right > ADV_EDGE) {  // (a grid too short to have an interior, yet longer than one workgroup's edges)
        if (pub && (pub->n || pub->zero)) k_publish_only<<<1, 64, 0, g->stream>>>(*pub);
        if (int rc = grid_apply_sync(g, recv_left, recv_right, ly, lz)) return rc;
        return opal_grid_advance(g, dt);
    }
    PublishBatch none{};
    prof_begin(KERNEL_FIELD, g->stream);
    k_sync_advance<<<1 + grid_interior_tiles(a), ADV_TILE, 0, g->stream>>>(sa, a, pub ? *pub : none);
    prof_end(KERNEL_FIELD, g->stream);
    OPAL_HIP(hipGetLastError());
    grid_advance_done(g);
    return OPAL_OK;
}

int grid_apply_sync(opal_grid_t *g, const double *recv_left, const double *recv_right, double ly, double lz)
{
    SyncArgs a;
    grid_sync_args(g, recv_left, recv_right, ly, lz, a);
    k_synchronize<<<1, 256, 0, g->stream>>>(a);
    OPAL_HIP(hipGetLastError());
    return OPAL_OK;
}

// packs this subdomain's halo into the link's messages; with `exchange` also moves the halo part now
// (yee.rs:358-409).  Returns device pointers to the received halos.
int grid_exchange_halo(opal_grid_t *g, opal_link_t *link, bool exchange, const double **recv_left, const double **recv_right,
                       hipStream_t st)
{
    if (!st) st = g->stream;
    double *sl = reinterpret_cast<double *>(link_send(link, 0) + LINK_HEADER_BYTES);
    double *sr = reinterpret_cast<double *>(link_send(link, 1) + LINK_HEADER_BYTES);
    if (int rc = grid_pack_halo(g, sl, sr, st)) return rc;
    if (exchange) {
        if (int rc = link_exchange_halo(link, g->left_bdy == OPAL_BDY_INTERNAL, g->right_bdy == OPAL_BDY_INTERNAL, st))
            return rc;
    }
    *recv_left = reinterpret_cast<const double *>(link_recv(link, 0) + LINK_HEADER_BYTES);
    *recv_right = reinterpret_cast<const double *>(link_recv(link, 1) + LINK_HEADER_BYTES);
    return OPAL_OK;
}

// The rows of a subdomain whose left cut has moved by dL cells, without any of its neighbours' data (sim.cpp: cuts that drift
// a cell at a time): array index j of the new array is index j + dL of the old one.  What a one-cell move uncovers is a cell
// the subdomain already holds -- a ghost cell, whose E and B the last field solve computed as the neighbour did (the two
// nearest ghost cells either side are exact after a step: yee.rs:1-58) -- and what falls off the far end of the array are
// outermost ghosts, which the next synchronize overwrites before anything reads them.
__global__ void k_grid_shift(const double *__restrict__ eb_in, double *__restrict__ eb_out, const double *__restrict__ j_in,
                             double *__restrict__ j_out, size_t lenp_in, size_t lenp_out, int old_len, int new_len, int dL)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= new_len) return;
    const int o = j + dL;
    const bool in = o >= 0 && o < old_len;
#pragma unroll
    for (int c = 0; c < 6; c++) eb_out[(size_t)c * lenp_out + j] = in ? eb_in[(size_t)c * lenp_in + o] : 0.0;
#pragma unroll
    for (int c = 0; c < 4; c++) j_out[(size_t)c * lenp_out + j] = in ? j_in[(size_t)c * lenp_in + o] : 0.0;
}

int grid_recut(opal_grid_t *g, size_t size, double offset, const double *rows);

int grid_shift(opal_grid_t *g, int dL, size_t new_size, double new_offset)
{
    const size_t new_len = new_size + (size_t)g->L + (size_t)g->R;
    if (new_size < (size_t)(2 * GHOST_SIZE)) {
        set_error("grid_shift: %zu cells fall below the minimum", new_size);
        return OPAL_ERR_ARG;
    }
    if (new_len > g->lenp) {
        // the subdomain has outgrown its rows (the last one of a domain whose cuts all drift away from it gains a cell a step):
        // longer rows -- a quarter to spare again --, the old ones shifted into them
        const size_t lenp = (new_len + new_len / 4 + 256 + 63) / 64 * 64, nb = sizeof(double) * lenp;
        double *eb[2] = {nullptr, nullptr}, *jb[2] = {nullptr, nullptr}, *pin = nullptr;
        hipError_t e = hipSuccess;
        for (int k = 0; k < 2 && e == hipSuccess; k++) {
            e = hipMalloc(&eb[k], 6 * nb);
            if (e == hipSuccess) e = hipMalloc(&jb[k], 4 * nb);
            if (e == hipSuccess) e = hipMemsetAsync(eb[k], 0, 6 * nb, g->stream);
            if (e == hipSuccess) e = hipMemsetAsync(jb[k], 0, 4 * nb, g->stream);
        }
        if (e == hipSuccess) e = hipHostMalloc(&pin, sizeof(double) * 11 * lenp);
        if (e == hipSuccess) {
            k_grid_shift<<<(unsigned)((new_len + 255) / 256), 256, 0, g->stream>>>(g->d_eb[g->cur], eb[0], g->d_jb[g->jcur], jb[0], g->lenp,
                                                                                  lenp, (int)g->len, (int)new_len, dL);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) {
            for (int k = 0; k < 2; k++) {
                if (eb[k]) hipFree(eb[k]);
                if (jb[k]) hipFree(jb[k]);
            }
            if (pin) hipHostFree(pin);
            return hip_fail(e, "grid_shift: longer rows");
        }
        hipFree(g->d_eb[0]); hipFree(g->d_eb[1]); hipFree(g->d_jb[0]); hipFree(g->d_jb[1]);
        hipHostFree(g->h_pin);
        g->d_eb[0] = eb[0]; g->d_eb[1] = eb[1]; g->d_jb[0] = jb[0]; g->d_jb[1] = jb[1];
        g->h_pin = pin;
        g->lenp = lenp;
        g->cur = 0;
        g->jcur = 0;
        g->j_spare_zero = true;
        g->size = new_size;
        g->len = new_len;
        g->offset = new_offset;
        return OPAL_OK;
    }
    if (!g->j_spare_zero) {  // (between a field solve and the next clear the spare current rows are zero: normally the case)
        OPAL_HIP(hipMemsetAsync(g->d_jb[g->jcur ^ 1], 0, sizeof(double) * 4 * g->lenp, g->stream));
        g->j_spare_zero = true;
    }
    k_grid_shift<<<(unsigned)((new_len + 255) / 256), 256, 0, g->stream>>>(g->d_eb[g->cur], g->d_eb[g->cur ^ 1], g->d_jb[g->jcur],
                                                                          g->d_jb[g->jcur ^ 1], g->lenp, g->lenp, (int)g->len, (int)new_len, dL);
    OPAL_HIP(hipGetLastError());
    // the rows just left become the spares: the old current rows zeroed (the next clear flips to them)
    OPAL_HIP(hipMemsetAsync(g->d_jb[g->jcur], 0, sizeof(double) * 4 * g->lenp, g->stream));
    g->cur ^= 1;
    g->jcur ^= 1;
    g->size = new_size;
    g->len = new_len;
    g->offset = new_offset;
    return OPAL_OK;
}

// A subdomain whose cuts have moved (sim.cpp: sim_recut): `size` interior cells from `offset` on, boundary regions as they
// were (yee.rs:278-292 goes by rank and boundary kind, which do not change), and the 10 rows rho jx jy jz Ex Ey Ez Bx By Bz of
// the new array -- `rows`, host memory, new_len doubles each -- as the state.  The handle stays the one the caller holds.
int grid_recut(opal_grid_t *g, size_t size, double offset, const double *rows)
{
    if (size < (size_t)(2 * GHOST_SIZE) || size > (size_t)1 << 28) {
        set_error("grid_recut: subdomain of %zu cells (min %d)", size, 2 * GHOST_SIZE);
        return OPAL_ERR_ARG;
    }
    OPAL_HIP(hipStreamSynchronize(g->stream));
    const size_t len = size + (size_t)g->L + (size_t)g->R;
    if (len <= g->lenp) {
        // the rows have room (a decomposed grid is built with some to spare, opal_grid_build): the same buffers, the same
        // row stride, new contents -- no allocation, no pinned-memory call on the path of a re-cut
        // (the pinned scratch as an image of the device rows -- jx jy jz rho, then Ex .. Bz, each lenp long: two copies)
        const size_t P = g->lenp, nb = sizeof(double) * P;
        double *pin = g->h_pin;
        memset(pin, 0, 10 * nb);
        for (int c = 0; c < 3; c++) memcpy(pin + (size_t)c * P, rows + (size_t)(1 + c) * len, sizeof(double) * len);
        memcpy(pin + 3 * P, rows, sizeof(double) * len);  // rho
        for (int c = 0; c < 6; c++) memcpy(pin + (size_t)(4 + c) * P, rows + (size_t)(4 + c) * len, sizeof(double) * len);
        OPAL_HIP(hipMemcpyAsync(g->d_jb[0], pin, 4 * nb, hipMemcpyHostToDevice, g->stream));
        OPAL_HIP(hipMemcpyAsync(g->d_eb[0], pin + 4 * P, 6 * nb, hipMemcpyHostToDevice, g->stream));
        OPAL_HIP(hipMemsetAsync(g->d_eb[1], 0, 6 * nb, g->stream));
        OPAL_HIP(hipMemsetAsync(g->d_jb[1], 0, 4 * nb, g->stream));
        OPAL_HIP(hipStreamSynchronize(g->stream));  // (the staging area is the grid's own scratch: free again)
    } else {
    const size_t lenp = (len + len / 4 + 256 + 63) / 64 * 64, nb = sizeof(double) * lenp;
    double *eb[2] = {nullptr, nullptr}, *jb[2] = {nullptr, nullptr}, *pin = nullptr;
    auto drop = [&]() {
        for (int k = 0; k < 2; k++) {
            if (eb[k]) hipFree(eb[k]);
            if (jb[k]) hipFree(jb[k]);
        }
        if (pin) hipHostFree(pin);
    };
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; k++) {
        e = hipMalloc(&eb[k], 6 * nb);
        if (e == hipSuccess) e = hipMalloc(&jb[k], 4 * nb);
        if (e == hipSuccess) e = hipMemsetAsync(eb[k], 0, 6 * nb, g->stream);
        if (e == hipSuccess) e = hipMemsetAsync(jb[k], 0, 4 * nb, g->stream);
    }
    if (e == hipSuccess) e = hipHostMalloc(&pin, sizeof(double) * 11 * lenp);
    if (e != hipSuccess) {
        drop();
        return hip_fail(e, "grid_recut: allocating the re-cut subdomain's rows");
    }
    // (rows are staged through the new pinned scratch: the caller's are pageable)
    memcpy(pin, rows, sizeof(double) * 10 * len);
    const size_t lb = sizeof(double) * len;
    e = hipMemcpyAsync(jb[0] + 3 * lenp, pin, lb, hipMemcpyHostToDevice, g->stream);  // rho
    for (int c = 0; c < 3 && e == hipSuccess; c++) e = hipMemcpyAsync(jb[0] + (size_t)c * lenp, pin + (size_t)(1 + c) * len, lb, hipMemcpyHostToDevice, g->stream);
    for (int c = 0; c < 6 && e == hipSuccess; c++) e = hipMemcpyAsync(eb[0] + (size_t)c * lenp, pin + (size_t)(4 + c) * len, lb, hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) {
        drop();
        return hip_fail(e, "grid_recut: uploading the re-cut subdomain's rows");
    }
    hipFree(g->d_eb[0]); hipFree(g->d_eb[1]); hipFree(g->d_jb[0]); hipFree(g->d_jb[1]);
    hipHostFree(g->h_pin);
    g->d_eb[0] = eb[0]; g->d_eb[1] = eb[1]; g->d_jb[0] = jb[0]; g->d_jb[1] = jb[1];
    g->h_pin = pin;
    g->lenp = lenp;
    }
    g->cur = 0;
    g->jcur = 0;
    g->j_spare_zero = true;
    g->size = size;
    g->len = len;
    g->offset = offset;
    // a transverse field that came with the new cells ends the electrostatic launches here too (opal_grid_set_raw)
    for (int c = 5; c < 10; c++)
        for (size_t i = 0; i < len; i++)
            if (rows[(size_t)c * len + i] != 0.0) *g->h_esflag = 1;
    if (g->right_bdy == OPAL_BDY_ABSORBING) {
        // yee.rs:465-468 with the new offset: the same cells, the same factors up to the rounding of cell_x
        double xmin_b = g->cell_x(g->size + (size_t)g->L);
        double xmax_b = g->cell_x(g->size + (size_t)g->L + (size_t)g->R - 1);
        double sigma_max = 10.0 / ((double)g->R);
        for (size_t i = g->len - (size_t)g->R + 1, k = 0; i < g->len; i++, k++)
            g->damp_host[k] = 1.0 - sigma_max * (g->cell_x(i) - xmin_b) / (xmax_b - xmin_b);
        OPAL_HIP(hipMemcpy(g->d_damp, g->damp_host.data(), sizeof(double) * g->damp_host.size(), hipMemcpyHostToDevice));
    }
    return OPAL_OK;
}

int grid_check_err(const opal_grid_t *g)
{
    int err = 0;
    OPAL_HIP(hipMemcpyAsync(&err, g->d_err, sizeof(int), hipMemcpyDeviceToHost, g->stream));
    OPAL_HIP(hipStreamSynchronize(g->stream));
    if (err) {
        set_error("grid invariant violated on device (flags 0x%x): fields_at index out of range (yee.rs:501-502)", err);
        return OPAL_ERR_ASSERT;
    }
    return OPAL_OK;
}
}  // namespace opal

extern "C" {

size_t opal_grid_min_size(void) { return (size_t)(2 * GHOST_SIZE); }

int opal_grid_build(opal_grid_t **out, int id, int numtasks, int left, int right, double xmin, double dx,
                    const size_t *nx, const double *offset, void *hip_stream)
{
    if (!out || !nx || !offset || numtasks < 1 || id < 0 || id >= numtasks) {
        set_error("opal_grid_build: bad arguments");
        return OPAL_ERR_ARG;
    }
    if (nx[id] < opal_grid_min_size() || nx[id] > (size_t)1 << 28) {
        set_error("opal_grid_build: subdomain of %zu cells (min %zu)", nx[id], opal_grid_min_size());
        return OPAL_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("opal_grid_build: no HIP device available (the HIP path has no CPU fallback)");
        return OPAL_ERR_HIP;
    }
    opal_grid_t *g = new opal_grid_t();
    // yee.rs:278-292
    if (id == 0 && left == OPAL_BDY_LASER) { g->left_bdy = OPAL_BDY_LASER; g->L = LASER_BDY_SIZE; }
    else { g->left_bdy = OPAL_BDY_INTERNAL; g->L = GHOST_SIZE; }
    if (id == numtasks - 1 && right == OPAL_BDY_ABSORBING) { g->right_bdy = OPAL_BDY_ABSORBING; g->R = ABSORBING_BDY_SIZE; }
    else if (id == numtasks - 1 && right == OPAL_BDY_CONDUCTING) { g->right_bdy = OPAL_BDY_CONDUCTING; g->R = CONDUCTING_BDY_SIZE; }
    else { g->right_bdy = OPAL_BDY_INTERNAL; g->R = GHOST_SIZE; }
    g->id = id;
    g->ngrids = numtasks;
    g->rank_to_left = (id - 1 + numtasks) % numtasks;
    g->rank_to_right = (id + 1) % numtasks;
    g->size = nx[id];
    g->dx = dx;
    g->xmin_global = xmin;
    g->offset = offset[id];
    g->len = g->size + (size_t)g->L + (size_t)g->R;
    g->lenp = (g->len + 63) / 64 * 64;
    // one of several subdomains: rows with room to spare, so that cuts that move (sim.cpp: sim_recut) rewrite the rows they
    // have instead of allocating new ones (a few KB per row)
    if (numtasks > 1) g->lenp = (g->len + g->len / 4 + 256 + 63) / 64 * 64;
    g->stream = (hipStream_t)hip_stream;
    g->cur = 0;
    g->d_damp = nullptr;
    size_t nb = sizeof(double) * g->lenp;
    OPAL_HIP(hipMalloc(&g->d_eb[0], 6 * nb));
    OPAL_HIP(hipMalloc(&g->d_eb[1], 6 * nb));
    OPAL_HIP(hipMalloc(&g->d_jb[0], 4 * nb));
    OPAL_HIP(hipMalloc(&g->d_jb[1], 4 * nb));
    g->jcur = 0;
    g->j_spare_zero = true;
    OPAL_HIP(hipMalloc(&g->d_send, sizeof(double) * 2 * HALO_DOUBLES));
    OPAL_HIP(hipMalloc(&g->d_scalar, sizeof(double) * 8));
    OPAL_HIP(hipMalloc(&g->d_err, sizeof(int)));
    OPAL_HIP(hipHostMalloc(&g->h_pin, sizeof(double) * 11 * g->lenp));
    OPAL_HIP(hipHostMalloc(&g->h_esflag, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    *g->h_esflag = 0;  // the fields start at zero
    OPAL_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&g->d_esflag), g->h_esflag, 0));
    OPAL_HIP(hipMemsetAsync(g->d_eb[0], 0, 6 * nb, g->stream));
    OPAL_HIP(hipMemsetAsync(g->d_eb[1], 0, 6 * nb, g->stream));
    OPAL_HIP(hipMemsetAsync(g->d_jb[0], 0, 4 * nb, g->stream));
    OPAL_HIP(hipMemsetAsync(g->d_jb[1], 0, 4 * nb, g->stream));
    OPAL_HIP(hipMemsetAsync(g->d_err, 0, sizeof(int), g->stream));
    if (g->right_bdy == OPAL_BDY_ABSORBING) {
        // yee.rs:465-468, evaluated once on the host with the reference's arithmetic
        double xmin_b = g->cell_x(g->size + (size_t)g->L);
        double xmax_b = g->cell_x(g->size + (size_t)g->L + (size_t)g->R - 1);
        double sigma_max = 10.0 / ((double)g->R);
        g->damp_host.resize((size_t)g->R - 1);
        for (size_t i = g->len - (size_t)g->R + 1, k = 0; i < g->len; i++, k++) {
            double sigma = sigma_max * (g->cell_x(i) - xmin_b) / (xmax_b - xmin_b);
            g->damp_host[k] = 1.0 - sigma;
        }
        OPAL_HIP(hipMalloc(&g->d_damp, sizeof(double) * g->damp_host.size()));
        OPAL_HIP(hipMemcpyAsync(g->d_damp, g->damp_host.data(), sizeof(double) * g->damp_host.size(),
                                hipMemcpyHostToDevice, g->stream));
    }
    OPAL_HIP(hipStreamSynchronize(g->stream));
    *out = g;
    return OPAL_OK;
}

int opal_grid_free(opal_grid_t *g)
{
    if (!g) return OPAL_OK;
    hipFree(g->d_eb[0]);
    hipFree(g->d_eb[1]);
    hipFree(g->d_jb[0]);
    hipFree(g->d_jb[1]);
    hipFree(g->d_send);
    hipFree(g->d_scalar);
    hipFree(g->d_err);
    if (g->d_damp) hipFree(g->d_damp);
    hipHostFree(g->h_pin);
    hipHostFree(g->h_esflag);
    delete g;
    return OPAL_OK;
}

int opal_grid_rank(const opal_grid_t *g) { return g ? g->id : -1; }
int opal_grid_ngrids(const opal_grid_t *g) { return g ? g->ngrids : 0; }
int opal_grid_set_self_exchange(opal_grid_t *g, int on)
{
    if (int rc = check_grid(g)) return rc;
    if (on && (g->ngrids != 1 || g->left_bdy != OPAL_BDY_INTERNAL || g->right_bdy != OPAL_BDY_INTERNAL)) {
        set_error("opal_grid_set_self_exchange: only a lone periodic subdomain is its own neighbour");
        return OPAL_ERR_ARG;
    }
    g->self_link = on != 0;
    return OPAL_OK;
}
int opal_grid_to_left(const opal_grid_t *g) { return g && g->left_bdy == OPAL_BDY_INTERNAL ? g->rank_to_left : -1; }
int opal_grid_to_right(const opal_grid_t *g) { return g && g->right_bdy == OPAL_BDY_INTERNAL ? g->rank_to_right : -1; }
double opal_grid_xmin(const opal_grid_t *g) { return g ? g->cell_x((size_t)g->L) : 0.0; }
double opal_grid_dx(const opal_grid_t *g) { return g ? g->dx : 0.0; }
size_t opal_grid_size(const opal_grid_t *g) { return g ? g->size : 0; }
size_t opal_grid_len(const opal_grid_t *g) { return g ? g->len : 0; }
size_t opal_grid_left_bdy_size(const opal_grid_t *g) { return g ? (size_t)g->L : 0; }
size_t opal_grid_right_bdy_size(const opal_grid_t *g) { return g ? (size_t)g->R : 0; }

int opal_grid_clear(opal_grid_t *g)
{
    if (int rc = check_grid(g)) return rc;
    if (g->j_spare_zero) {
        // the last field solve already zeroed the spare buffer: switch to it
        g->jcur ^= 1;
        g->j_spare_zero = false;
    } else {
        OPAL_HIP(hipMemsetAsync(g->j(), 0, sizeof(double) * 4 * g->lenp, g->stream));
    }
    return OPAL_OK;
}

int opal_grid_advance(opal_grid_t *g, double dt)
{
    if (int rc = check_grid(g)) return rc;
    int nblk = (int)((g->len + FT - 1) / FT);
    AdvArgs a;
    grid_advance_args(g, dt, a);
    a.edge_left = a.edge_right = 0;
    prof_begin(KERNEL_FIELD, g->stream);
    k_advance<<<nblk, FT, 0, g->stream>>>(a);
    prof_end(KERNEL_FIELD, g->stream);
    OPAL_HIP(hipGetLastError());
    grid_advance_done(g);
    return OPAL_OK;
}

int opal_grid_fields_at(const opal_grid_t *g, size_t n, const int32_t *c, const double *xi, double *E, double *B)
{
    if (int rc = check_grid(g)) return rc;
    if (n == 0) return OPAL_OK;
    int *dc;
    double *dxi, *dE, *dB;
    OPAL_HIP(hipMalloc(&dc, n * sizeof(int)));
    OPAL_HIP(hipMalloc(&dxi, n * sizeof(double)));
    OPAL_HIP(hipMalloc(&dE, 3 * n * sizeof(double)));
    OPAL_HIP(hipMalloc(&dB, 3 * n * sizeof(double)));
    OPAL_HIP(hipMemcpyAsync(dc, c, n * sizeof(int), hipMemcpyHostToDevice, g->stream));
    OPAL_HIP(hipMemcpyAsync(dxi, xi, n * sizeof(double), hipMemcpyHostToDevice, g->stream));
    k_fields_at<<<(unsigned)((n + 255) / 256), 256, 0, g->stream>>>(g->d_eb[g->cur], g->lenp, g->L, (int)g->size, (int)n,
                                                                   dc, dxi, dE, dB, g->d_err);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpyAsync(E, dE, 3 * n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    OPAL_HIP(hipMemcpyAsync(B, dB, 3 * n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    int rc = grid_check_err(g);
    hipFree(dc); hipFree(dxi); hipFree(dE); hipFree(dB);
    if (rc) OPAL_HIP(hipMemsetAsync(g->d_err, 0, sizeof(int), g->stream));
    return rc;
}

int opal_grid_em_field_energy(const opal_grid_t *g, double *energy)
{
    if (int rc = check_grid(g)) return rc;
    k_field_energy<<<1, 256, 0, g->stream>>>(g->d_eb[g->cur], g->lenp, g->L, (int)g->size, g->dx, g->d_scalar);
    OPAL_HIP(hipGetLastError());
    OPAL_HIP(hipMemcpyAsync(energy, g->d_scalar, sizeof(double), hipMemcpyDeviceToHost, g->stream));
    OPAL_HIP(hipStreamSynchronize(g->stream));
    return OPAL_OK;
}

static int grid_fetch(const opal_grid_t *g)
{
    // h_pin rows: 0 rho? keep opal_comp order 1..10 (x is computed)
    double *h = g->h_pin;
    size_t nb = sizeof(double) * g->lenp;
    OPAL_HIP(hipMemcpyAsync(h + 1 * g->lenp, g->j() + 3 * g->lenp, nb, hipMemcpyDeviceToHost, g->stream));       // rho
    OPAL_HIP(hipMemcpyAsync(h + 2 * g->lenp, g->j(), 3 * nb, hipMemcpyDeviceToHost, g->stream));                  // jx jy jz
    OPAL_HIP(hipMemcpyAsync(h + 5 * g->lenp, g->d_eb[g->cur], 6 * nb, hipMemcpyDeviceToHost, g->stream));         // E, B
    OPAL_HIP(hipStreamSynchronize(g->stream));
    for (size_t i = 0; i < g->len; i++) h[i] = g->cell_x(i);
    return OPAL_OK;
}

int opal_grid_get_raw(const opal_grid_t *g, int comp, double *out)
{
    if (int rc = check_grid(g)) return rc;
    if (comp < 0 || comp > 10 || !out) {
        set_error("opal_grid_get_raw: bad component %d", comp);
        return OPAL_ERR_ARG;
    }
    if (comp == OPAL_X) {
        for (size_t i = 0; i < g->len; i++) out[i] = g->cell_x(i);
        return OPAL_OK;
    }
    const double *src = comp == OPAL_RHO ? g->j() + 3 * g->lenp
                        : comp <= OPAL_JZ ? g->j() + (size_t)(comp - OPAL_JX) * g->lenp
                                          : g->d_eb[g->cur] + (size_t)(comp - OPAL_EX) * g->lenp;
    OPAL_HIP(hipMemcpyAsync(out, src, sizeof(double) * g->len, hipMemcpyDeviceToHost, g->stream));
    OPAL_HIP(hipStreamSynchronize(g->stream));
    return OPAL_OK;
}

int opal_grid_set_raw(opal_grid_t *g, int comp, const double *in)
{
    if (int rc = check_grid(g)) return rc;
    if (comp < 1 || comp > 10 || !in) {
        set_error("opal_grid_set_raw: bad component %d", comp);
        return OPAL_ERR_ARG;
    }
    double *dst = comp == OPAL_RHO ? g->j() + 3 * g->lenp
                  : comp <= OPAL_JZ ? g->j() + (size_t)(comp - OPAL_JX) * g->lenp
                                    : g->d_eb[g->cur] + (size_t)(comp - OPAL_EX) * g->lenp;
    OPAL_HIP(hipMemcpyAsync(dst, in, sizeof(double) * g->len, hipMemcpyHostToDevice, g->stream));
    OPAL_HIP(hipStreamSynchronize(g->stream));
    if (comp > OPAL_EX)  // a transverse component: Ey, Ez or any of B
        for (size_t i = 0; i < g->len; i++)
            if (in[i] != 0.0) *g->h_esflag = 1;
    return OPAL_OK;
}

int opal_grid_assume_electrostatic(opal_grid_t *g)
{
    if (int rc = check_grid(g)) return rc;
    OPAL_HIP(hipStreamSynchronize(g->stream));  // no kernel that could still raise the flag is in flight
    *g->h_esflag = 0;
    return OPAL_OK;
}

int opal_grid_interpolate(const opal_grid_t *g, double *out)
{
    if (int rc = check_grid(g)) return rc;
    if (int rc = grid_fetch(g)) return rc;
    const double *h = g->h_pin;
    size_t P = g->lenp;
    auto at = [&](int comp, size_t i) { return h[(size_t)comp * P + i]; };
    size_t lo = (size_t)g->L - 1;
    for (size_t k = 0; k < g->size; k++) {  // yee.rs:821-832
        size_t a = lo + k, b = lo + k + 1;
        double *o = out + 11 * k;
        o[0] = at(OPAL_X, b);
        o[1] = at(OPAL_RHO, b);
        o[2] = 0.5 * (at(OPAL_JX, a) + at(OPAL_JX, b));
        o[3] = at(OPAL_JY, b);
        o[4] = at(OPAL_JZ, b);
        o[5] = 0.5 * (at(OPAL_EX, a) + at(OPAL_EX, b));
        o[6] = at(OPAL_EY, b);
        o[7] = at(OPAL_EZ, b);
        o[8] = at(OPAL_BX, b);
        o[9] = 0.5 * (at(OPAL_BY, a) + at(OPAL_BY, b));
        o[10] = 0.5 * (at(OPAL_BZ, a) + at(OPAL_BZ, b));
    }
    return OPAL_OK;
}

int opal_grid_synchronize(opal_grid_t *g, opal_link_t *link, opal_fn2 laser_y, opal_fn2 laser_z, void *user,
                          double t, double dt)
{
    if (int rc = check_grid(g)) return rc;
    double ly = 0.0, lz = 0.0;
    if (g->left_bdy == OPAL_BDY_LASER) {  // yee.rs:456-462
        if (!laser_y || !laser_z) {
            set_error("opal_grid_synchronize: laser boundary needs laser_y and laser_z");
            return OPAL_ERR_ARG;
        }
        double x = g->cell_x((size_t)(g->L - 2));
        double r = SPEED_OF_LIGHT * dt / g->dx;
        ly = 2.0 * r * laser_y(t, x, user);
        lz = 2.0 * r * laser_z(t, x, user);
    }
    bool need_halo = g->left_bdy == OPAL_BDY_INTERNAL || g->right_bdy == OPAL_BDY_INTERNAL;
    const double *recv_left = nullptr, *recv_right = nullptr;
    if (need_halo && g->ngrids == 1 && !g->self_link) {
        // yee.rs:365-370, 390-395: a lone periodic subdomain receives its own sends; the overlay
        // kernel reads them straight from the arrays
    } else if (need_halo) {
        if (!link) {
            set_error("opal_grid_synchronize: %d subdomains need a link", g->ngrids);
            return OPAL_ERR_ARG;
        }
        if (int rc = grid_exchange_halo(g, link, true, &recv_left, &recv_right, nullptr)) return rc;
    }
    return grid_apply_sync(g, recv_left, recv_right, ly, lz);
}

}  // extern "C"

namespace opal {
// all rows of the subdomain on the host (pinned scratch of the grid): [comp * lenp + i], comp = OPAL_X .. OPAL_BZ
const double *grid_fetch_rows(const opal_grid_t *g) { return grid_fetch(g) == OPAL_OK ? g->h_pin : nullptr; }
}
