// Diagnostic: what the push kernel's ACCESS PATTERN alone can reach on this GPU -- 44 B read (cell:i32, x, ux,
// uy, uz, w) and 36 B written back in place (cell, x, ux, uy, uz) per particle slot, non-temporal, one
// workgroup per 256 x PPT consecutive slots, a trivial amount of arithmetic -- with one slot per lane and
// access (4/8-byte accesses, what k_push does) against two or four adjacent slots per lane (8/16-byte
// accesses).  Prices the 80 B/update roofline against what HBM delivers for this mix.
//   hipcc -O3 --offload-arch=gfx950 scripts/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

template <class T>
__device__ __forceinline__ T ldnt(const T *p) { return __builtin_nontemporal_load(p); }
template <class T>
__device__ __forceinline__ void stnt(T *p, T v) { __builtin_nontemporal_store(v, p); }

typedef double double2v __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));

// V slots per lane and access
template <int V, int PPT>
__global__ __launch_bounds__(256) void k_stream(int *cell, double *x, double *ux, double *uy, double *uz, const double *w, size_t n)
{
    const size_t chunk0 = (size_t)blockIdx.x * 256 * PPT;
    for (int p = 0; p < PPT / V; p++) {
        size_t i = chunk0 + ((size_t)p * 256 + threadIdx.x) * V;
        if (i + V > n) return;
        if (V == 1) {
            int c = ldnt(cell + i);
            double a = ldnt(x + i), b = ldnt(ux + i), d = ldnt(uy + i), e = ldnt(uz + i), ww = ldnt(w + i);
            a += 1e-9 * ww; b += 1e-9 * a; d += 1e-9 * b; e += 1e-9 * d;
            stnt(cell + i, c + (a > 1e30));
            stnt(x + i, a); stnt(ux + i, b); stnt(uy + i, d); stnt(uz + i, e);
        } else if (V == 2) {
            int2v c = ldnt((const int2v *)(cell + i));
            double2v a = ldnt((const double2v *)(x + i)), b = ldnt((const double2v *)(ux + i)), d = ldnt((const double2v *)(uy + i)),
                     e = ldnt((const double2v *)(uz + i)), ww = ldnt((const double2v *)(w + i));
            a += 1e-9 * ww; b += 1e-9 * a; d += 1e-9 * b; e += 1e-9 * d;
            c.x += a.x > 1e30; c.y += a.y > 1e30;
            stnt((int2v *)(cell + i), c);
            stnt((double2v *)(x + i), a); stnt((double2v *)(ux + i), b); stnt((double2v *)(uy + i), d); stnt((double2v *)(uz + i), e);
        } else {
            int4v c = ldnt((const int4v *)(cell + i));
            double2v a[2], b[2], d[2], e[2], ww[2];
            for (int h = 0; h < 2; h++) {
                a[h] = ldnt((const double2v *)(x + i) + h); b[h] = ldnt((const double2v *)(ux + i) + h);
                d[h] = ldnt((const double2v *)(uy + i) + h); e[h] = ldnt((const double2v *)(uz + i) + h);
                ww[h] = ldnt((const double2v *)(w + i) + h);
            }
            for (int h = 0; h < 2; h++) { a[h] += 1e-9 * ww[h]; b[h] += 1e-9 * a[h]; d[h] += 1e-9 * b[h]; e[h] += 1e-9 * d[h]; }
            c.x += a[0].x > 1e30;
            stnt((int4v *)(cell + i), c);
            for (int h = 0; h < 2; h++) {
                stnt((double2v *)(x + i) + h, a[h]); stnt((double2v *)(ux + i) + h, b[h]);
                stnt((double2v *)(uy + i) + h, d[h]); stnt((double2v *)(uz + i) + h, e[h]);
            }
        }
    }
}

// one slot per lane; OUT: write to a second set of arrays instead of in place; NT: non-temporal hints
template <int PPT, bool OUT, bool NT>
__global__ __launch_bounds__(256) void k_stream2(const int *cell, const double *x, const double *ux, const double *uy, const double *uz,
                                                 const double *w, int *ocell, double *ox, double *oux, double *ouy, double *ouz, size_t n)
{
    const size_t chunk0 = (size_t)blockIdx.x * 256 * PPT;
    for (int p = 0; p < PPT; p++) {
        size_t i = chunk0 + (size_t)p * 256 + threadIdx.x;
        if (i >= n) return;
        int c = NT ? ldnt(cell + i) : cell[i];
        double a = NT ? ldnt(x + i) : x[i], b = NT ? ldnt(ux + i) : ux[i], d = NT ? ldnt(uy + i) : uy[i],
               e = NT ? ldnt(uz + i) : uz[i], ww = NT ? ldnt(w + i) : w[i];
        a += 1e-9 * ww; b += 1e-9 * a; d += 1e-9 * b; e += 1e-9 * d;
        c += (a > 1e30);
        if (NT) { stnt(ocell + i, c); stnt(ox + i, a); stnt(oux + i, b); stnt(ouy + i, d); stnt(ouz + i, e); }
        else { ocell[i] = c; ox[i] = a; oux[i] = b; ouy[i] = d; ouz[i] = e; }
    }
}

// AoSoA: blocks of 256 slots, each block = [cell x256 (1 KB)][x][ux][uy][uz][w] (5 x 2 KB) = 11 KB contiguous
template <int PPT, bool OUT>
__global__ __launch_bounds__(256) void k_stream_aosoa(char *buf, char *obuf, size_t n)
{
    const size_t blk0 = (size_t)blockIdx.x * PPT;
    for (int p = 0; p < PPT; p++) {
        size_t blk = blk0 + p;
        if (blk * 256 >= n) return;
        char *b = buf + blk * 11264;
        char *o = (OUT ? obuf : buf) + blk * 11264;
        int c = ldnt((const int *)b + threadIdx.x);
        const double *f = (const double *)(b + 1024);
        double a = ldnt(f + threadIdx.x), bb = ldnt(f + 256 + threadIdx.x), d = ldnt(f + 512 + threadIdx.x),
               e = ldnt(f + 768 + threadIdx.x), ww = ldnt(f + 1024 + threadIdx.x);
        a += 1e-9 * ww; bb += 1e-9 * a; d += 1e-9 * bb; e += 1e-9 * d;
        c += (a > 1e30);
        double *of = (double *)(o + 1024);
        stnt((int *)o + threadIdx.x, c);
        stnt(of + threadIdx.x, a); stnt(of + 256 + threadIdx.x, bb); stnt(of + 512 + threadIdx.x, d); stnt(of + 768 + threadIdx.x, e);
    }
}

template <class F>
int timeit(const char *name, size_t n, F launch)
{
    hipEvent_t s, e;
    CHECK(hipEventCreate(&s));
    CHECK(hipEventCreate(&e));
    for (int it = 0; it < 3; it++) launch();
    CHECK(hipDeviceSynchronize());
    const int iters = 20;
    CHECK(hipEventRecord(s));
    for (int it = 0; it < iters; it++) launch();
    CHECK(hipEventRecord(e));
    CHECK(hipEventSynchronize(e));
    float ms;
    CHECK(hipEventElapsedTime(&ms, s, e));
    ms /= iters;
    printf("%-52s %8.4f ms  %6.3f TB/s  (%.3f of 8 TB/s)\n", name, ms, 80.0 * n / ms / 1e9, 80.0 * n / ms / 1e9 / 8.0);
    return 0;
}

template <int V, int PPT>
int run(const char *name, int *cell, double **f, size_t n)
{
    hipEvent_t s, e;
    CHECK(hipEventCreate(&s));
    CHECK(hipEventCreate(&e));
    unsigned nblk = (unsigned)((n + 256 * PPT - 1) / (256 * PPT));
    for (int it = 0; it < 3; it++) k_stream<V, PPT><<<nblk, 256>>>(cell, f[0], f[1], f[2], f[3], f[4], n);
    CHECK(hipDeviceSynchronize());
    const int iters = 20;
    CHECK(hipEventRecord(s));
    for (int it = 0; it < iters; it++) k_stream<V, PPT><<<nblk, 256>>>(cell, f[0], f[1], f[2], f[3], f[4], n);
    CHECK(hipEventRecord(e));
    CHECK(hipEventSynchronize(e));
    float ms;
    CHECK(hipEventElapsedTime(&ms, s, e));
    ms /= iters;
    printf("%-44s %8.4f ms  %6.3f TB/s  (%.3f of 8 TB/s)\n", name, ms, 80.0 * n / ms / 1e9, 80.0 * n / ms / 1e9 / 8.0);
    return 0;
}

int main(int argc, char **argv)
{
    size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)100007936;
    int *cell;
    double *f[5];
    CHECK(hipMalloc(&cell, n * sizeof(int)));
    CHECK(hipMemset(cell, 0, n * sizeof(int)));
    for (int k = 0; k < 5; k++) {
        CHECK(hipMalloc(&f[k], n * sizeof(double)));
        CHECK(hipMemset(f[k], 0, n * sizeof(double)));
    }
    printf("n = %zu slots, 80 B per slot\n", n);
    if (run<1, 8>("1 slot/lane/access (4+8 B), ppt 8", cell, f, n)) return 1;
    if (run<1, 4>("1 slot/lane/access (4+8 B), ppt 4", cell, f, n)) return 1;
    if (run<2, 8>("2 slots/lane/access (8+16 B), ppt 8", cell, f, n)) return 1;
    if (run<2, 4>("2 slots/lane/access (8+16 B), ppt 4", cell, f, n)) return 1;
    if (run<4, 8>("4 slots/lane/access (16+2x16 B), ppt 8", cell, f, n)) return 1;
    if (run<4, 16>("4 slots/lane/access (16+2x16 B), ppt 16", cell, f, n)) return 1;
    // second set of arrays (what the re-binning already keeps allocated): out-of-place
    int *ocell;
    double *o[4];
    CHECK(hipMalloc(&ocell, n * sizeof(int)));
    for (int k = 0; k < 4; k++) CHECK(hipMalloc(&o[k], n * sizeof(double)));
    unsigned nb8 = (unsigned)((n + 2047) / 2048), nb4 = (unsigned)((n + 1023) / 1024), nb16 = (unsigned)((n + 4095) / 4096);
    if (timeit("SoA in place, nt, ppt 8 (k_stream2)", n, [&] { k_stream2<8, false, true><<<nb8, 256>>>(cell, f[0], f[1], f[2], f[3], f[4], cell, f[0], f[1], f[2], f[3], n); })) return 1;
    if (timeit("SoA in place, no nt, ppt 8", n, [&] { k_stream2<8, false, false><<<nb8, 256>>>(cell, f[0], f[1], f[2], f[3], f[4], cell, f[0], f[1], f[2], f[3], n); })) return 1;
    if (timeit("SoA out of place, nt, ppt 8", n, [&] { k_stream2<8, true, true><<<nb8, 256>>>(cell, f[0], f[1], f[2], f[3], f[4], ocell, o[0], o[1], o[2], o[3], n); })) return 1;
    if (timeit("SoA out of place, no nt, ppt 8", n, [&] { k_stream2<8, true, false><<<nb8, 256>>>(cell, f[0], f[1], f[2], f[3], f[4], ocell, o[0], o[1], o[2], o[3], n); })) return 1;
    if (timeit("SoA out of place, nt, ppt 4", n, [&] { k_stream2<4, true, true><<<nb4, 256>>>(cell, f[0], f[1], f[2], f[3], f[4], ocell, o[0], o[1], o[2], o[3], n); })) return 1;
    if (timeit("SoA out of place, nt, ppt 16", n, [&] { k_stream2<16, true, true><<<nb16, 256>>>(cell, f[0], f[1], f[2], f[3], f[4], ocell, o[0], o[1], o[2], o[3], n); })) return 1;
    char *a, *ao;
    size_t nblk256 = (n + 255) / 256;
    CHECK(hipMalloc(&a, nblk256 * 11264));
    CHECK(hipMalloc(&ao, nblk256 * 11264));
    CHECK(hipMemset(a, 0, nblk256 * 11264));
    if (timeit("AoSoA-256 in place, nt, 8 blocks/WG", n, [&] { k_stream_aosoa<8, false><<<(unsigned)((nblk256 + 7) / 8), 256>>>(a, ao, n); })) return 1;
    if (timeit("AoSoA-256 out of place, nt, 8 blocks/WG", n, [&] { k_stream_aosoa<8, true><<<(unsigned)((nblk256 + 7) / 8), 256>>>(a, ao, n); })) return 1;
    if (timeit("AoSoA-256 in place, nt, 4 blocks/WG", n, [&] { k_stream_aosoa<4, false><<<(unsigned)((nblk256 + 3) / 4), 256>>>(a, ao, n); })) return 1;
    return 0;
}
