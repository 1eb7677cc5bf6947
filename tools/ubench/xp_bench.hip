// xp_bench -- the shape of CG's x / p update at 512^3 (x += a p ; p = r + b p: 3 reads + 2 writes of 1.07 GB each):
// which access pattern gets closest to the copy ceiling?  (tuning aid for mk_stream_kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

template <int UN, int NTL, int NTS, int MODE>
__global__ __launch_bounds__(256) void xp(const d2v* __restrict__ r, d2v* __restrict__ x, d2v* __restrict__ p, double a, double b, long nv) {
    long start, end, stride;
    if (MODE == 0) { start = (long)blockIdx.x * 256 + threadIdx.x; end = nv; stride = (long)gridDim.x * 256; }
    else { long chunk = (nv + gridDim.x - 1) / gridDim.x; chunk = (chunk + 255) / 256 * 256; start = blockIdx.x * chunk + threadIdx.x; end = min(nv, (long)(blockIdx.x + 1) * chunk); stride = 256; }
    long i = start;
    for (; i + (UN - 1) * stride < end; i += UN * stride) {
        d2v rv[UN], xv[UN], pv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long j = i + u * stride;
            if (NTL) { rv[u] = __builtin_nontemporal_load(&r[j]); xv[u] = __builtin_nontemporal_load(&x[j]); pv[u] = __builtin_nontemporal_load(&p[j]); }
            else { rv[u] = r[j]; xv[u] = x[j]; pv[u] = p[j]; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long j = i + u * stride;
            d2v nx, np;
            nx.x = xv[u].x + a * pv[u].x; nx.y = xv[u].y + a * pv[u].y;
            np.x = rv[u].x + b * pv[u].x; np.y = rv[u].y + b * pv[u].y;
            if (NTS) { __builtin_nontemporal_store(nx, &x[j]); __builtin_nontemporal_store(np, &p[j]); }
            else { x[j] = nx; p[j] = np; }
        }
    }
    for (; i < end; i += stride) {
        d2v rv = r[i], xv = x[i], pv = p[i], nx, np;
        nx.x = xv.x + a * pv.x; nx.y = xv.y + a * pv.y; np.x = rv.x + b * pv.x; np.y = rv.y + b * pv.y;
        x[i] = nx; p[i] = np;
    }
}

// the production structure: op by value (pointers not restrict), first pair loaded before the prologue, grid-stride
struct OpXP {
    const double *r; double *p, *x; double alpha, beta;
    struct Regs { double2 rv, pv, xv; };
    __device__ void load2(long i, Regs& g) const { g.rv = *(const double2*)(r + i); g.pv = *(const double2*)(p + i); g.xv = *(const double2*)(x + i); }
    __device__ void apply2(Regs& g) const { g.xv.x = g.xv.x + alpha * g.pv.x; g.xv.y = g.xv.y + alpha * g.pv.y; g.pv.x = beta * g.pv.x - g.rv.x; g.pv.y = beta * g.pv.y - g.rv.y; }
    __device__ void store2(long i, const Regs& g) const { *(double2*)(x + i) = g.xv; *(double2*)(p + i) = g.pv; }
};
__global__ __launch_bounds__(256) void xp_prod(OpXP op, long n, const int* flag) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    OpXP::Regs first; const bool has = g < npair;
    if (has) op.load2(2 * g, first);
    if (flag[0]) return;
    if (has) { op.apply2(first); op.store2(2 * g, first); }
    for (long q = g + S; q < npair; q += S) { OpXP::Regs r; op.load2(2 * q, r); op.apply2(r); op.store2(2 * q, r); }
}

// phased (round 4): never two write streams at once.  The vectors are cut into regions of `rv` 16-byte elements; all workgroups
// sweep a region twice -- first x += a p (reads p, x; writes x), then p = b p - r (reads r, p again: from the caches; writes p) --
// so at any time the chip writes ONE array.  No barrier between the sweeps (approximate, like the column phases of format 3).
__global__ __launch_bounds__(256) void xp_phased(const d2v* __restrict__ r, d2v* __restrict__ x, d2v* __restrict__ p, double a, double b, long nv, long rv) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x;
    for (long r0 = 0; r0 < nv; r0 += rv) {
        const long r1 = min(nv, r0 + rv);
        for (long i = r0 + g; i < r1; i += S) { const d2v pv = p[i], xv = x[i]; d2v nx; nx.x = xv.x + a * pv.x; nx.y = xv.y + a * pv.y; x[i] = nx; }
        for (long i = r0 + g; i < r1; i += S) { const d2v pv = p[i], rr = r[i]; d2v np; np.x = b * pv.x - rr.x; np.y = b * pv.y - rr.y; p[i] = np; }
    }
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 134217728L;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    double *r, *x, *p;
    CK(hipMalloc(&r, n * 8)); CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&p, n * 8));
    CK(hipMemset(r, 0, n * 8)); CK(hipMemset(x, 0, n * 8)); CK(hipMemset(p, 0, n * 8));
    const long nv = n / 2;
    printf("n = %ld: 5 streams = %.2f GB per launch\n", n, 40.0 * n / 1e9);
#define RUN(UN, NTL, NTS, MODE) for (int g : {256, 512, 1024, 2048, 4096}) { \
        float ms = timeit([&] { hipLaunchKernelGGL((xp<UN, NTL, NTS, MODE>), dim3(g), dim3(256), 0, 0, (const d2v*)r, (d2v*)x, (d2v*)p, 1e-9, 0.5, nv); }, reps); \
        printf("unroll=%d ntload=%d ntstore=%d mode=%d grid=%4d : %7.1f us  %.2f TB/s\n", UN, NTL, NTS, MODE, g, ms * 1e3, 40.0 * n / ms / 1e9); }
    int* flag; CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    for (int g : {256, 512, 1024}) {
        float ms = timeit([&] { hipLaunchKernelGGL(xp_prod, dim3(g), dim3(256), 0, 0, OpXP{r, p, x, 1e-9, 0.5}, n, flag); }, reps);
        printf("production structure grid=%4d : %7.1f us  %.2f TB/s\n", g, ms * 1e3, 40.0 * n / ms / 1e9);
    }
    for (long mb : {4L, 16L, 64L, 256L, 1024L}) for (int g : {512, 1024, 2048}) {
        float ms = timeit([&] { hipLaunchKernelGGL(xp_phased, dim3(g), dim3(256), 0, 0, (const d2v*)r, (d2v*)x, (d2v*)p, 1e-9, 0.5, nv, (mb << 20) / 16); }, reps);
        printf("phased, regions of %4ld MB per vector, grid=%4d : %7.1f us  %.2f TB/s (of the 40 n bytes)\n", mb, g, ms * 1e3, 40.0 * n / ms / 1e9);
    }
    if (getenv("PHASED_ONLY")) return 0;
    RUN(1, 0, 0, 0) RUN(2, 0, 0, 0) RUN(4, 0, 0, 0)
    RUN(1, 0, 1, 0) RUN(1, 1, 1, 0) RUN(2, 1, 1, 0) RUN(1, 1, 0, 0)
    RUN(1, 0, 0, 1) RUN(2, 0, 0, 1) RUN(2, 1, 1, 1)
    return 0;
}
