// place3_bench -- CG's two fused update kernels with separately allocated vectors: time as a function of 4 KiB offsets
// of the vectors inside over-sized allocations.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void xp_prod(const double *r, double *p, double *x, double alpha, double beta, long n) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    for (long q = g; q < npair; q += S) {
        const double2 rv = *(const double2 *)(r + 2 * q), pv = *(const double2 *)(p + 2 * q), xv = *(const double2 *)(x + 2 * q);
        double2 nx, np;
        nx.x = xv.x + alpha * pv.x; nx.y = xv.y + alpha * pv.y;
        np.x = beta * pv.x - rv.x; np.y = beta * pv.y - rv.y;
        *(double2 *)(x + 2 * q) = nx; *(double2 *)(p + 2 * q) = np;
    }
}
__global__ __launch_bounds__(256) void upd_r(const double *Ap, double *r, double alpha, long n, double *part) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    double acc = 0.0;
    for (long q = g; q < npair; q += S) {
        const double2 av = *(const double2 *)(Ap + 2 * q);
        double2 rv = *(const double2 *)(r + 2 * q);
        rv.x = rv.x + alpha * av.x; rv.y = rv.y + alpha * av.y;
        acc += rv.x * rv.x; acc += rv.y * rv.y;
        *(double2 *)(r + 2 * q) = rv;
    }
    if (acc == 12345.678) part[0] = acc;
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 134217728L;
    const long bytes = n * 8, extra = 4L << 20;
    double *part; CK(hipMalloc(&part, 64));
    for (int trial = 0; trial < 3; ++trial) {
        char *v[4];
        for (int i = 0; i < 4; ++i) { CK(hipMalloc(&v[i], bytes + extra)); CK(hipMemset(v[i], 0, bytes + extra)); }
        printf("trial %d: r %p p %p x %p Ap %p\n", trial, (void *)v[0], (void *)v[1], (void *)v[2], (void *)v[3]);
        printf("  x/p update, offsets (r, x) in 4 KiB units, p fixed:\n");
        for (int orr = 0; orr < 4; ++orr) {
            printf("    r+%d:", orr);
            for (int ox = 0; ox < 8; ++ox) {
                double *r = (double *)(v[0] + orr * 4096L), *p = (double *)v[1], *x = (double *)(v[2] + ox * 4096L);
                float ms = timeit([&] { hipLaunchKernelGGL(xp_prod, dim3(512), dim3(256), 0, 0, r, p, x, 1e-9, 0.5, n); }, 5);
                printf(" x+%d:%.0f", ox, ms * 1e3);
            }
            printf("\n");
        }
        printf("  r update, offset of Ap in 4 KiB units, r fixed:");
        for (int oa = 0; oa < 8; ++oa) {
            double *r = (double *)v[0], *Ap = (double *)(v[3] + oa * 4096L);
            float ms = timeit([&] { hipLaunchKernelGGL(upd_r, dim3(512), dim3(256), 0, 0, Ap, r, 1e-9, n, part); }, 5);
            printf(" %d:%.0f", oa, ms * 1e3);
        }
        printf("\n");
    }
    return 0;
}
