// skew_bench -- does the RELATIVE placement of the vectors of a fused BLAS-1 pass matter?  CG's x / p update streams
// five arrays at the same index (3 reads + 2 writes); at 512^3 every vector is exactly 2^30 bytes, so separately
// allocated vectors tend to sit at addresses that are congruent modulo every power of two up to 2^30 -- all five
// streams then walk the HBM channels / banks in lockstep.  Here the three vectors are carved out of ONE allocation at
// base + i * (bytes + skew) and the production-shaped kernel is timed per skew.
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
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int grid = argc > 3 ? atoi(argv[3]) : 512;
    const long bytes = n * 8;
    char *slab; CK(hipMalloc(&slab, 3 * bytes + (64L << 20)));
    CK(hipMemset(slab, 0, 3 * bytes + (64L << 20)));
    double *part; CK(hipMalloc(&part, 64));
    printf("n = %ld (vector = %ld bytes), grid %d, slab base %p\n", n, bytes, grid, (void *)slab);
    const long skews[] = {0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576,
                          2097152 + 4096, 4096 + 256, 3 * 4096 + 768, 1048576 + 12288 + 256};
    for (long sk : skews) {
        double *r = (double *)slab, *x = (double *)(slab + bytes + sk), *p = (double *)(slab + 2 * (bytes + sk));
        float ms = timeit([&] { hipLaunchKernelGGL(xp_prod, dim3(grid), dim3(256), 0, 0, r, p, x, 1e-9, 0.5, n); }, reps);
        float ms2 = timeit([&] { hipLaunchKernelGGL(upd_r, dim3(grid), dim3(256), 0, 0, x, r, 1e-9, n, part); }, reps);
        printf("skew %9ld B : x/p update %7.1f us = %.2f TB/s   r update %7.1f us = %.2f TB/s\n", sk, ms * 1e3, 40.0 * n / ms / 1e9,
               ms2 * 1e3, 24.0 * n / ms2 / 1e9);
    }
    return 0;
}
