// place_bench -- is the speed of a streaming kernel a property of WHERE its buffers were placed?  Allocates K slabs
// (all kept alive), times the x/p-update-shaped kernel on each, then frees every other slab, allocates again and
// re-times: the same code on the same chip in the same process, only the placement differs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
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
__global__ __launch_bounds__(256) void copyk(const double *a, double *b, long n) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    for (long q = g; q < npair; q += S) *(double2 *)(b + 2 * q) = *(const double2 *)(a + 2 * q);
}
__global__ __launch_bounds__(256) void readk(const double *a, long n, double *out) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    double acc = 0.0;
    for (long q = g; q < npair; q += S) { const double2 v = *(const double2 *)(a + 2 * q); acc += v.x; acc += v.y; }
    if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256) void read3k(const double *a, const double *b, const double *c, long n, double *out) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    double acc = 0.0;
    for (long q = g; q < npair; q += S) {
        const double2 v = *(const double2 *)(a + 2 * q), w = *(const double2 *)(b + 2 * q), u = *(const double2 *)(c + 2 * q);
        acc += v.x; acc += v.y; acc += w.x * u.x; acc += w.y * u.y; }
    if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256) void write2k(double *a, double *b, long n) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    for (long q = g; q < npair; q += S) { *(double2 *)(a + 2 * q) = double2{1.0, 2.0}; *(double2 *)(b + 2 * q) = double2{3.0, 4.0}; }
}
__global__ __launch_bounds__(256) void write1k(double *a, long n) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    for (long q = g; q < npair; q += S) *(double2 *)(a + 2 * q) = double2{1.0, 2.0};
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char **argv) {
    const long n = 134217728L, bytes = n * 8;
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<char *> slabs(K);
    double *scratch; CK(hipMalloc(&scratch, 64));
    auto probe = [&](int k) {
        double *r = (double *)slabs[k], *x = (double *)(slabs[k] + bytes), *p = (double *)(slabs[k] + 2 * bytes);
        float ms = timeit([&] { hipLaunchKernelGGL(xp_prod, dim3(512), dim3(256), 0, 0, r, p, x, 1e-9, 0.5, n); }, 10);
        float ms2 = timeit([&] { hipLaunchKernelGGL(copyk, dim3(2048), dim3(256), 0, 0, r, x, n); }, 10);
        float t[3];
        for (int i = 0; i < 3; ++i) { double *a = (double *)(slabs[k] + i * bytes); t[i] = timeit([&] { hipLaunchKernelGGL(readk, dim3(1024), dim3(256), 0, 0, a, n, scratch); }, 10); }
        float t3 = timeit([&] { hipLaunchKernelGGL(read3k, dim3(512), dim3(256), 0, 0, r, p, x, n, scratch); }, 10);
        float tw = timeit([&] { hipLaunchKernelGGL(write2k, dim3(512), dim3(256), 0, 0, p, x, n); }, 10);
        float w1[3], w2[3];
        for (int i = 0; i < 3; ++i) { double *a = (double *)(slabs[k] + i * bytes); w1[i] = timeit([&] { hipLaunchKernelGGL(write1k, dim3(512), dim3(256), 0, 0, a, n); }, 10); }
        for (int i = 0; i < 3; ++i) { double *a = (double *)(slabs[k] + i * bytes), *b = (double *)(slabs[k] + ((i + 1) % 3) * bytes);
            w2[i] = timeit([&] { hipLaunchKernelGGL(write2k, dim3(512), dim3(256), 0, 0, a, b, n); }, 10); }
        printf("  write1 %5.1f %5.1f %5.1f us (%.2f %.2f %.2f TB/s) ; write2 pairs 01 %5.1f 12 %5.1f 20 %5.1f us\n", w1[0] * 1e3, w1[1] * 1e3, w1[2] * 1e3,
               8.0 * n / w1[0] / 1e9, 8.0 * n / w1[1] / 1e9, 8.0 * n / w1[2] / 1e9, w2[0] * 1e3, w2[1] * 1e3, w2[2] * 1e3);
        printf("  slab %2d at %p : x/p update %7.1f us = %.2f TB/s ; copy %6.1f ; read each %5.1f %5.1f %5.1f ; read3 %6.1f (%.2f TB/s) ; write2 %6.1f (%.2f TB/s)\n", k, (void *)slabs[k], ms * 1e3,
               40.0 * n / ms / 1e9, ms2 * 1e3, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t3 * 1e3, 24.0 * n / t3 / 1e9, tw * 1e3, 16.0 * n / tw / 1e9);
    };
    for (int k = 0; k < K; ++k) { CK(hipMalloc(&slabs[k], 3 * bytes)); CK(hipMemset(slabs[k], 0, 3 * bytes)); }
    printf("first placement\n");
    for (int k = 0; k < K; ++k) probe(k);
    for (int k = 0; k < K; k += 2) CK(hipFree(slabs[k]));
    for (int k = 0; k < K; k += 2) { CK(hipMalloc(&slabs[k], 3 * bytes)); CK(hipMemset(slabs[k], 0, 3 * bytes)); }
    printf("after freeing and re-allocating the even slabs\n");
    for (int k = 0; k < K; ++k) probe(k);
    // three separately allocated vectors, as the solver does
    double *v[3];
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&v[i], bytes + 16)); CK(hipMemset(v[i], 0, bytes)); }
    float ms = timeit([&] { hipLaunchKernelGGL(xp_prod, dim3(512), dim3(256), 0, 0, v[0], v[1], v[2], 1e-9, 0.5, n); }, 10);
    printf("three separate allocations %p %p %p : %7.1f us = %.2f TB/s\n", (void *)v[0], (void *)v[1], (void *)v[2], ms * 1e3, 40.0 * n / ms / 1e9);
    return 0;
}
