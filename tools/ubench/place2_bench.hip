// place2_bench -- two concurrent WRITE streams (CG's x and p): time as a function of the offset of the second array
// inside an over-sized allocation (multiples of 256 KB and of 2 MB), for several independent allocations.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void write2k(double *a, double *b, long n) {
    const long S = (long)gridDim.x * 256, g = (long)blockIdx.x * 256 + threadIdx.x, npair = n >> 1;
    for (long q = g; q < npair; q += S) { *(double2 *)(a + 2 * q) = double2{1.0, 2.0}; *(double2 *)(b + 2 * q) = double2{3.0, 4.0}; }
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 134217728L;
    const long bytes = n * 8, extra = 80L << 20;
    for (int trial = 0; trial < 4; ++trial) {
        char *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes + extra));
        CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes + extra));
        printf("trial %d: a %p b %p\n  offset of b in units of 2 MB:", trial, (void *)a, (void *)b);
        for (int k = 0; k <= 32; ++k) {
            double *bb = (double *)(b + (long)k * (2L << 20));
            float ms = timeit([&] { hipLaunchKernelGGL(write2k, dim3(512), dim3(256), 0, 0, (double *)a, bb, n); }, 5);
            printf(" %d:%.0f", k, ms * 1e3);
        }
        printf("\n  offset of b in units of 256 KB:");
        for (int k = 0; k <= 16; ++k) {
            double *bb = (double *)(b + (long)k * (256L << 10));
            float ms = timeit([&] { hipLaunchKernelGGL(write2k, dim3(512), dim3(256), 0, 0, (double *)a, bb, n); }, 5);
            printf(" %d:%.0f", k, ms * 1e3);
        }
        printf("\n  offset of b in units of 4 KB:");
        for (int k = 0; k <= 16; ++k) {
            double *bb = (double *)(b + (long)k * 4096L);
            float ms = timeit([&] { hipLaunchKernelGGL(write2k, dim3(512), dim3(256), 0, 0, (double *)a, bb, n); }, 5);
            printf(" %d:%.0f", k, ms * 1e3);
        }
        printf("\n");
        // keep them allocated: the next trial gets other memory
    }
    return 0;
}
