// mtype_bench -- the x/p-update-shaped kernel on vectors allocated with different memory flags (default, uncached,
// fine-grained): does a memory type put the streaming kernel into the fast state deterministically?
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
template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    const long n = 134217728L, bytes = n * 8;
    // ballast first, so that the vectors are not the first allocations of the process (as in the solver)
    char *ballast; CK(hipMalloc(&ballast, 20L << 30)); CK(hipMemset(ballast, 0, 20L << 30));
    const unsigned flags[3] = {hipDeviceMallocDefault, hipDeviceMallocUncached, hipDeviceMallocFinegrained};
    const char *names[3] = {"default", "uncached", "finegrained"};
    for (int round = 0; round < 3; ++round)
        for (int f = 0; f < 3; ++f) {
            double *v[3];
            for (int i = 0; i < 3; ++i) { CK(hipExtMallocWithFlags((void **)&v[i], bytes, flags[f])); CK(hipMemset(v[i], 0, bytes)); }
            float ms = timeit([&] { hipLaunchKernelGGL(xp_prod, dim3(512), dim3(256), 0, 0, v[0], v[1], v[2], 1e-9, 0.5, n); }, 10);
            printf("round %d %-12s : %7.1f us = %.2f TB/s\n", round, names[f], ms * 1e3, 40.0 * n / ms / 1e9);
            if (round < 2) for (int i = 0; i < 3; ++i) CK(hipFree(v[i]));   // (last round: keep, so later ones get other memory)
        }
    return 0;
}
