// gridbar -- cost of a grid-wide barrier inside a persistent kernel on gfx950 (all workgroups co-resident), with
// agent-scope release / acquire so that data written before it by one XCD is visible to the others after it.
// Compared with the ~4 us a dependent kernel boundary costs in the solver loops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)

__device__ inline void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                             // release: this XCD's L2 writes become visible
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

// each round: every workgroup writes a slice, barrier, reads the slice written by a workgroup of ANOTHER XCD
template <int WORK>
__global__ __launch_bounds__(256) void persist(unsigned* counter, double* buf, long n, int rounds, double* out) {
    const int G = gridDim.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        if (WORK) {
            for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)G * 256) buf[i] = (double)(r + 1);
        }
        grid_barrier(counter, (unsigned)(G * (r + 1)));
        if (WORK) {
            const int other = (blockIdx.x + 3) % G;                  // a workgroup on another XCD
            for (long i = (long)other * 256 + threadIdx.x; i < n; i += (long)G * 256) acc += __builtin_nontemporal_load(&buf[i]) - (double)(r + 1);
            grid_barrier(counter + 32, (unsigned)(G * (r + 1)));     // (writes of the next round must wait for these reads)
        }
    }
    if (acc != 0.0) out[0] = acc;                                    // stale data would show here
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 1000;
    unsigned* counter; double *buf, *out;
    const long n = 1 << 20;                                          // 8 MB
    CK(hipMalloc(&counter, 256)); CK(hipMalloc(&buf, n * 8)); CK(hipMalloc(&out, 8));
    for (int G : {256, 512, 1024, 2048}) {
        for (int work = 0; work < 2; ++work) {
            CK(hipMemset(counter, 0, 256)); CK(hipMemset(out, 0, 8));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a));
            if (work) hipLaunchKernelGGL(persist<1>, dim3(G), dim3(256), 0, 0, counter, buf, n, rounds, out);
            else hipLaunchKernelGGL(persist<0>, dim3(G), dim3(256), 0, 0, counter, buf, n, rounds, out);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            double bad; CK(hipMemcpy(&bad, out, 8, hipMemcpyDeviceToHost));
            printf("grid=%4d %s: %.2f us per round%s\n", G, work ? "write 8 MB + barrier + read 8 MB + barrier" : "barrier only", ms * 1e3 / rounds,
                   bad != 0.0 ? "   !! stale data seen" : "");
        }
    }
    return 0;
}
