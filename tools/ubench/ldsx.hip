// ldsx -- price of "x through LDS": ONE workgroup per CU pulls a whole vector (8 MB) through a ring of LDS slots with
// global_load_lds while 125 KB of the CU's LDS are taken by row accumulators (the one-launch least-squares product of
// VERDICT r4 item 4 in its LDS form: 4e6 rows = 15 625 per CU with their sums resident, every entry's x read from LDS
// instead of gathered through the L1 at 2.9 clocks per entry).  What the ring can deliver is Little's law: bytes in
// flight / latency; the accumulators leave 32 KB for it.
//   ./ldsx [n_doubles=1048576] [reps=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// THREADS per workgroup, SLOT doubles per ring slot, DEPTH slots, CONS: 0 = stream only, 1 = every lane does one
// x read + accumulator read-modify-write per slot (an upper bound of the real walk: ~160 entries per slot and CU)
template <int THREADS, int SLOT, int DEPTH, int CONS>
__global__ __launch_bounds__(THREADS) void ldsx(const double* __restrict__ x, long n, int acc_doubles, double* __restrict__ out) {
    extern __shared__ double lds[];
    double* acc = lds;
    double* ring = lds + acc_doubles;
    constexpr int PER = SLOT * 8 / (THREADS * 16);          // 16-byte DMA instructions per thread and slot
    static_assert(PER >= 1 && PER * THREADS * 16 == SLOT * 8, "slot must be a whole number of wave-level copies");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long nslots = n / SLOT;
    for (int i = tid; i < acc_doubles; i += THREADS) acc[i] = 0.0;
    auto issue = [&](long s) {
        const long sc = s < nslots ? s : nslots - 1;        // (unconditional loads: clamp)
        double* dst = ring + (s % DEPTH) * SLOT;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int c0 = (j * (THREADS / 64) + wv) * 128;  // 128 doubles per wave-level copy
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + sc * SLOT + c0 + 2 * lane),
                                             (__attribute__((address_space(3))) void*)(dst + c0), 16, 0, 0);
        }
    };
    for (int s = 0; s < DEPTH - 1; ++s) issue(s);
    double sum = 0.0;
    unsigned h = tid * 2654435761u;
    for (long s = 0; s < nslots; ++s) {
        wait_vm<(DEPTH - 2) * PER>();                        // slot s has landed (this thread's share)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // ... everybody's share; and slot s-1 is no longer read (no fence:
                                                             // __syncthreads() would wait for every DMA in flight)
        issue(s + DEPTH - 1);                                // into the buffer of slot s-1
        const double* cur = ring + (s % DEPTH) * SLOT;
        if (CONS) {
            h = h * 1664525u + 1013904223u;
            const int c = (h >> 8) & (SLOT - 1);
            const int r = (wv * (acc_doubles / (THREADS / 64))) + ((h >> 20) % (acc_doubles / (THREADS / 64)));
            const double t = acc[r] + 0.5 * cur[c];
            acc[r] = t;
        } else {
            sum += cur[tid & (SLOT - 1)];
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (CONS) sum = acc[tid];
    if (sum == 12345.678) out[0] = sum;
}

template <int THREADS, int SLOT, int DEPTH, int CONS>
static void run(const double* x, long n, double* out, int reps, int acc_kb) {
    const int acc_doubles = acc_kb * 128;
    const size_t lds = (size_t)acc_doubles * 8 + (size_t)SLOT * DEPTH * 8;
    if (lds > 160 * 1024) { printf("threads %4d slot %5d x %d  acc %3d KB: %zu B of LDS -- does not fit\n", THREADS, SLOT, DEPTH, acc_kb, lds); return; }
    auto k = ldsx<THREADS, SLOT, DEPTH, CONS>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int grid : {256, 512}) {
        if (grid * lds > 256 * 160 * 1024) continue;
        hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, x, n, acc_doubles, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, x, n, acc_doubles, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double us = ms * 1e3 / reps;
        printf("threads %4d slot %5d x %d  acc %3d KB  cons %d  grid %3d: %8.1f us per sweep of %.1f MB  = %6.1f GB/s per workgroup, %5.0f barriers\n",
               THREADS, SLOT, DEPTH, acc_kb, CONS, grid, us, n * 8 / 1e6, n * 8 / us / 1e3, (double)(n / SLOT));
    }
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 1 << 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    double *x, *out;
    CK(hipMalloc(&x, n * 8 + 4096)); CK(hipMalloc(&out, 8));
    CK(hipMemset(x, 0, n * 8 + 4096));
    // the budget the accumulators leave (125 KB + 32 KB), then what more ring would buy with fewer resident rows
    run<1024, 2048, 2, 0>(x, n, out, reps, 125);
    run<512, 1024, 4, 0>(x, n, out, reps, 125);
    run<256, 1024, 4, 0>(x, n, out, reps, 125);
    run<256, 512, 8, 0>(x, n, out, reps, 125);
    run<512, 1024, 4, 1>(x, n, out, reps, 125);
    run<1024, 2048, 2, 1>(x, n, out, reps, 125);
    run<256, 1024, 4, 1>(x, n, out, reps, 125);
    run<1024, 2048, 4, 0>(x, n, out, reps, 62);
    run<1024, 2048, 6, 0>(x, n, out, reps, 62);
    run<512, 2048, 6, 0>(x, n, out, reps, 62);
    run<1024, 4096, 4, 0>(x, n, out, reps, 16);
    run<1024, 4096, 8, 0>(x, n, out, reps, 16);
    run<1024, 2048, 16, 0>(x, n, out, reps, 16);
    run<256, 2048, 16, 0>(x, n, out, reps, 16);
    return 0;
}
