// sgather -- can the SCALAR memory path carry part of a scattered product's gathers?
// The scattered products (BASELINE config 3, the least-squares products) are bound by divergent 8-byte gathers that hit the
// L2: 2.9 clocks per gathered entry and CU through the vector L1.  A CU also has a scalar data cache
// with its own way to the L2.  This prices gathers issued as s_load_dwordx2 (one wave-level instruction per ENTRY: the lane's
// index comes down with v_readlane, the value goes back up as an SGPR operand of a vector add) against the vector gathers,
// alone and side by side, on a slice of x that fits an L2 (1.5 MiB, what a column phase of format 3 walks).
//   ./sgather [slice_doubles=196608] [entries=33554432] [reps=5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)

template <class T>
__device__ __forceinline__ T sload(const T* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

// MODE 0: vector gathers; 1: scalar gathers; 2: both -- of every 64 entries per lane-slot step, lanes gather through the
// vector path while the wave walks SHARE of the step's 64 entries of a second index block through the scalar path
template <int MODE, int SHARE>
__global__ __launch_bounds__(256) void gather(const int* __restrict__ idx, const double* __restrict__ x, long n, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * 256) >> 6;
    double vsum = 0.0, ssum = 0.0;
    for (long base = wave * 64; base + 64 <= n; base += nwaves * 64) {
        const int i = idx[base + lane];                      // (coalesced)
        if (MODE == 0 || MODE == 2) vsum += x[i];
        if (MODE == 1 || MODE == 2) {
            // the scalar share walks the indices of the NEXT wave's block in mode 2 (other entries: no reuse of the vector path's lines)
            const int j0 = MODE == 2 ? idx[(base + nwaves * 32 + lane) % (n - 64)] : i;
            constexpr int CNT = MODE == 2 ? SHARE : 64;
#pragma unroll
            for (int b = 0; b < CNT; b += 16) {
                double v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {               // sixteen scalar loads in flight, ONE wait (they return out of order)
                    const double* p = x + __builtin_amdgcn_readlane(j0, b + k);
                    asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(v[k]) : "s"(p));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    asm volatile("" : "+s"(v[k]));           // (the values are defined from here on)
                    ssum += v[k];
                }
            }
        }
    }
    const double t = vsum + ssum;
    if (t == 12345.6789) out[0] = t;
}

template <int MODE, int SHARE>
static void run(const char* name, const int* idx, const double* x, long n, double* out, int reps, int wgs_per_cu) {
    const int grid = 256 * wgs_per_cu;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<MODE, SHARE>), dim3(grid), dim3(256), 0, 0, idx, x, n, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gather<MODE, SHARE>), dim3(grid), dim3(256), 0, 0, idx, x, n, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    const double ent = (MODE == 0 ? 1.0 : (MODE == 1 ? 1.0 : 1.0 + SHARE / 64.0)) * (double)(n / 64 * 64);
    printf("%-34s %d wg/CU: %9.1f us  %7.1f G entries/s  = %5.2f clocks per entry and CU (2.4 GHz)\n", name, wgs_per_cu, us, ent / us / 1e3,
           us * 1e-6 * 2.4e9 * 256 / ent);
}

int main(int argc, char** argv) {
    const long W = argc > 1 ? atol(argv[1]) : 196608;
    const long n = argc > 2 ? atol(argv[2]) : 33554432;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    std::vector<int> h(n);
    uint64_t s = 88172645463325252ULL;
    for (long k = 0; k < n; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[k] = (int)(s % (uint64_t)W); }
    int* idx; double *x, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&x, W * 8)); CK(hipMalloc(&out, 8));
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, W * 8));
    printf("slice %ld doubles (%.2f MiB), %ld gathered entries per launch\n", W, W * 8 / 1048576.0, n);
    for (int w : {4, 8}) {
        run<0, 0>("vector gathers", idx, x, n, out, reps, w);
        run<1, 0>("scalar gathers (s_load per entry)", idx, x, n, out, reps, w);
        run<2, 16>("vector + 16/64 more by scalar", idx, x, n, out, reps, w);
        run<2, 32>("vector + 32/64 more by scalar", idx, x, n, out, reps, w);
        run<2, 64>("vector + 64/64 more by scalar", idx, x, n, out, reps, w);
    }
    return 0;
}
