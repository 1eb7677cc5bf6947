// L1-resident load-rate micro-benchmark (gfx950): how many bytes per clock a CU's vector memory pipe returns for
// 16-byte coalesced, 8-byte coalesced and 8-byte scattered loads that all hit the L1.  Tuning aid: explains why
// the x gather of the CSR-stream SpMV costs what it costs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

// MODE 0: 16 B/lane coalesced; 1: 8 B/lane coalesced; 2: 8 B/lane scattered (LCG permutation inside the table);
// 3: 8 B/lane, stencil-like (lane l reads table[(7*l + k) % T]: 7 consecutive lanes hit neighbouring entries)
template <int MODE>
__global__ __launch_bounds__(256) void rd(const double* __restrict__ tab, int tabn, int iters, double* out) {
    const int tid = threadIdx.x;
    double acc = 0.0;
    unsigned idx = (unsigned)(tid * 2654435761u) % (unsigned)tabn;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { const int j = ((it * 8 + u) * 512 + 2 * tid) % tabn; d2v v = *(const d2v*)(tab + j); acc += v.x + v.y; }
            else if (MODE == 1) { const int j = ((it * 8 + u) * 256 + tid) % tabn; acc += tab[j]; }
            else if (MODE == 2) { idx = (idx * 1664525u + 1013904223u) % (unsigned)tabn; acc += tab[idx]; }
            else { const int j = (7 * tid + (it * 8 + u) * 97) % tabn; acc += tab[j]; }
        }
    }
    if (acc == 123.456) out[0] = acc;
}

template <int MODE>
void run(const char* name, const double* tab, int tabn, double* out, int bytes_per_lane) {
    const int iters = 2000, grid = 256 * 8;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(rd<MODE>, dim3(grid), dim3(256), 0, 0, tab, tabn, 10, out);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(rd<MODE>, dim3(grid), dim3(256), 0, 0, tab, tabn, iters, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)grid * 256 * iters * 8 * bytes_per_lane;
    printf("%-34s table %6d B : %8.1f GB/s total, %6.1f GB/s per CU (%.1f B/clk at 2.4 GHz), %.1f clk per wave instr per CU\n",
           name, tabn * 8, bytes / ms / 1e6, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4,
           2.4e9 * ms * 1e-3 / ((double)grid / 256 * 4 * iters * 8));
}

int main() {
    double *tab, *out;
    const int maxn = 1 << 20;
    CK(hipMalloc(&tab, maxn * 8)); CK(hipMalloc(&out, 64)); CK(hipMemset(tab, 0, maxn * 8));
    for (int tabn : {1024, 8192, 131072}) {
        run<0>("16 B/lane coalesced", tab, tabn, out, 16);
        run<1>("8 B/lane coalesced", tab, tabn, out, 8);
        run<2>("8 B/lane scattered", tab, tabn, out, 8);
        run<3>("8 B/lane stride-7 (CSR-like)", tab, tabn, out, 8);
    }
    return 0;
}
