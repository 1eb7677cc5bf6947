// Micro-benchmark of streaming access patterns on gfx950 (tuning aid for mk_stream_kernel).
// y = a*x + y ; optional second output; variants of width / ownership / store policy / unroll.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)

template <int W> struct Vec;
template <> struct Vec<1> { using T = double; };
typedef double d2v __attribute__((ext_vector_type(2)));
template <> struct Vec<2> { using T = d2v; };

__device__ inline double fma_(double a, double x, double y) { return y + a * x; }
__device__ inline d2v fma_(double a, d2v x, d2v y) { d2v r; r.x = y.x + a * x.x; r.y = y.y + a * x.y; return r; }

// MODE 0: grid-stride over the whole array; MODE 1: each block owns a contiguous chunk
template <int W, int MODE, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void axpy_kernel(const double* __restrict__ x, double* __restrict__ y, double a, long n) {
    using T = typename Vec<W>::T;
    const T* xv = (const T*)x; T* yv = (T*)y;
    const long nv = n / W;
    long start, end, stride;
    if (MODE == 0) { start = (long)blockIdx.x * 256 + threadIdx.x; end = nv; stride = (long)gridDim.x * 256; }
    else { long chunk = (nv + gridDim.x - 1) / gridDim.x; chunk = (chunk + 255) / 256 * 256; start = blockIdx.x * chunk + threadIdx.x; end = min(nv, (long)(blockIdx.x + 1) * chunk); stride = 256; }
    long i = start;
    for (; i + (UNROLL - 1) * stride < end; i += UNROLL * stride) {
        T xs[UNROLL], ys[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { xs[u] = xv[i + u * stride]; ys[u] = yv[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            T r = fma_(a, xs[u], ys[u]);
            if (NT) __builtin_nontemporal_store(r, &yv[i + u * stride]); else yv[i + u * stride] = r;
        }
    }
    for (; i < end; i += stride) { T r = fma_(a, xv[i], yv[i]); yv[i] = r; }
}

// 4 reads + 2 writes (CG's x/r update shape)
template <int W, int MODE, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void xr_kernel(const double* __restrict__ p, const double* __restrict__ q, double* __restrict__ x, double* __restrict__ r, double a, long n) {
    using T = typename Vec<W>::T;
    const T* pv = (const T*)p; const T* qv = (const T*)q; T* xv = (T*)x; T* rv = (T*)r;
    const long nv = n / W;
    long start, end, stride;
    if (MODE == 0) { start = (long)blockIdx.x * 256 + threadIdx.x; end = nv; stride = (long)gridDim.x * 256; }
    else { long chunk = (nv + gridDim.x - 1) / gridDim.x; chunk = (chunk + 255) / 256 * 256; start = blockIdx.x * chunk + threadIdx.x; end = min(nv, (long)(blockIdx.x + 1) * chunk); stride = 256; }
    long i = start;
    for (; i + (UNROLL - 1) * stride < end; i += UNROLL * stride) {
        T a0[UNROLL], a1[UNROLL], a2[UNROLL], a3[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { a0[u] = pv[i + u * stride]; a1[u] = qv[i + u * stride]; a2[u] = xv[i + u * stride]; a3[u] = rv[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            T nx = fma_(a, a0[u], a2[u]), nr = fma_(a, a1[u], a3[u]);
            if (NT) { __builtin_nontemporal_store(nx, &xv[i + u * stride]); __builtin_nontemporal_store(nr, &rv[i + u * stride]); }
            else { xv[i + u * stride] = nx; rv[i + u * stride] = nr; }
        }
    }
    for (; i < end; i += stride) { xv[i] = fma_(a, pv[i], xv[i]); rv[i] = fma_(a, qv[i], rv[i]); }
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : (1L << 27);
    int reps = argc > 2 ? atoi(argv[2]) : 20;
    double *x, *y, *p, *q;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&q, n * 8));
    CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8)); CK(hipMemset(p, 0, n * 8)); CK(hipMemset(q, 0, n * 8));
    printf("n = %ld (%.1f MB per vector)\n", n, n * 8 / 1e6);
    int grids[] = {256, 512, 1024, 2048, 4096};
#define RUN_AXPY(W, MODE, UN, NT) for (int g : grids) { float ms = timeit([&] { hipLaunchKernelGGL((axpy_kernel<W, MODE, UN, NT>), dim3(g), dim3(256), 0, 0, x, y, 1e-9, n); }, reps); \
        printf("axpy W=%d mode=%d unroll=%d nt=%d grid=%4d : %8.1f us  %.2f TB/s\n", W*8, MODE, UN, NT, g, ms * 1e3, 24.0 * n / ms / 1e9); }
#define RUN_XR(W, MODE, UN, NT) for (int g : grids) { float ms = timeit([&] { hipLaunchKernelGGL((xr_kernel<W, MODE, UN, NT>), dim3(g), dim3(256), 0, 0, p, q, x, y, 1e-9, n); }, reps); \
        printf("xr   W=%d mode=%d unroll=%d nt=%d grid=%4d : %8.1f us  %.2f TB/s\n", W*8, MODE, UN, NT, g, ms * 1e3, 48.0 * n / ms / 1e9); }
    RUN_AXPY(2, 0, 1, false) RUN_AXPY(2, 0, 2, false) RUN_AXPY(2, 0, 4, false) RUN_AXPY(1, 0, 1, false) RUN_AXPY(1, 0, 4, false)
    RUN_AXPY(2, 1, 1, false) RUN_AXPY(2, 1, 4, false) RUN_AXPY(2, 0, 2, true) RUN_AXPY(1, 0, 4, true) RUN_AXPY(1, 1, 4, false)
    RUN_XR(2, 0, 1, false) RUN_XR(2, 0, 2, false) RUN_XR(1, 0, 2, false) RUN_XR(1, 0, 4, false) RUN_XR(2, 1, 2, false) RUN_XR(2, 0, 2, true) RUN_XR(1, 0, 2, true)
    return 0;
}
