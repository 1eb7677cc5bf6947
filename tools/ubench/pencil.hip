// pencil -- prices a z-marching product for stencil matrices whose farthest off-diagonal is tile aligned.
// 7-point operator on an n^3 grid, constant coefficients (6 / -1), Dirichlet.  A workgroup owns ONE tile column (256 consecutive
// rows of a line) and marches through `zc` planes: the x entries of the planes z-1, z, z+1 at the lane's own position ride in
// registers (each loaded ONCE), the near neighbours (+-1, +-line) come from loads that hit L1/L2 (the same lines this and the
// neighbouring workgroups loaded one step earlier).  Row sums left to right in column order: -P, -L, -1, 0, +1, +L, +P.
// Compared with: `flat` -- the same arithmetic, tiles in natural order, all seven from loads (what an unblocked kernel does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double rowsum(int gx, int gy, int gz, int n, double xm, double xl, double xw, double xc, double xe, double xu, double xp) {
    double s = 0.0;
    if (gz > 0) s = s + (-1.0) * xm;
    if (gy > 0) s = s + (-1.0) * xl;
    if (gx > 0) s = s + (-1.0) * xw;
    s = s + 6.0 * xc;
    if (gx < n - 1) s = s + (-1.0) * xe;
    if (gy < n - 1) s = s + (-1.0) * xu;
    if (gz < n - 1) s = s + (-1.0) * xp;
    return s;
}
template <bool NT>
__global__ __launch_bounds__(256, 8) void pencil(int n, int zc, const double* __restrict__ x, double* __restrict__ y, double* __restrict__ part) {
    const long L = n, P = (long)n * n;
    const int cols = (int)(P / 256);                         // tile columns per plane
    const int col = blockIdx.x % cols, chunk = blockIdx.x / cols;
    const int z0 = chunk * zc, z1 = min(n, z0 + zc);
    const long inplane = (long)col * 256 + threadIdx.x;
    const int gx = (int)(inplane % n), gy = (int)(inplane / n);
    long r = (long)z0 * P + inplane;
    double xm = z0 > 0 ? x[r - P] : 0.0, xc = x[r], acc = 0.0;
    for (int z = z0; z < z1; ++z, r += P) {
        const double xp = (z + 1 < n) ? x[r + P] : 0.0;
        const double xl = gy > 0 ? x[r - L] : 0.0, xu = gy < n - 1 ? x[r + L] : 0.0;
        const double xw = gx > 0 ? x[r - 1] : 0.0, xe = gx < n - 1 ? x[r + 1] : 0.0;
        const double s = rowsum(gx, gy, z, n, xm, xl, xw, xc, xe, xu, xp);
        if (NT) __builtin_nontemporal_store(s, y + r); else y[r] = s;
        acc += xc * s;
        xm = xc; xc = xp;
    }
    part[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256, 8) void flat(int n, const double* __restrict__ x, double* __restrict__ y, double* __restrict__ part) {
    const long L = n, P = (long)n * n, N = P * n;
    double acc = 0.0;
    for (long t = blockIdx.x; t < N / 256; t += gridDim.x) {
        const long r = t * 256 + threadIdx.x;
        const int gx = (int)(r % n), gy = (int)((r / n) % n), gz = (int)(r / P);
        const double s = rowsum(gx, gy, gz, n, gz > 0 ? x[r - P] : 0.0, gy > 0 ? x[r - L] : 0.0, gx > 0 ? x[r - 1] : 0.0, x[r],
                                gx < n - 1 ? x[r + 1] : 0.0, gy < n - 1 ? x[r + L] : 0.0, gz < n - 1 ? x[r + P] : 0.0);
        __builtin_nontemporal_store(s, y + r);
        acc += x[r] * s;
    }
    part[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const long N = (long)n * n * n;
    double *x, *y, *y2, *part;
    CK(hipMalloc(&x, 8 * N)); CK(hipMalloc(&y, 8 * N)); CK(hipMalloc(&y2, 8 * N)); CK(hipMalloc(&part, 8L * 65536 * 256));
    std::vector<double> hx(N);
    for (long i = 0; i < N; ++i) hx[i] = 1.0 + (double)((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy(x, hx.data(), 8 * N, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto f) { f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 20 * 1e3f; };
    const int cols = (int)((long)n * n / 256);
    float tf = timeit([&] { hipLaunchKernelGGL(flat, dim3(1792), dim3(256), 0, 0, n, x, y2, part); });
    printf("n = %d: flat (7 loads per row, natural order, grid 1792)        : %8.1f us  (%.2f TB/s of the compulsory 16 N bytes)\n", n, tf, 16.0 * N / tf / 1e6);
    for (int chunks : {1, 2, 4, 8, 16}) {
        if ((long)chunks * cols > 65536 || n % chunks) continue;
        const int zc = n / chunks;
        float t = timeit([&] { hipLaunchKernelGGL((pencil<true>), dim3(cols * chunks), dim3(256), 0, 0, n, zc, x, y, part); });
        printf("n = %d: pencil, %2d z-chunks of %3d planes (grid %5d)            : %8.1f us  (%.2f TB/s of 16 N)\n", n, chunks, zc, cols * chunks, t, 16.0 * N / t / 1e6);
    }
    std::vector<double> a(N), b(N);
    CK(hipMemcpy(a.data(), y, 8 * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y2, 8 * N, hipMemcpyDeviceToHost));
    long bad = 0; for (long i = 0; i < N; ++i) bad += a[i] != b[i];
    printf("pencil vs flat: %ld rows differ\n", bad);
    return 0;
}
