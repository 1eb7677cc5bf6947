// pencil2 -- a FAIR ceiling for the z-marching product of the constant-coefficient 512^3 operator.
// Round 4's pencil.hip loaded plane z+1 at the top of step z and used it in the same step: 512 dependent HBM round trips per
// workgroup (0.59 ms = 512 x 1.15 us) -- a latency chain, not a bandwidth measurement; and neighbouring tile columns sat on
// different XCDs (blockIdx % 8), so every +-line neighbour crossed the fabric again.  Here:
//   * XCD-aware brick map: XCD k owns the lines [n/8 k, n/8 (k+1)) of every plane, so in-plane neighbours share an L2;
//   * two adjacent cells per lane: 16-byte loads of x, 16-byte non-temporal stores of y;
//   * planes z+1 .. z+D are in flight in registers (D-deep software prefetch, consumed in issue order);
//   * `line`  variant: a workgroup owns one 512-cell line; +-line and +-1 neighbours by loads (L1 / L2 hits), also D deep;
//   * `brick` variant: a workgroup owns a 128 x 4 brick; the current plane goes through a double-buffered LDS tile with a
//     one-cell halo (halo cells loaded D deep, one per lane), +-1 and +-line neighbours are LDS reads, one barrier per plane.
// Arithmetic: the CSR row sum left to right in column order (-P, -L, -1, 0, +1, +L, +P), -ffp-contract=off, fused <x, Ax>.
// Checked against the flat kernel row by row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

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
// the same sum with its first term (the -plane neighbour's) formed by the caller: s0 = (gz > 0) ? 0.0 + (-1.0) * xm : 0.0
__device__ __forceinline__ double rowsum_from(double s, int gx, int gy, int gz, int n, double xl, double xw, double xc, double xe, double xu, double xp) {
    if (gy > 0) s = s + (-1.0) * xl;
    if (gx > 0) s = s + (-1.0) * xw;
    s = s + 6.0 * xc;
    if (gx < n - 1) s = s + (-1.0) * xe;
    if (gy < n - 1) s = s + (-1.0) * xu;
    if (gz < n - 1) s = s + (-1.0) * xp;
    return s;
}
__device__ __forceinline__ d2 ld2(const double *p) { return *reinterpret_cast<const d2 *>(p); }
__device__ __forceinline__ void st2nt(double *p, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p)); }

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

// Software pipeline WITHOUT register copies: planes live in a ring of R register slots and the loop is unrolled R times, so
// every slot index is static.  (A rotating xm <- xc <- xp with a prefetch queue makes the compiler copy queue registers at the
// end of the unrolled body; a copy of a register with a load in flight needs that load, and the loads return in order: the
// loop drained its pipeline once per round.)  At step zz the slots hold the planes zz-1 .. zz+R-2: prefetch depth R - 2.
// Every load is UNCONDITIONAL (addresses clamped into the grid; rowsum ignores what a missing neighbour's load returns):
// with loads under branches the compiler cannot count the loads in flight and waits for all of them.

// ---- line variant: one 512-cell line per workgroup (n == 512), everything in register rings -------------------------------
template <int R, bool XCD, int OCC>
__global__ __launch_bounds__(256, OCC) void line_k(int n, int zc, const double* __restrict__ x, double* __restrict__ y, double* __restrict__ part) {
    const long L = n, P = (long)n * n;
    const int cols = (int)(P / 512);
    int col, chunk;
    if (XCD) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, cpx = cols / 8; col = xcd * cpx + slot % cpx; chunk = slot / cpx; }
    else { col = blockIdx.x % cols; chunk = blockIdx.x / cols; }
    const int z0 = chunk * zc, z1 = min(n, z0 + zc);
    const long c = (long)col * 512 + 2 * threadIdx.x;
    const int gx = (int)(c % n), gy = (int)(c / n);
    const double *xb = x + c;
    d2 ring[R], nl[R], nu[R];
    double nw[R], ne[R];
    const long oL = gy > 0 ? -L : 0, oU = gy < n - 1 ? L : 0, oW = gx > 0 ? -1 : 0, oE = gx + 2 < n ? 2 : 0;
    auto plane = [&](int p) { return ld2(xb + (long)min(max(p, 0), n - 1) * P); };
    auto neigh = [&](int p, int d) {
        const double *xp = xb + (long)min(p, n - 1) * P;
        nl[d] = ld2(xp + oL);
        nu[d] = ld2(xp + oU);
        nw[d] = xp[oW];
        ne[d] = xp[oE];
    };
#pragma unroll
    for (int d = 0; d < R; ++d) {
        ring[d] = plane(z0 - 1 + d);
        neigh(z0 + d, d);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): a clean entry edge for the loop header's wait count
    double acc = 0.0;
    for (int z = z0; z < z1; z += R) {
#pragma unroll
        for (int d = 0; d < R; ++d) {
            const int zz = z + d;
            const d2 xm = ring[d], xc = ring[(d + 1) % R], xp = ring[(d + 2) % R], l = nl[d], u = nu[d];
            const double w = nw[d], e = ne[d];
            d2 s;
            s.x = rowsum(gx, gy, zz, n, xm.x, l.x, w, xc.x, xc.y, u.x, xp.x);
            s.y = rowsum(gx + 1, gy, zz, n, xm.y, l.y, xc.x, xc.y, e, u.y, xp.y);
            ring[d] = plane(zz + R - 1);
            neigh(zz + R, d);
            st2nt(y + (long)zz * P + c, s);
            acc += xc.x * s.x;
            acc += xc.y * s.y;
        }
    }
    part[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}

// ---- brick variant: 128 x 4 cells per workgroup, in-plane neighbours through LDS --------------------------------------------
template <int R, bool XCD, int OCC>
__global__ __launch_bounds__(256, OCC) void brick_k(int n, int zc, const double* __restrict__ x, double* __restrict__ y, double* __restrict__ part) {
    constexpr int RS = 132;                                   // LDS row: [0] pad, [1] west halo, [2..129] cells, [130] east halo, [131] pad
    __shared__ __attribute__((aligned(16))) double tile[2][6][RS];
    const long L = n, P = (long)n * n;
    const int bpl = n / 128, bpp = bpl * (n / 4);            // bricks per brick row, per plane
    int bi, chunk;
    if (XCD) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = bpp / 8; bi = xcd * bpx + slot % bpx; chunk = slot / bpx; }
    else { bi = blockIdx.x % bpp; chunk = blockIdx.x / bpp; }
    const int x0 = (bi % bpl) * 128, y0 = (bi / bpl) * 4;
    const int z0 = chunk * zc, z1 = min(n, z0 + zc);
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int gx = x0 + 2 * l, gy = y0 + w;
    const long c = (long)gy * L + gx;
    const double *xb = x + c;
    // halo cell of this lane: lanes 0..127 the line below the brick, 128..255 the line above; lanes 0..7 also a west / east cell
    const int hy = min(max(tid < 128 ? y0 - 1 : y0 + 4, 0), n - 1), hx = x0 + (tid & 127);
    const double *hb = x + (long)hy * L + hx;
    const int ey = y0 + (tid & 3), ex = min(max((tid & 4) ? x0 + 128 : x0 - 1, 0), n - 1);
    const double *eb = x + (long)ey * L + ex;
    double *hdst = &tile[0][tid < 128 ? 0 : 5][2 + (tid & 127)];
    __shared__ double dump[2 * 6 * RS + 256];                 // lanes without a west / east cell write here: no divergent branch in the loop
    double *edst = tid < 8 ? &tile[0][1 + (tid & 3)][(tid & 4) ? 130 : 1] : dump + tid;
    double *cdst = &tile[0][1 + w][2 + 2 * l];
    d2 ring[R];
    double h[R], e[R];
    auto plane = [&](int p) { return ld2(xb + (long)min(max(p, 0), n - 1) * P); };
    auto halo = [&](int p, int d) {
        p = min(p, n - 1);
        h[d] = hb[(long)p * P];
        e[d] = eb[(long)p * P];
    };
#pragma unroll
    for (int d = 0; d < R; ++d) {
        ring[d] = plane(z0 - 1 + d);
        halo(z0 + d, d);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
    double acc = 0.0;
    int z = z0;
    do {
#pragma unroll
        for (int d = 0; d < R; ++d) {
            const int zz = z + d;
            const int bo = (d & 1) * 6 * RS;                  // (R is even: the buffer alternates across rounds too)
            // (the -plane term first: its slot is dead before the slot's next load is issued -- no register copies in the loop)
            const d2 xm = ring[d], xc = ring[(d + 1) % R], xp = ring[(d + 2) % R];
            d2 s;
            s.x = zz > 0 ? 0.0 + (-1.0) * xm.x : 0.0;
            s.y = zz > 0 ? 0.0 + (-1.0) * xm.y : 0.0;
            *reinterpret_cast<d2 *>(cdst + bo) = xc;
            hdst[bo] = h[d];
            edst[bo] = e[d];
            __builtin_amdgcn_sched_barrier(0);                // (the slots' last uses stay ABOVE their reloads)
            ring[d] = plane(zz + R - 1);
            halo(zz + R, d);
            __syncthreads();
            const double *row = &tile[0][1 + w][2 + 2 * l] + bo;
            const d2 lo = *reinterpret_cast<const d2 *>(row - RS), up = *reinterpret_cast<const d2 *>(row + RS);
            const double we = row[-1], ea = row[2];
            s.x = rowsum_from(s.x, gx, gy, zz, n, lo.x, we, xc.x, xc.y, up.x, xp.x);
            s.y = rowsum_from(s.y, gx + 1, gy, zz, n, lo.y, xc.x, xc.y, ea, up.y, xp.y);
            st2nt(y + (long)zz * P + c, s);
            acc += xc.x * s.x;
            acc += xc.y * s.y;
        }
        z += R;
    } while (z < z1);
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
    std::vector<double> a(N), b(N);
    float tf = timeit([&] { hipLaunchKernelGGL(flat, dim3(1792), dim3(256), 0, 0, n, x, y2, part); });
    printf("n = %d: flat (7 loads per row, natural order, grid 1792): %8.1f us  (%.2f TB/s of the compulsory 16 N bytes)\n", n, tf, 16.0 * N / tf / 1e6);
    CK(hipMemcpy(b.data(), y2, 8 * N, hipMemcpyDeviceToHost));
    auto check = [&](const char *what) {
        CK(hipMemcpy(a.data(), y, 8 * N, hipMemcpyDeviceToHost));
        long bad = 0; for (long i = 0; i < N; ++i) bad += a[i] != b[i];
        if (bad) printf("   !! %s: %ld rows differ from flat\n", what, bad);
        CK(hipMemset(y, 0, 8 * N));
    };
    const int cols = (int)((long)n * n / 512);
#define RUN(KERN, NAME, D, XCD, OCC, CH) do { \
        if (n % (CH) == 0 && (n / (CH)) % (D) == 0 && (long)(CH) * cols <= 65536) { \
            const int zc = n / (CH); \
            float t = timeit([&] { hipLaunchKernelGGL((KERN<D, XCD, OCC>), dim3(cols * (CH)), dim3(256), 0, 0, n, zc, x, y, part); }); \
            printf("n = %d: %-5s R=%d xcd=%d occ=%d chunks=%2d (grid %5d): %8.1f us  (%.2f TB/s of 16 N)\n", n, NAME, D, (int)XCD, OCC, CH, cols * (CH), t, 16.0 * N / t / 1e6); \
            check(NAME); } } while (0)
#define SWEEP(KERN, NAME) do { \
        RUN(KERN, NAME, 2, true, 4, 4); \
        RUN(KERN, NAME, 4, false, 4, 4); \
        RUN(KERN, NAME, 4, true, 4, 1); RUN(KERN, NAME, 4, true, 4, 2); RUN(KERN, NAME, 4, true, 4, 4); RUN(KERN, NAME, 4, true, 4, 8); RUN(KERN, NAME, 4, true, 4, 16); \
        RUN(KERN, NAME, 4, true, 8, 4); RUN(KERN, NAME, 4, true, 8, 8); RUN(KERN, NAME, 4, true, 8, 16); \
        RUN(KERN, NAME, 8, true, 2, 1); RUN(KERN, NAME, 8, true, 2, 2); RUN(KERN, NAME, 8, true, 2, 4); \
        RUN(KERN, NAME, 8, true, 4, 2); RUN(KERN, NAME, 8, true, 4, 4); RUN(KERN, NAME, 8, true, 4, 8); RUN(KERN, NAME, 8, false, 4, 4); \
        RUN(KERN, NAME, 16, true, 2, 1); RUN(KERN, NAME, 16, true, 2, 2); RUN(KERN, NAME, 16, true, 2, 4); \
    } while (0)
    if (n == 512) SWEEP(line_k, "line");
    SWEEP(brick_k, "brick");
    return 0;
}
