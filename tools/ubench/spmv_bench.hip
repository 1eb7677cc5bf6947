// SpMV kernel-variant micro-benchmark on gfx950 (tuning aid for mk_spmv_tiles).  Builds a 7-point (or 5-point)
// Poisson CSR on the device and times variants of the CSR-stream kernel; checks results against variant 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256, TILE = 2048, ROWS = 256;
typedef double d2v __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));

__host__ __device__ inline long pre3(long r, long nx, long ny, long nz) {
    long pl = nx * ny, n = pl * nz, zc = r / pl, rem = r % pl, c = 7 * r;
    c -= (r < pl ? r : pl); c -= (r > n - pl ? r - (n - pl) : 0);
    c -= zc * nx + (rem < nx ? rem : nx); c -= zc * nx + (rem > pl - nx ? rem - (pl - nx) : 0);
    c -= (r + nx - 1) / nx; c -= r / nx; return c;
}
__global__ void gen3(long nx, long ny, long nz, int* ip, int* ix, double* dv) {
    long n = nx * ny * nz, pl = nx * ny;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r <= n; r += (long)gridDim.x * 256) {
        long p = pre3(r, nx, ny, nz); ip[r] = (int)p; if (r == n) break;
        long gx = r % nx, gy = (r / nx) % ny, gz = r / pl;
        if (gz > 0) { ix[p] = r - pl; dv[p++] = -1; } if (gy > 0) { ix[p] = r - nx; dv[p++] = -1; } if (gx > 0) { ix[p] = r - 1; dv[p++] = -1; }
        ix[p] = r; dv[p++] = 6;
        if (gx < nx - 1) { ix[p] = r + 1; dv[p++] = -1; } if (gy < ny - 1) { ix[p] = r + nx; dv[p++] = -1; } if (gz < nz - 1) { ix[p] = r + pl; dv[p++] = -1; }
    }
}

__device__ inline double block_sum(double v, double* s4) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads(); if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v; __syncthreads();
    return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

// VAR 0: current library kernel (8B/4B loads).  VAR 1: 16B value / 8B index loads (2 nnz per lane).
// VAR 2: VAR 1 + nontemporal matrix loads.  VAR 3: VAR 0 without the x gather (x = 1).  VAR 4: VAR 0, products
// summed per lane without LDS (wrong rows; timing only).  XCD: 1 = XCD-contiguous tile order.
// VAR 5/6: indices and values staged in LDS (4/8-byte loads), x gathered in the row phase.  VAR 7: VAR 0 with the
// gather folded into an 8 KB table (always an L1 hit; timing only) and VAR 8: folded into the tile's own z-plane --
// the two experiments that show the gather's cost is the instruction, not the miss.  VAR 9: VAR 5 with 16-byte
// value / 8-byte index loads.  VAR 10: 16-byte index loads, 4 nonzeros per lane (what the library kernel now does).
// VAR 11: VAR 10 + row-phase gather with the tile's own 256 x entries in LDS.
template <int VAR, int XCD>
__global__ __launch_bounds__(BLOCK) void spmv(const int* __restrict__ ip, const int* __restrict__ ix, const double* __restrict__ dv,
                                              const double* __restrict__ x, double* __restrict__ y, long nrows, long ntiles, double* part) {
    __shared__ double prod[TILE + 4];
    __shared__ double s4[4];
    const int tid = threadIdx.x;
    double acc = 0.0;
    const int G = gridDim.x, nxcd = (XCD && G % 8 == 0) ? 8 : 1, per = G / nxcd;
    const long chunk = (ntiles + nxcd - 1) / nxcd, c0 = (long)(blockIdx.x % nxcd) * chunk, cend = min(c0 + chunk, ntiles);
    for (long tile = c0 + blockIdx.x / nxcd; tile < cend; tile += per) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = ip[r0], p_hi = ip[rend];
        int my_lo = p_hi, my_hi = p_hi;
        if (r < rend) { my_lo = ip[r]; my_hi = ip[r + 1]; }
        double sum = 0.0;
        if (VAR == 5 || VAR == 6) {
            // indices and values staged in LDS by a coalesced pass; the gather of x happens in the row phase
            // (lane = row): for banded matrices consecutive lanes then read consecutive x entries.
            __shared__ int sidx[TILE];
            for (int base = p_lo; base < p_hi; base += TILE) {
                const int cnt = min(TILE, p_hi - base);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int j = k * BLOCK + tid; if (j < cnt) { sidx[j] = ix[base + j]; prod[j] = dv[base + j]; } }
                __syncthreads();
                const int lo = max(my_lo, base) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
                double a[8], xv[8]; int c[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { c[k] = (k < len) ? sidx[lo + k] : 0; a[k] = (k < len) ? prod[lo + k] : 0.0; }
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = (k < len) ? x[c[k]] : 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < len) sum += a[k] * xv[k];
                for (int k = 8; k < len; ++k) sum += prod[lo + k] * x[sidx[lo + k]];
                __syncthreads();
            }
        } else if (VAR == 11) {
            // VAR 9 (16-byte index loads) + own-block x window: x[r0 .. r0+255] sits in LDS (one coalesced load per
            // row, which CG needs anyway for <p, Ap>); in the row phase entries whose column falls into the tile's
            // own row range read LDS, only the others are gathered from memory.
            typedef int i4v __attribute__((ext_vector_type(4)));
            __shared__ int sidx[TILE + 4];
            __shared__ double sx[ROWS];
            const double xr = x[min(r, nrows - 1)];
            const int abase = p_lo & ~3;
            for (int base = abase; base < p_hi; base += TILE) {
                const int cnt = min(TILE, p_hi - base);
                i4v col[2]; d2v val[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    int j = 4 * (k * BLOCK + tid); j = (j < cnt) ? j : ((cnt - 1) & ~3);
                    col[k] = *(const i4v*)(ix + base + j);
                    val[k][0] = *(const d2v*)(dv + base + j); val[k][1] = *(const d2v*)(dv + base + j + 2);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int j = 4 * (k * BLOCK + tid);
                    *(i4v*)(sidx + j) = col[k]; *(d2v*)(prod + j) = val[k][0]; *(d2v*)(prod + j + 2) = val[k][1];
                }
                if (base == abase) sx[tid] = xr;
                __syncthreads();
                const int lo = max(my_lo, max(base, p_lo)) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
                double a[8], xv[8]; int c[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int q = (k < len) ? lo + k : 0; c[k] = sidx[q]; a[k] = prod[q]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned off = (unsigned)(c[k] - (int)r0);
                    if (k < len && off >= (unsigned)ROWS) xv[k] = x[c[k]];        // outside the block: memory
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned off = (unsigned)(c[k] - (int)r0);
                    if (off < (unsigned)ROWS) xv[k] = sx[off];                    // inside the block: LDS
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < len) sum += a[k] * xv[k];
                for (int k = 8; k < len; ++k) sum += prod[lo + k] * x[sidx[lo + k]];
                __syncthreads();
            }
        } else if (VAR == 9) {
            // wide coalesced loads (16 B values, 8 B indices, 2 nnz per lane) staged in LDS in storage order, then the
            // row phase (lane = row) walks its segment left to right and gathers x: for banded matrices consecutive
            // lanes read consecutive x entries, which the address coalescer handles at the coalesced rate.
            __shared__ int sidx[TILE + 2];
            const int abase = p_lo & ~1;
            for (int base = abase; base < p_hi; base += TILE) {
                const int cnt = min(TILE, p_hi - base);
                i2v col[4]; d2v val[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int j = 2 * (k * BLOCK + tid); j = (j < cnt) ? j : ((cnt - 1) & ~1);
                    col[k] = *(const i2v*)(ix + base + j); val[k] = *(const d2v*)(dv + base + j);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = 2 * (k * BLOCK + tid);
                    if (j < cnt) { *(i2v*)(sidx + j) = col[k]; *(d2v*)(prod + j) = val[k]; }
                }
                __syncthreads();
                const int lo = max(my_lo, max(base, p_lo)) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
                double a[8], xv[8]; int c[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int q = (k < len) ? lo + k : 0; c[k] = sidx[q]; a[k] = prod[q]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = x[(k < len) ? c[k] : 0];
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < len) sum += a[k] * xv[k];
                for (int k = 8; k < len; ++k) sum += prod[lo + k] * x[sidx[lo + k]];
                __syncthreads();
            }
        } else if (VAR == 10) {
            // VAR 1 with 16-byte index loads: 4 nonzeros per lane (1 index load, 2 value loads, 4 gathers)
            typedef int i4v __attribute__((ext_vector_type(4)));
            const int abase = p_lo & ~3;
            for (int base = abase; base < p_hi; base += TILE) {
                const int cnt = min(TILE, p_hi - base);
                i4v col[2]; d2v val[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    int j = 4 * (k * BLOCK + tid); j = (j < cnt) ? j : ((cnt - 1) & ~3);
                    col[k] = *(const i4v*)(ix + base + j);
                    val[k][0] = *(const d2v*)(dv + base + j); val[k][1] = *(const d2v*)(dv + base + j + 2);
                }
                d2v xv[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k) { xv[k][0].x = x[col[k].x]; xv[k][0].y = x[col[k].y]; xv[k][1].x = x[col[k].z]; xv[k][1].y = x[col[k].w]; }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int j = 4 * (k * BLOCK + tid);
                    if (j < cnt) {
                        d2v p0, p1; p0.x = val[k][0].x * xv[k][0].x; p0.y = val[k][0].y * xv[k][0].y; p1.x = val[k][1].x * xv[k][1].x; p1.y = val[k][1].y * xv[k][1].y;
                        *(d2v*)(prod + j) = p0; *(d2v*)(prod + j + 2) = p1;
                    }
                }
                __syncthreads();
                const int lo = max(my_lo, max(base, p_lo)) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
                double t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = (k < len) ? prod[lo + k] : 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < len) sum += t[k];
                for (int k = 8; k < len; ++k) sum += prod[lo + k];
                __syncthreads();
            }
        } else if (VAR == 1 || VAR == 2) {
            const int abase = p_lo & ~1;                       // 16-byte aligned start
            for (int base = abase; base < p_hi; base += TILE) {
                const int cnt = min(TILE, p_hi - base);        // entries [base, base+cnt); entries < p_lo are masked
                i2v col[4]; d2v val[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int j = 2 * (k * BLOCK + tid); j = (j < cnt) ? j : ((cnt - 1) & ~1);
                    if (VAR == 2) { col[k] = __builtin_nontemporal_load((const i2v*)(ix + base + j)); val[k] = __builtin_nontemporal_load((const d2v*)(dv + base + j)); }
                    else { col[k] = *(const i2v*)(ix + base + j); val[k] = *(const d2v*)(dv + base + j); }
                }
                d2v xv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { xv[k].x = x[col[k].x]; xv[k].y = x[col[k].y]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = 2 * (k * BLOCK + tid);
                    if (j < cnt) { d2v pr; pr.x = val[k].x * xv[k].x; pr.y = val[k].y * xv[k].y; *(d2v*)(prod + j) = pr; }
                }
                __syncthreads();
                const int lo = max(my_lo, max(base, p_lo)) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
                double t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = (k < len) ? prod[lo + k] : 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < len) sum += t[k];
                for (int k = 8; k < len; ++k) sum += prod[lo + k];
                __syncthreads();
            }
        } else {
            for (int base = p_lo; base < p_hi; base += TILE) {
                const int cnt = min(TILE, p_hi - base);
                int col[8]; double val[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { int j = k * BLOCK + tid; j = (j < cnt) ? j : cnt - 1; col[k] = ix[base + j]; val[k] = dv[base + j]; }
                double xv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = (VAR == 3) ? 1.0 : (VAR == 7) ? x[col[k] & 1023] : (VAR == 8) ? x[(col[k] & 0x3ffff) + (int)(r0 & ~0x3ffffL)] : x[col[k]];
                if (VAR == 4) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k * BLOCK + tid < cnt) sum += val[k] * xv[k];
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int j = k * BLOCK + tid; if (j < cnt) prod[j] = val[k] * xv[k]; }
                __syncthreads();
                const int lo = max(my_lo, base) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
                double t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = (k < len) ? prod[lo + k] : 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < len) sum += t[k];
                for (int k = 8; k < len; ++k) sum += prod[lo + k];
                __syncthreads();
            }
        }
        if (r < rend) { y[r] = sum; acc += x[r] * sum; }
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}

// VAR 6: software pipeline over tiles.  While tile t is gathered / summed, the (idx, val) stream of tile t+1 is
// already in flight into a second register set and the row pointers of tile t+2 are being fetched.
template <int XCD>
__global__ __launch_bounds__(BLOCK) void spmv_pipe(const int* __restrict__ ip, const int* __restrict__ ix, const double* __restrict__ dv,
                                                   const double* __restrict__ x, double* __restrict__ y, long nrows, long ntiles, double* part) {
    __shared__ double prod[TILE];
    __shared__ double s4[4];
    const int tid = threadIdx.x;
    double acc = 0.0;
    const int G = gridDim.x, nxcd = (XCD && G % 8 == 0) ? 8 : 1, per = G / nxcd;
    const long chunk = (ntiles + nxcd - 1) / nxcd, c0 = (long)(blockIdx.x % nxcd) * chunk, cend = min(c0 + chunk, ntiles);
    struct Meta { int p_lo, p_hi, my_lo, my_hi; };
    auto load_meta = [&](long tile, Meta& m) {
        if (tile >= cend) { m.p_lo = m.p_hi = m.my_lo = m.my_hi = 0; return; }
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = m.p_hi; m.my_hi = m.p_hi;
        if (r < rend) { m.my_lo = ip[r]; m.my_hi = ip[r + 1]; }
    };
    auto load_stream = [&](const Meta& m, int (&col)[8], double (&val)[8]) {
        const int cnt = m.p_hi - m.p_lo;
        if (cnt <= 0) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) { int j = k * BLOCK + tid; j = (j < cnt) ? j : cnt - 1; col[k] = ix[m.p_lo + j]; val[k] = dv[m.p_lo + j]; }
    };
    long tile = c0 + blockIdx.x / nxcd;
    Meta m0, m1, m2;
    int colA[8]; double valA[8]; int colB[8]; double valB[8];
    load_meta(tile, m0); load_meta(tile + per, m1);
    load_stream(m0, colA, valA);
    for (; tile < cend; tile += per) {
        load_meta(tile + 2 * per, m2);            // row pointers two tiles ahead
        load_stream(m1, colB, valB);              // matrix stream one tile ahead (stays in flight below)
        const int cnt = min(TILE, m0.p_hi - m0.p_lo);   // (ubench: tiles fit one chunk)
        double xv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = x[colA[k]];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int j = k * BLOCK + tid; if (j < cnt) prod[j] = valA[k] * xv[k]; }
        __syncthreads();
        const int lo = m0.my_lo - m0.p_lo, hi = min(m0.my_hi, m0.p_lo + cnt) - m0.p_lo, len = hi - lo;
        double t[8], sum = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (k < len) ? prod[lo + k] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < len) sum += t[k];
        for (int k = 8; k < len; ++k) sum += prod[lo + k];
        __syncthreads();
        const long r = tile * ROWS + tid;
        if (r < nrows) { y[r] = sum; acc += x[r] * sum; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { colA[k] = colB[k]; valA[k] = valB[k]; }
        m0 = m1; m1 = m2;
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}

// VAR 7: x window [r0 - H, r0 + 256 + H) staged in LDS by coalesced loads; gathers inside the window come from
// LDS, the rest from global memory.
template <int XCD, int H, int BF = 0>
__global__ __launch_bounds__(BLOCK) void spmv_win(const int* __restrict__ ip, const int* __restrict__ ix, const double* __restrict__ dv,
                                                  const double* __restrict__ x, double* __restrict__ y, long nrows, long ncols, long ntiles, double* part) {
    constexpr int WIN = ROWS + 2 * H;
    __shared__ double prod[TILE];
    __shared__ double xw[WIN];
    __shared__ double s4[4];
    const int tid = threadIdx.x;
    double acc = 0.0;
    const int G = gridDim.x, nxcd = (XCD && G % 8 == 0) ? 8 : 1, per = G / nxcd;
    const long chunk = (ntiles + nxcd - 1) / nxcd, c0 = (long)(blockIdx.x % nxcd) * chunk, cend = min(c0 + chunk, ntiles);
    for (long tile = c0 + blockIdx.x / nxcd; tile < cend; tile += per) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const long w0 = r0 - H;
        const int p_lo = ip[r0], p_hi = ip[rend];
        int my_lo = p_hi, my_hi = p_hi;
        if (r < rend) { my_lo = ip[r]; my_hi = ip[r + 1]; }
        double sum = 0.0;
        for (int base = p_lo; base < p_hi; base += TILE) {
            const int cnt = min(TILE, p_hi - base);
            int col[8]; double val[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { int j = k * BLOCK + tid; j = (j < cnt) ? j : cnt - 1; col[k] = ix[base + j]; val[k] = dv[base + j]; }
            if (base == p_lo) {
                for (int t = tid; t < WIN; t += BLOCK) { const long c = w0 + t; xw[t] = (c >= 0 && c < ncols) ? x[c] : 0.0; }
                __syncthreads();
            }
            double xv[8];
            if (BF) {
                // branch-free: every lane issues one LDS read (clamped index) and one global load (out-of-window
                // lanes their real column, in-window lanes one shared dummy address), then selects
                double xl[8], xg[8]; bool inw[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const long rel = (long)col[k] - w0; inw[k] = (rel >= 0 && rel < WIN); xl[k] = xw[inw[k] ? rel : 0]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) xg[k] = x[inw[k] ? r0 : (long)col[k]];
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = inw[k] ? xl[k] : xg[k];
            } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const long rel = (long)col[k] - w0; xv[k] = (rel >= 0 && rel < WIN) ? xw[rel] : x[col[k]]; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = k * BLOCK + tid; if (j < cnt) prod[j] = val[k] * xv[k]; }
            __syncthreads();
            const int lo = max(my_lo, base) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = (k < len) ? prod[lo + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k < len) sum += t[k];
            for (int k = 8; k < len; ++k) sum += prod[lo + k];
            __syncthreads();
        }
        if (r < rend) { y[r] = sum; acc += xw[H + tid] * sum; }
        __syncthreads();
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}

// VAR 8: (idx, val) of a tile staged in LDS by a coalesced pass whose loads were issued one tile earlier (register
// prefetch); x is gathered in the row phase (lane = row), where banded matrices give consecutive addresses.
template <int XCD>
__global__ __launch_bounds__(BLOCK) void spmv_rowgather(const int* __restrict__ ip, const int* __restrict__ ix, const double* __restrict__ dv,
                                                        const double* __restrict__ x, double* __restrict__ y, long nrows, long ntiles, double* part) {
    __shared__ double sval[TILE];
    __shared__ int sidx[TILE];
    __shared__ double s4[4];
    const int tid = threadIdx.x;
    double acc = 0.0;
    const int G = gridDim.x, nxcd = (XCD && G % 8 == 0) ? 8 : 1, per = G / nxcd;
    const long chunk = (ntiles + nxcd - 1) / nxcd, c0 = (long)(blockIdx.x % nxcd) * chunk, cend = min(c0 + chunk, ntiles);
    struct Meta { int p_lo, p_hi, my_lo, my_hi; };
    auto load_meta = [&](long tile, Meta& m) {
        if (tile >= cend) { m.p_lo = m.p_hi = m.my_lo = m.my_hi = 0; return; }
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = m.p_hi; m.my_hi = m.p_hi;
        if (r < rend) { m.my_lo = ip[r]; m.my_hi = ip[r + 1]; }
    };
    auto load_stream = [&](const Meta& m, int (&col)[8], double (&val)[8]) {
        const int cnt = m.p_hi - m.p_lo;
        if (cnt <= 0) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) { int j = k * BLOCK + tid; j = (j < cnt) ? j : cnt - 1; col[k] = ix[m.p_lo + j]; val[k] = dv[m.p_lo + j]; }
    };
    long tile = c0 + blockIdx.x / nxcd;
    Meta m0, m1, m2;
    int colA[8]; double valA[8];
    load_meta(tile, m0); load_meta(tile + per, m1);
    load_stream(m0, colA, valA);
    for (; tile < cend; tile += per) {
        const int cnt = min(TILE, m0.p_hi - m0.p_lo);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int j = k * BLOCK + tid; if (j < cnt) { sidx[j] = colA[k]; sval[j] = valA[k]; } }
        load_meta(tile + 2 * per, m2);
        load_stream(m1, colA, valA);              // next tile's stream goes in flight now
        __syncthreads();
        const int lo = m0.my_lo - m0.p_lo, hi = min(m0.my_hi, m0.p_lo + cnt) - m0.p_lo, len = hi - lo;
        double a[8], xv[8], sum = 0.0; int c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { c[k] = (k < len) ? sidx[lo + k] : 0; a[k] = (k < len) ? sval[lo + k] : 0.0; }
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = (k < len) ? x[c[k]] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < len) sum += a[k] * xv[k];
        for (int k = 8; k < len; ++k) sum += sval[lo + k] * x[sidx[lo + k]];
        const long r = tile * ROWS + tid;
        if (r < nrows) { y[r] = sum; acc += x[r] * sum; }
        __syncthreads();
        m0 = m1; m1 = m2;
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    long nx = argc > 1 ? atol(argv[1]) : 512, ny = argc > 2 ? atol(argv[2]) : 512, nz = argc > 3 ? atol(argv[3]) : 256;
    int reps = argc > 4 ? atoi(argv[4]) : 10;
    long n = nx * ny * nz, nnz = pre3(n, nx, ny, nz), ntiles = (n + ROWS - 1) / ROWS;
    int *ip, *ix; double *dv, *x, *y, *y0, *part;
    CK(hipMalloc(&ip, (n + 1) * 4)); CK(hipMalloc(&ix, (nnz + 8) * 4)); CK(hipMalloc(&dv, (nnz + 8) * 8));
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&y0, n * 8)); CK(hipMalloc(&part, 8192 * 8));
    hipLaunchKernelGGL(gen3, dim3(4096), dim3(256), 0, 0, nx, ny, nz, ip, ix, dv);
    std::vector<double> hx(n); for (long i = 0; i < n; ++i) hx[i] = 1.0 + (double)(i % 977) * 1e-3;
    CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice)); CK(hipDeviceSynchronize());
    const double bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    printf("grid %ldx%ldx%ld: n=%ld nnz=%ld, B_spmv=%.1f MB\n", nx, ny, nz, n, nnz, bytes / 1e6);
    std::vector<double> h0(n), h1(n);
    int grids[] = {1024, 2048};
#define RUN(VAR, XCD) for (int g : grids) { float ms = timeit([&] { hipLaunchKernelGGL((spmv<VAR, XCD>), dim3(g), dim3(BLOCK), 0, 0, ip, ix, dv, x, (VAR == 0 && XCD == 0) ? y0 : y, n, ntiles, part); }, reps); \
        const char* ok = "-"; if (VAR <= 2 || VAR == 5 || VAR >= 9) { CK(hipMemcpy(h0.data(), y0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), (VAR == 0 && XCD == 0) ? y0 : y, n * 8, hipMemcpyDeviceToHost)); ok = "bit-exact"; for (long i = 0; i < n; ++i) if (h0[i] != h1[i]) { ok = "MISMATCH"; break; } } \
        printf("var=%d xcd=%d grid=%4d : %9.1f us  %.2f TB/s  %s\n", VAR, XCD, g, ms * 1e3, bytes / ms / 1e9, ok); }
    RUN(0, 0) RUN(10, 0) RUN(11, 0) RUN(11, 1) RUN(3, 0)
    return 0;
#define RUNP(XCD) for (int g : {512, 1024, 2048}) { float ms = timeit([&] { hipLaunchKernelGGL((spmv_pipe<XCD>), dim3(g), dim3(BLOCK), 0, 0, ip, ix, dv, x, y, n, ntiles, part); }, reps); \
        CK(hipMemcpy(h0.data(), y0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y, n * 8, hipMemcpyDeviceToHost)); const char* ok = "bit-exact"; for (long i = 0; i < n; ++i) if (h0[i] != h1[i]) { ok = "MISMATCH"; break; } \
        printf("pipe  xcd=%d grid=%4d : %9.1f us  %.2f TB/s  %s\n", XCD, g, ms * 1e3, bytes / ms / 1e9, ok); }
    RUNP(0)
#define RUNW(XCD, H) for (int g : {1024, 2048}) { float ms = timeit([&] { hipLaunchKernelGGL((spmv_win<XCD, H>), dim3(g), dim3(BLOCK), 0, 0, ip, ix, dv, x, y, n, n, ntiles, part); }, reps); \
        CK(hipMemcpy(h0.data(), y0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y, n * 8, hipMemcpyDeviceToHost)); const char* ok = "bit-exact"; for (long i = 0; i < n; ++i) if (h0[i] != h1[i]) { ok = "MISMATCH"; break; } \
        printf("win   xcd=%d H=%4d grid=%4d : %9.1f us  %.2f TB/s  %s\n", XCD, H, g, ms * 1e3, bytes / ms / 1e9, ok); }
#define RUNR(XCD) for (int g : {1024, 1280, 2048}) { float ms = timeit([&] { hipLaunchKernelGGL((spmv_rowgather<XCD>), dim3(g), dim3(BLOCK), 0, 0, ip, ix, dv, x, y, n, ntiles, part); }, reps); \
        CK(hipMemcpy(h0.data(), y0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y, n * 8, hipMemcpyDeviceToHost)); const char* ok = "bit-exact"; for (long i = 0; i < n; ++i) if (h0[i] != h1[i]) { ok = "MISMATCH"; break; } \
        printf("rowg  xcd=%d grid=%4d : %9.1f us  %.2f TB/s  %s\n", XCD, g, ms * 1e3, bytes / ms / 1e9, ok); }
    RUNR(0)
#define RUNWB(XCD, H) for (int g : {1024, 2048}) { float ms = timeit([&] { hipLaunchKernelGGL((spmv_win<XCD, H, 1>), dim3(g), dim3(BLOCK), 0, 0, ip, ix, dv, x, y, n, n, ntiles, part); }, reps); \
        CK(hipMemcpy(h0.data(), y0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y, n * 8, hipMemcpyDeviceToHost)); const char* ok = "bit-exact"; for (long i = 0; i < n; ++i) if (h0[i] != h1[i]) { ok = "MISMATCH"; break; } \
        printf("winBF xcd=%d H=%4d grid=%4d : %9.1f us  %.2f TB/s  %s\n", XCD, H, g, ms * 1e3, bytes / ms / 1e9, ok); }
    RUNWB(0, 512) RUNWB(1, 512) RUNWB(0, 1024) RUNWB(1, 1024) RUNWB(1, 64)
    return 0;
}
