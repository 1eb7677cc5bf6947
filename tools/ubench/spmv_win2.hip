// spmv_win2 -- micro-benchmark of the "windowed tile" SpMV (tuning aid for mk_spmv_tiles, gfx950).
//
// Per 256-row tile the column set is covered at build time by <= NW contiguous windows of x; the kernel stages the
// windows in LDS with coalesced 16-byte loads and multiplies against ds_read_b64, so the product phase issues NO
// gather through the texture-address path.  The per-nonzero LDS slot is a uint16 (2 B instead of the 4 B column).
// Variants timed here:   base  -- the round-1 library kernel (16-byte index loads + 4 gathers per lane)
//                        win   -- windows in LDS, uint16 slots, raw fp64 values
//                        winvc -- win + per-tile value dictionary (uint8 codes)
// All variants must be bit-identical to `base` (row sums are formed left to right in every one).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256, TILE = 2048, ROWS = 256, NW = 8, CAP = 4096;
typedef double d2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

struct WDesc { int nwin, L; int delta[NW]; int off[NW]; };

__host__ __device__ inline long pre3(long r, long nx, long ny, long nz) {
    long pl = nx * ny, n = pl * nz, zc = r / pl, rem = r % pl, c = 7 * r;
    c -= (r < pl ? r : pl); c -= (r > n - pl ? r - (n - pl) : 0);
    c -= zc * nx + (rem < nx ? rem : nx); c -= zc * nx + (rem > pl - nx ? rem - (pl - nx) : 0);
    c -= (r + nx - 1) / nx; c -= r / nx; return c;
}
__global__ void gen3(long nx, long ny, long nz, int* ip, int* ix, double* dv, uint8_t* vc) {
    long n = nx * ny * nz, pl = nx * ny;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r <= n; r += (long)gridDim.x * 256) {
        long p = pre3(r, nx, ny, nz); ip[r] = (int)p; if (r == n) break;
        long gx = r % nx, gy = (r / nx) % ny, gz = r / pl;
        if (gz > 0) { ix[p] = r - pl; vc[p] = 0; dv[p++] = -1; } if (gy > 0) { ix[p] = r - nx; vc[p] = 0; dv[p++] = -1; } if (gx > 0) { ix[p] = r - 1; vc[p] = 0; dv[p++] = -1; }
        ix[p] = r; vc[p] = 1; dv[p++] = 6;
        if (gx < nx - 1) { ix[p] = r + 1; vc[p] = 0; dv[p++] = -1; } if (gy < ny - 1) { ix[p] = r + nx; vc[p] = 0; dv[p++] = -1; } if (gz < nz - 1) { ix[p] = r + pl; vc[p] = 0; dv[p++] = -1; }
    }
}

__device__ inline double block_sum(double v, double* s4) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads(); if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v; __syncthreads();
    return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

// ------------------------------------------------------------------------------------------------- cover builder
__device__ inline void bitonic_sort(int* a, int n2) {
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += BLOCK) {
                const int q = i ^ j;
                if (q > i) {
                    const int u = a[i], v = a[q];
                    const bool up = ((i & k) == 0);
                    if ((u > v) == up) { a[i] = v; a[q] = u; }
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(BLOCK) void cover_kernel(const int* __restrict__ ip, const int* __restrict__ ix, long nrows, long ntiles,
                                                      int wmax, WDesc* __restrict__ wd, uint16_t* __restrict__ sl, int* __restrict__ stats) {
    __shared__ int key[CAP];
    __shared__ int heads[NW + 1];
    __shared__ int wst[NW], wof[NW];
    __shared__ int nh, s_ok, s_nw;
    const int tid = threadIdx.x;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows);
        const int p_lo = ip[r0], p_hi = ip[rend], cnt = p_hi - p_lo;
        bool ok = cnt > 0 && cnt <= CAP;
        if (ok) {
            int n2 = 2; while (n2 < cnt) n2 <<= 1;
            for (int i = tid; i < n2; i += BLOCK) key[i] = (i < cnt) ? (ix[p_lo + i] >> 1) : 0x7fffffff;
            __syncthreads();
            bitonic_sort(key, n2);
            bool found = false;
            for (int G = 8; G <= 4096 && !found; G <<= 3) {
                if (tid == 0) nh = 0;
                __syncthreads();
                for (int i = tid; i < cnt; i += BLOCK)
                    if (i == 0 || key[i] - key[i - 1] > G) { const int q = atomicAdd(&nh, 1); if (q < NW) heads[q] = i; }
                __syncthreads();
                found = (nh <= NW);
                __syncthreads();
            }
            if (tid == 0) {
                int okk = found ? 1 : 0, nw = found ? nh : 0, L = 0;
                if (found) {
                    for (int a = 1; a < nw; ++a) { const int v = heads[a]; int b = a - 1; while (b >= 0 && heads[b] > v) { heads[b + 1] = heads[b]; --b; } heads[b + 1] = v; }
                    heads[nw] = cnt;
                    for (int k = 0; k < nw; ++k) {
                        const int st = key[heads[k]] * 2, en = (key[heads[k + 1] - 1] + 1) * 2;
                        wst[k] = st; wof[k] = L; L += en - st;
                    }
                    if (L > wmax) okk = 0;
                }
                s_ok = okk; s_nw = nw;
                WDesc d; d.nwin = okk ? nw : 0; d.L = okk ? L : 0;
                for (int k = 0; k < NW; ++k) { d.delta[k] = (okk && k < nw) ? wst[k] - wof[k] : 0; d.off[k] = (okk && k < nw) ? wof[k] : 0x7fffffff; }
                wd[tile] = d;
                if (okk) { atomicMax(&stats[0], L); atomicAdd(&stats[1], 1); }
            }
            __syncthreads();
            ok = s_ok != 0;
            if (ok) {
                const int nw = s_nw;
                for (int j = tid; j < cnt; j += BLOCK) {
                    const int col = ix[p_lo + j];
                    int k = 0;
                    for (int q = 1; q < nw; ++q) k += (wst[q] <= col) ? 1 : 0;
                    sl[p_lo + j] = (uint16_t)(wof[k] + col - wst[k]);
                }
            }
            __syncthreads();
        } else if (tid == 0) {
            WDesc d; d.nwin = 0; d.L = 0;
            for (int k = 0; k < NW; ++k) { d.delta[k] = 0; d.off[k] = 0x7fffffff; }
            wd[tile] = d;
        }
    }
}

// ------------------------------------------------------------------------------------------------- kernels
__global__ __launch_bounds__(BLOCK) void spmv_base(const int* __restrict__ ip, const int* __restrict__ ix, const double* __restrict__ dv,
                                                   const double* __restrict__ x, double* __restrict__ y, long nrows, long ntiles, double* part) {
    __shared__ double prod[TILE + 4];
    __shared__ double s4[4];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = ip[r0], p_hi = ip[rend];
        int my_lo = p_hi, my_hi = p_hi;
        if (r < rend) { my_lo = ip[r]; my_hi = ip[r + 1]; }
        double sum = 0.0;
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
        if (r < rend) { y[r] = sum; acc += x[r] * sum; }
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}

// physical position of chunk-relative product idx in the transposed staging buffer (conflict-free writes)
__device__ __forceinline__ int phys(int idx) { return (idx & 7) * BLOCK + (idx >> 3); }

// VC: 0 raw fp64 values, 1 uint8 codes into a dictionary.  WL: 16-byte window loads per lane (WL*512 >= L).
// LDSDMA: windows go to LDS with global_load_lds (no VGPR round trip).
template <int VC, int WL, int LDSDMA>
__global__ __launch_bounds__(BLOCK) void spmv_win(const int* __restrict__ ip, const uint16_t* __restrict__ sl, const double* __restrict__ dv,
                                                  const uint8_t* __restrict__ vc, const double* __restrict__ dict, const WDesc* __restrict__ wd,
                                                  const double* __restrict__ x, double* __restrict__ y, long nrows, long ntiles, double* part) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* prod = smem;                 // TILE
    double* xw = smem + TILE;            // WL * 512
    __shared__ double s4[4];
    __shared__ int sptr[BLOCK + 1];
    __shared__ double sdict[256];
    const int tid = threadIdx.x;
    if (VC) { sdict[tid] = dict[tid]; }
    double acc = 0.0;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = ip[r0], p_hi = ip[rend];
        const int my_lo = ip[(r < rend) ? r : rend];
        const double xr = (r < rend) ? x[r] : 0.0;
        const WDesc* d = wd + tile;
        const int L = d->L;
        // ---- windows: 16-byte coalesced loads of the concatenated cover
        d2v w[WL];
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const int pos = 2 * (k * BLOCK + tid);
            int dl = d->delta[0];
#pragma unroll
            for (int q = 1; q < NW; ++q) dl = (pos >= d->off[q]) ? d->delta[q] : dl;
            if (LDSDMA) {
                if (pos + 2 <= L) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + pos + dl), (__attribute__((address_space(3))) void*)(xw + 2 * (k * BLOCK + (tid & ~63))), 16, 0, 0);
            } else {
                w[k].x = w[k].y = 0.0;
                if (pos + 2 <= L) w[k] = *(const d2v*)(x + pos + dl);
                else if (pos < L) w[k].x = x[pos + dl];
            }
        }
        double sum = 0.0;
        int my_hi = p_hi;
        const int abase = p_lo & ~7;
        for (int base = abase; base < p_hi; base += TILE) {
            const int cnt = min(TILE, p_hi - base);
            int j = 8 * tid; j = (j < cnt) ? j : ((cnt - 1) & ~7);
            const u4v s = *(const u4v*)(sl + base + j);
            d2v val[4];
            u2v code;
            if (VC) code = *(const u2v*)(vc + base + j);
            else {
#pragma unroll
                for (int h = 0; h < 4; ++h) val[h] = *(const d2v*)(dv + base + j + 2 * h);
            }
            if (base == abase) {
                if (!LDSDMA) {
#pragma unroll
                    for (int k = 0; k < WL; ++k) { const int pos = 2 * (k * BLOCK + tid); if (pos < L) *(d2v*)(xw + pos) = w[k]; }
                }
                sptr[tid] = my_lo;
                if (tid == 0) sptr[BLOCK] = p_hi;
                if (LDSDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                my_hi = sptr[tid + 1];
            }
            const unsigned sw[4] = {s.x, s.y, s.z, s.w};
            double pr[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const double x0 = xw[sw[h] & 0xffffu], x1 = xw[sw[h] >> 16];
                double v0, v1;
                if (VC) {
                    const unsigned cw = (h < 2) ? code.x : code.y;
                    v0 = sdict[(cw >> (16 * (h & 1))) & 0xffu]; v1 = sdict[(cw >> (16 * (h & 1) + 8)) & 0xffu];
                } else { v0 = val[h].x; v1 = val[h].y; }
                pr[2 * h] = v0 * x0; pr[2 * h + 1] = v1 * x1;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) prod[i * BLOCK + tid] = pr[i];
            __syncthreads();
            const int lo = max(my_lo, max(base, p_lo)) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int idx = lo + k; t[k] = prod[phys((idx < TILE && idx >= 0) ? idx : 0)]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const double s2 = sum + t[k]; sum = (k < len) ? s2 : sum; }
            for (int k = 8; k < len; ++k) sum += prod[phys(lo + k)];
            __syncthreads();
        }
        if (r < rend) { y[r] = sum; acc += xr * sum; }
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}


// ================================================================================================= version 2
// Windows are laid out in LDS in chunks of 128 doubles (one wave-level 16-byte load each); chunk c of a tile is
// loaded by wave c % 4.  Per tile and wave: an int4 with the global start of its (<= 4) chunks and a dword with
// their half-lengths (bytes).  Everything about a chunk is wave-uniform (scalar registers): no per-lane searches.
constexpr int CMAX = 16;

__global__ __launch_bounds__(BLOCK) void cover2_kernel(const int* __restrict__ ip, const int* __restrict__ ix, long nrows, long ntiles,
                                                       int cmax, int* __restrict__ wg, unsigned* __restrict__ wn, uint16_t* __restrict__ sl,
                                                       int* __restrict__ stats) {
    __shared__ int key[CAP];
    __shared__ int heads[CMAX + 1];
    __shared__ int wst[CMAX], wof[CMAX];
    __shared__ int nh, s_ok, s_nw;
    const int tid = threadIdx.x;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows);
        const int p_lo = ip[r0], p_hi = ip[rend], cnt = p_hi - p_lo;
        if (tid < CMAX) wg[tile * CMAX + tid] = 0;
        if (tid < 4) wn[tile * 4 + tid] = 0;
        if (cnt <= 0 || cnt > CAP) continue;
        int n2 = 2; while (n2 < cnt) n2 <<= 1;
        for (int i = tid; i < n2; i += BLOCK) key[i] = (i < cnt) ? (ix[p_lo + i] >> 1) : 0x7fffffff;
        __syncthreads();
        bitonic_sort(key, n2);
        bool done = false;
        for (int G = 4; G <= 16384 && !done; G <<= 2) {
            if (tid == 0) nh = 0;
            __syncthreads();
            for (int i = tid; i < cnt; i += BLOCK)
                if (i == 0 || key[i] - key[i - 1] > G) { const int q = atomicAdd(&nh, 1); if (q < cmax) heads[q] = i; }
            __syncthreads();
            const bool fits = (nh <= cmax);
            if (fits && tid == 0) {
                const int nw = nh;
                for (int a = 1; a < nw; ++a) { const int v = heads[a]; int b = a - 1; while (b >= 0 && heads[b] > v) { heads[b + 1] = heads[b]; --b; } heads[b + 1] = v; }
                heads[nw] = cnt;
                int C = 0;
                for (int k = 0; k < nw; ++k) {
                    const int st = key[heads[k]] * 2, en = (key[heads[k + 1] - 1] + 1) * 2;
                    wst[k] = st; wof[k] = C * 128; C += (en - st + 127) / 128;
                }
                s_ok = (C <= cmax) ? 1 : 0; s_nw = nw;
                if (C <= cmax) {
                    for (int k = 0; k < nw; ++k) {
                        const int en = (key[heads[k + 1] - 1] + 1) * 2;
                        for (int c = wof[k] / 128, g = wst[k]; g < en; ++c, g += 128) {
                            const int wv = c & 3, i = c >> 2, half = min(128, en - g) / 2;
                            wg[tile * CMAX + wv * 4 + i] = g;
                            wn[tile * 4 + wv] |= (unsigned)half << (8 * i);
                        }
                    }
                    atomicMax(&stats[0], C); atomicAdd(&stats[1], 1);
                }
            }
            if (!fits && tid == 0) s_ok = 0;
            __syncthreads();
            done = (s_ok != 0);
            __syncthreads();
        }
        if (done) {
            const int nw = s_nw;
            for (int j = tid; j < cnt; j += BLOCK) {
                const int col = ix[p_lo + j];
                int k = 0;
                for (int q = 1; q < nw; ++q) k += (wst[q] <= col) ? 1 : 0;
                sl[p_lo + j] = (uint16_t)(wof[k] + col - wst[k]);
            }
        }
        __syncthreads();
    }
}

template <int VC, int PF>
__global__ __launch_bounds__(BLOCK) void spmv_w2(const int* __restrict__ ip, const uint16_t* __restrict__ sl, const double* __restrict__ dv,
                                                 const uint8_t* __restrict__ vc, const double* __restrict__ dict, const int* __restrict__ wg,
                                                 const unsigned* __restrict__ wn, const double* __restrict__ x, double* __restrict__ y,
                                                 long nrows, long ntiles, int nchunk, double* part) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* prod = smem;                 // TILE
    double* xw = smem + TILE;            // nchunk * 128
    __shared__ double s4[4];
    __shared__ int sptr[BLOCK + 1];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (VC) { sdict[tid] = dict[tid]; }
    double acc = 0.0;
    struct Meta { int p_lo, p_hi, my_lo; };
    auto load_meta = [&](long tile, Meta& m) {
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (tile < ntiles) {
            const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
            m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = ip[(r < rend) ? r : rend];
        }
    };
    Meta cur, nxt;
    long tile = blockIdx.x;
    load_meta(tile, cur);
    for (; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = cur.p_lo, p_hi = cur.p_hi, my_lo = cur.my_lo;
        const double xr = (r < rend) ? x[r] : 0.0;
        // ---- windows: this wave's chunks (scalar descriptor)
        const i4v g = *(const i4v*)(wg + (tile * 4 + wv) * 4);
        const unsigned nvw = wn[tile * 4 + wv];
        const int gs[4] = {g.x, g.y, g.z, g.w};
        d2v w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hc = (int)((nvw >> (8 * i)) & 0xffu);
            w[i].x = w[i].y = 0.0;
            if (hc > 0) { const int l2 = min(lane, hc - 1); w[i] = *(const d2v*)(x + gs[i] + 2 * l2); }
        }
        double sum = 0.0;
        int my_hi = p_hi;
        const int abase = p_lo & ~7;
        for (int base = abase; base < p_hi; base += TILE) {
            const int cnt = min(TILE, p_hi - base);
            int j = 8 * tid; j = (j < cnt) ? j : ((cnt - 1) & ~7);
            const u4v s = *(const u4v*)(sl + base + j);
            d2v val[4];
            u2v code;
            if (VC) code = *(const u2v*)(vc + base + j);
            else {
#pragma unroll
                for (int h = 0; h < 4; ++h) val[h] = *(const d2v*)(dv + base + j + 2 * h);
            }
            if (base == abase) {
                load_meta(tile + gridDim.x, nxt);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hc = (int)((nvw >> (8 * i)) & 0xffu);
                    if (hc > 0) *(d2v*)(xw + (wv + 4 * i) * 128 + 2 * lane) = w[i];
                }
                sptr[tid] = my_lo;
                if (tid == 0) sptr[BLOCK] = p_hi;
                __syncthreads();
                my_hi = sptr[tid + 1];
            }
            const unsigned sw[4] = {s.x, s.y, s.z, s.w};
            double pr[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const double x0 = xw[sw[h] & 0xffffu], x1 = xw[sw[h] >> 16];
                double v0, v1;
                if (VC) {
                    const unsigned cw = (h < 2) ? code.x : code.y;
                    v0 = sdict[(cw >> (16 * (h & 1))) & 0xffu]; v1 = sdict[(cw >> (16 * (h & 1) + 8)) & 0xffu];
                } else { v0 = val[h].x; v1 = val[h].y; }
                pr[2 * h] = v0 * x0; pr[2 * h + 1] = v1 * x1;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) prod[i * BLOCK + tid] = pr[i];
            __syncthreads();
            const int lo = max(my_lo, max(base, p_lo)) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int idx = lo + k; t[k] = prod[phys((idx < TILE && idx >= 0) ? idx : 0)]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const double s2 = sum + t[k]; sum = (k < len) ? s2 : sum; }
            for (int k = 8; k < len; ++k) sum += prod[phys(lo + k)];
            __syncthreads();
        }
        if (r < rend) { y[r] = sum; acc += xr * sum; }
        cur = nxt;
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}


// version 3: v2 + register prefetch of the next tile's whole input (matrix stream and windows) issued as soon as the
// current tile's registers have been consumed, and two barriers per tile.  (ubench: every tile covered, nnz <= TILE)
template <int VC, int MAP, int NT>
__global__ __launch_bounds__(BLOCK) void spmv_w3(const int* __restrict__ ip, const uint16_t* __restrict__ sl, const double* __restrict__ dv,
                                                 const uint8_t* __restrict__ vc, const double* __restrict__ dict, const int* __restrict__ wg,
                                                 const unsigned* __restrict__ wn, const double* __restrict__ x, double* __restrict__ y,
                                                 long nrows, long ntiles, int nchunk, double* part) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* prod = smem;                 // TILE
    double* xw = smem + TILE;            // nchunk * 128
    __shared__ double s4[4];
    __shared__ int sptr[BLOCK + 1];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (VC) { sdict[tid] = dict[tid]; }
    double acc = 0.0;
    struct Meta { int p_lo, p_hi, my_lo; };
    struct Regs { u4v s; d2v val[4]; u2v code; d2v w[4]; unsigned nvw; double xr; };
    auto load_meta = [&](long tile, Meta& m) {
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (tile < ntiles) {
            const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
            m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = ip[(r < rend) ? r : rend];
        }
    };
    auto issue = [&](long tile, const Meta& m, Regs& R) {
        R.nvw = 0;
        if (tile >= ntiles) return;
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        R.xr = (r < rend) ? x[r] : 0.0;
        const i4v g = *(const i4v*)(wg + (tile * 4 + wv) * 4);
        R.nvw = wn[tile * 4 + wv];
        const int gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hc = (int)((R.nvw >> (8 * i)) & 0xffu);
            if (hc > 0) { const int l2 = min(lane, hc - 1); R.w[i] = *(const d2v*)(x + gs[i] + 2 * l2); }
        }
        const int base = m.p_lo & ~7, cnt = m.p_hi - base;
        int j = 8 * tid; j = (j < cnt) ? j : ((cnt - 1) & ~7);
        j = j < 0 ? 0 : j;
        if (NT) {
            R.s = __builtin_nontemporal_load((const u4v*)(sl + base + j));
            if (VC) R.code = __builtin_nontemporal_load((const u2v*)(vc + base + j));
            else {
#pragma unroll
                for (int h = 0; h < 4; ++h) R.val[h] = __builtin_nontemporal_load((const d2v*)(dv + base + j + 2 * h));
            }
        } else {
            R.s = *(const u4v*)(sl + base + j);
            if (VC) R.code = *(const u2v*)(vc + base + j);
            else {
#pragma unroll
                for (int h = 0; h < 4; ++h) R.val[h] = *(const d2v*)(dv + base + j + 2 * h);
            }
        }
    };
    Meta m0, m1, m2;
    Regs R;
    const long G = gridDim.x;
    long tile = (MAP && G % 8 == 0) ? (long)(blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    load_meta(tile, m0); load_meta(tile + G, m1);
    issue(tile, m0, R);
    for (; tile < ntiles; tile += G) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = m0.p_lo, p_hi = m0.p_hi, my_lo = m0.my_lo;
        const int base = p_lo & ~7, cnt = p_hi - base;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hc = (int)((R.nvw >> (8 * i)) & 0xffu);
            if (hc > 0) *(d2v*)(xw + (wv + 4 * i) * 128 + 2 * lane) = R.w[i];
        }
        sptr[tid] = my_lo;
        if (tid == 0) sptr[BLOCK] = p_hi;
        __syncthreads();
        const int my_hi = sptr[tid + 1];
        const unsigned sw[4] = {R.s.x, R.s.y, R.s.z, R.s.w};
        double pr[8];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double x0 = xw[sw[h] & 0xffffu], x1 = xw[sw[h] >> 16];
            double v0, v1;
            if (VC) {
                const unsigned cw = (h < 2) ? R.code.x : R.code.y;
                v0 = sdict[(cw >> (16 * (h & 1))) & 0xffu]; v1 = sdict[(cw >> (16 * (h & 1) + 8)) & 0xffu];
            } else { v0 = R.val[h].x; v1 = R.val[h].y; }
            pr[2 * h] = v0 * x0; pr[2 * h + 1] = v1 * x1;
        }
        const double xr = R.xr;
#pragma unroll
        for (int i = 0; i < 8; ++i) prod[i * BLOCK + tid] = pr[i];
        // next tile's input goes in flight now; it lands while this tile's row sums are formed
        load_meta(tile + 2 * G, m2);
        issue(tile + G, m1, R);
        __syncthreads();
        const int lo = max(my_lo, p_lo) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
        double t[8], sum = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int idx = lo + k; t[k] = prod[phys((idx < TILE && idx >= 0) ? idx : 0)]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double s2 = sum + t[k]; sum = (k < len) ? s2 : sum; }
        for (int k = 8; k < len; ++k) sum += prod[phys(lo + k)];
        if (r < rend) { y[r] = sum; acc += xr * sum; }
        m0 = m1; m1 = m2;
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}


// version 4: v3 with TWO tiles in flight (register sets RA / RB, loop unrolled by two)
template <int VC>
__global__ __launch_bounds__(BLOCK) void spmv_w4(const int* __restrict__ ip, const uint16_t* __restrict__ sl, const double* __restrict__ dv,
                                                 const uint8_t* __restrict__ vc, const double* __restrict__ dict, const int* __restrict__ wg,
                                                 const unsigned* __restrict__ wn, const double* __restrict__ x, double* __restrict__ y,
                                                 long nrows, long ntiles, int nchunk, double* part) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* prod = smem;                 // TILE
    double* xw = smem + TILE;            // nchunk * 128
    __shared__ double s4[4];
    __shared__ int sptr[BLOCK + 1];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (VC) { sdict[tid] = dict[tid]; }
    double acc = 0.0;
    struct Meta { int p_lo, p_hi, my_lo; };
    struct Regs { u4v s; d2v val[4]; u2v code; d2v w[4]; unsigned nvw; double xr; };
    auto load_meta = [&](long tile, Meta& m) {
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (tile < ntiles) {
            const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
            m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = ip[(r < rend) ? r : rend];
        }
    };
    auto issue = [&](long tile, const Meta& m, Regs& R) {
        R.nvw = 0;
        if (tile >= ntiles) return;
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        R.xr = (r < rend) ? x[r] : 0.0;
        const i4v g = *(const i4v*)(wg + (tile * 4 + wv) * 4);
        R.nvw = wn[tile * 4 + wv];
        const int gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hc = (int)((R.nvw >> (8 * i)) & 0xffu);
            if (hc > 0) { const int l2 = min(lane, hc - 1); R.w[i] = *(const d2v*)(x + gs[i] + 2 * l2); }
        }
        const int base = m.p_lo & ~7, cnt = m.p_hi - base;
        int j = 8 * tid; j = (j < cnt) ? j : ((cnt - 1) & ~7);
        j = j < 0 ? 0 : j;
        R.s = *(const u4v*)(sl + base + j);
        if (VC) R.code = *(const u2v*)(vc + base + j);
        else {
#pragma unroll
            for (int h = 0; h < 4; ++h) R.val[h] = *(const d2v*)(dv + base + j + 2 * h);
        }
    };
    const long G = gridDim.x;
    long tile = blockIdx.x;
    Meta m0, m1, m2, m3;
    Regs RA, RB;
    load_meta(tile, m0); load_meta(tile + G, m1); load_meta(tile + 2 * G, m2);
    issue(tile, m0, RA); issue(tile + G, m1, RB);
    auto body = [&](long tile, const Meta& mc, Regs& R, const Meta& mnext2, long tnext2, Meta& mload, long tload) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = mc.p_lo, p_hi = mc.p_hi, my_lo = mc.my_lo;
        const int base = p_lo & ~7, cnt = p_hi - base;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hc = (int)((R.nvw >> (8 * i)) & 0xffu);
            if (hc > 0) *(d2v*)(xw + (wv + 4 * i) * 128 + 2 * lane) = R.w[i];
        }
        sptr[tid] = my_lo;
        if (tid == 0) sptr[BLOCK] = p_hi;
        __syncthreads();
        const int my_hi = sptr[tid + 1];
        const unsigned sw[4] = {R.s.x, R.s.y, R.s.z, R.s.w};
        double pr[8];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double x0 = xw[sw[h] & 0xffffu], x1 = xw[sw[h] >> 16];
            double v0, v1;
            if (VC) {
                const unsigned cw = (h < 2) ? R.code.x : R.code.y;
                v0 = sdict[(cw >> (16 * (h & 1))) & 0xffu]; v1 = sdict[(cw >> (16 * (h & 1) + 8)) & 0xffu];
            } else { v0 = R.val[h].x; v1 = R.val[h].y; }
            pr[2 * h] = v0 * x0; pr[2 * h + 1] = v1 * x1;
        }
        const double xr = R.xr;
#pragma unroll
        for (int i = 0; i < 8; ++i) prod[i * BLOCK + tid] = pr[i];
        load_meta(tload, mload);
        issue(tnext2, mnext2, R);                 // two tiles ahead, into the register set just consumed
        __syncthreads();
        const int lo = max(my_lo, p_lo) - base, hi = min(my_hi, base + cnt) - base, len = hi - lo;
        double t[8], sum = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int idx = lo + k; t[k] = prod[phys((idx < TILE && idx >= 0) ? idx : 0)]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double s2 = sum + t[k]; sum = (k < len) ? s2 : sum; }
        for (int k = 8; k < len; ++k) sum += prod[phys(lo + k)];
        if (r < rend) { y[r] = sum; acc += xr * sum; }
    };
    for (; tile < ntiles; tile += 2 * G) {
        body(tile, m0, RA, m2, tile + 2 * G, m3, tile + 3 * G);        // consumes RA, refills it with tile+2G (meta m2); loads meta m3
        if (tile + G < ntiles) body(tile + G, m1, RB, m3, tile + 3 * G, m0, tile + 4 * G);   // consumes RB, refills with tile+3G (m3); loads m0 := meta(tile+4G)
        else break;
        // rotate: next iteration's current = tile+2G (in RA, meta m2), then tile+3G (RB, meta m3); m0 holds meta(tile+4G)
        Meta t0 = m0; m0 = m2; m1 = m3; m2 = t0;
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}


// ================================================================================================= version 5
// Wave-independent tiles: wave w of a workgroup owns rows [64 w, 64 w + 64) of each 256-row tile, with its own
// cover (windows of the 64 rows), its own LDS regions and NO workgroup barriers (LDS operations of one wave execute
// in order).  Thread t still owns row 256 tile + t, so the fused-dot partial sums are unchanged.
constexpr int CW = 6;                                    // window chunks per wave-tile
struct __attribute__((aligned(64))) QDesc { int g[CW]; unsigned short off[CW]; unsigned char half[CW]; unsigned char ok; unsigned char pad[64 - 4 * CW - 2 * CW - CW - 1]; };

__global__ __launch_bounds__(BLOCK) void cover5_kernel(const int* __restrict__ ip, const int* __restrict__ ix, long nrows, long nq,
                                                       QDesc* __restrict__ qd, uint16_t* __restrict__ sl, int* __restrict__ stats) {
    // one workgroup per 64-row quarter (simple; the builder runs once per matrix)
    __shared__ int key[1024];
    __shared__ int heads[CW + 1];
    __shared__ int wst[CW], wof[CW];
    __shared__ int nh, s_ok, s_nw;
    const int tid = threadIdx.x;
    for (long q = blockIdx.x; q < nq; q += gridDim.x) {
        const long r0 = q * 64, rend = min(r0 + 64, nrows);
        __syncthreads();
        if (tid == 0) { QDesc d; memset(&d, 0, sizeof(d)); qd[q] = d; }
        if (r0 >= nrows) continue;
        const int p_lo = ip[r0], p_hi = ip[rend], cnt = p_hi - p_lo;
        if (cnt <= 0 || p_hi - (p_lo & ~7) > 512) continue;
        int n2 = 2; while (n2 < cnt) n2 <<= 1;
        for (int i = tid; i < n2; i += BLOCK) key[i] = (i < cnt) ? (ix[p_lo + i] >> 1) : 0x7fffffff;
        __syncthreads();
        bitonic_sort(key, n2);
        bool done = false;
        for (int G = 4; G <= 16384 && !done; G <<= 2) {
            if (tid == 0) nh = 0;
            __syncthreads();
            for (int i = tid; i < cnt; i += BLOCK)
                if (i == 0 || key[i] - key[i - 1] > G) { const int k = atomicAdd(&nh, 1); if (k < CW) heads[k] = i; }
            __syncthreads();
            if (tid == 0) {
                bool ok = nh <= CW; const int nw = nh;
                if (ok) {
                    for (int a = 1; a < nw; ++a) { const int v = heads[a]; int b = a - 1; while (b >= 0 && heads[b] > v) { heads[b + 1] = heads[b]; --b; } heads[b + 1] = v; }
                    heads[nw] = cnt;
                    int L = 0, C = 0;
                    for (int k = 0; k < nw; ++k) {
                        const int st = key[heads[k]] * 2, en = (key[heads[k + 1] - 1] + 1) * 2;
                        wst[k] = st; wof[k] = L; L += en - st; C += (en - st + 127) / 128;
                    }
                    ok = C <= CW;
                    if (ok) {
                        QDesc d; memset(&d, 0, sizeof(d)); d.ok = 1;
                        int c = 0;
                        for (int k = 0; k < nw; ++k) {
                            const int en = (key[heads[k + 1] - 1] + 1) * 2;
                            for (int g = wst[k], o = wof[k]; g < en; g += 128, o += 128, ++c) { d.g[c] = g; d.off[c] = (unsigned short)o; d.half[c] = (unsigned char)(min(128, en - g) / 2); }
                        }
                        qd[q] = d;
                        atomicMax(&stats[0], L); atomicAdd(&stats[1], 1);
                    }
                }
                s_ok = ok ? 1 : 0; s_nw = nw;
            }
            __syncthreads();
            done = s_ok != 0;
            __syncthreads();
        }
        if (done) {
            const int nw = s_nw;
            for (int j = tid; j < cnt; j += BLOCK) {
                const int col = ix[p_lo + j];
                int k = 0;
                for (int t = 1; t < nw; ++t) k += (wst[t] <= col) ? 1 : 0;
                sl[p_lo + j] = (uint16_t)(wof[k] + col - wst[k]);
            }
        }
    }
}

template <int VC>
__global__ __launch_bounds__(BLOCK) void spmv_w5(const int* __restrict__ ip, const uint16_t* __restrict__ sl, const double* __restrict__ dv,
                                                 const uint8_t* __restrict__ vc, const double* __restrict__ dict, const QDesc* __restrict__ qd,
                                                 const double* __restrict__ x, double* __restrict__ y, long nrows, long ntiles, int wcap, double* part) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int PLD = 65, PSZ = 8 * PLD;               // per-wave product staging [8][65], column 64 = zeros
    __shared__ double s4[4];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* prod = smem + wv * (PSZ + wcap);
    double* xw = prod + PSZ;
    if (VC) { sdict[tid] = dict[tid]; __syncthreads(); }
    if (lane < 8) prod[lane * PLD + 64] = 0.0;
    double acc = 0.0;
    struct Meta { int p_lo, p_hi, my_lo; };
    struct Regs { u4v s; d2v val[4]; u2v code; d2v w[CW]; unsigned hv0, hv1; unsigned short off[CW]; double xr; };
    auto load_meta = [&](long tile, Meta& m) {
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (tile < ntiles) {
            const long r0 = tile * ROWS + wv * 64, rend = min(r0 + 64, nrows), r = r0 + lane;
            if (r0 < nrows) { m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = ip[(r < rend) ? r : rend]; }
        }
    };
    auto issue = [&](long tile, const Meta& m, Regs& R) {
        R.hv0 = R.hv1 = 0;
        if (tile >= ntiles) return;
        const long r = tile * ROWS + tid;
        R.xr = (r < nrows) ? x[r] : 0.0;
        const QDesc* d = qd + (tile * 4 + wv);
        unsigned h0 = 0, h1 = 0;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int hc = d->half[c];
            R.off[c] = d->off[c];
            if (c < 4) h0 |= (unsigned)hc << (8 * c); else h1 |= (unsigned)hc << (8 * (c - 4));
            if (hc > 0) { const int l2 = min(lane, hc - 1); R.w[c] = *(const d2v*)(x + d->g[c] + 2 * l2); }
        }
        R.hv0 = h0; R.hv1 = h1;
        const int base = m.p_lo & ~7, cnt = m.p_hi - base;
        int j = 8 * lane; j = (j < cnt) ? j : ((cnt - 1) & ~7);
        j = j < 0 ? 0 : j;
        R.s = *(const u4v*)(sl + base + j);
        if (VC) R.code = *(const u2v*)(vc + base + j);
        else {
#pragma unroll
            for (int h = 0; h < 4; ++h) R.val[h] = *(const d2v*)(dv + base + j + 2 * h);
        }
    };
    Meta m0, m1, m2;
    Regs R;
    long tile = blockIdx.x;
    const long G = gridDim.x;
    load_meta(tile, m0); load_meta(tile + G, m1);
    issue(tile, m0, R);
    for (; tile < ntiles; tile += G) {
        const long r = tile * ROWS + tid;
        const int p_lo = m0.p_lo, p_hi = m0.p_hi, my_lo = m0.my_lo;
        const int base = p_lo & ~7;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int hc = (int)(((c < 4 ? R.hv0 >> (8 * c) : R.hv1 >> (8 * (c - 4)))) & 0xffu);
            if (hc > 0 && lane < hc) *(d2v*)(xw + R.off[c] + 2 * lane) = R.w[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int my_hi = __shfl_down(my_lo, 1, 64);
        if (lane == 63) my_hi = p_hi;
        const unsigned sw[4] = {R.s.x, R.s.y, R.s.z, R.s.w};
        double pr[8];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double x0 = xw[sw[h] & 0xffffu], x1 = xw[sw[h] >> 16];
            double v0, v1;
            if (VC) {
                const unsigned cw = (h < 2) ? R.code.x : R.code.y;
                v0 = sdict[(cw >> (16 * (h & 1))) & 0xffu]; v1 = sdict[(cw >> (16 * (h & 1) + 8)) & 0xffu];
            } else { v0 = R.val[h].x; v1 = R.val[h].y; }
            pr[2 * h] = v0 * x0; pr[2 * h + 1] = v1 * x1;
        }
        const double xr = R.xr;
#pragma unroll
        for (int i = 0; i < 8; ++i) prod[i * PLD + lane] = pr[i];
        load_meta(tile + 2 * G, m2);
        issue(tile + G, m1, R);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int lo = my_lo - base, len = my_hi - my_lo;
        const int a = lo & 7;
        const int adA = a * PLD + (lo >> 3), adB = adA - (8 * PLD - 1);
        double t[8], sum = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int ad = (a + k >= 8) ? adB : adA;
            ad = (k < len) ? ad : 64;
            t[k] = prod[ad + k * PLD];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += t[k];
        for (int k = 8; k < len; ++k) { const int idx = lo + k; sum += prod[(idx & 7) * PLD + (idx >> 3)]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (r < nrows) { y[r] = sum; acc += xr * sum; }
        m0 = m1; m1 = m2;
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}


// ================================================================================================= version 6
// Row-phase-only kernel: slots, value codes (or raw values) and the x windows of a tile go straight to LDS with
// global_load_lds (no VGPR round trip, no per-nonzero staging work); after ONE barrier lane t walks row t left to
// right, reading slot, code, x and value from LDS.  No product staging, few registers (8 workgroups per CU).
constexpr int T6 = TILE + 16;                           // nonzeros staged per tile (stream start aligned to 16)
#define MK_DMA16(gptr, lptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int VC, int RUNL, int MAP, int WSKIP = 0>
__global__ __launch_bounds__(BLOCK, 8) void spmv_w6(const int* __restrict__ ip, const uint16_t* __restrict__ sl, const double* __restrict__ dv,
                                                    const uint8_t* __restrict__ vc, const double* __restrict__ dict, const int* __restrict__ wg,
                                                    const unsigned* __restrict__ wn, const double* __restrict__ x, double* __restrict__ y,
                                                    long nrows, long ntiles, int nchunk, double* part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    double* xw = (double*)smem6;                                     // nchunk * 128 doubles (+ 2)
    unsigned short* sslot = (unsigned short*)(xw + nchunk * 128 + 2);
    unsigned char* scode = (unsigned char*)(sslot + T6);             // VC: codes ; raw: values (16-byte aligned)
    double* sval = (double*)(sslot + T6);
    __shared__ double s4[4];
    __shared__ int sptr[BLOCK + 1];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (VC) sdict[tid] = dict[tid];
    double acc = 0.0;
    struct Meta { int p_lo, p_hi, my_lo; };
    auto load_meta = [&](long tile, Meta& m) {
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (tile < ntiles) {
            const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
            m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = ip[(r < rend) ? r : rend];
        }
    };
    Meta cur, nxt;
    const long G = gridDim.x;
    // runs of RUNL consecutive tiles per workgroup: position q -> tile (q / RUNL) * G * RUNL ... i.e. workgroup b handles
    // tiles [ (s*G + b)*RUNL, +RUNL ) for s = 0, 1, ...
    long q = 0;                                          // index within this workgroup's sequence
    const long bx = (MAP && G % 8 == 0) ? (long)(blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;   // XCD-contiguous blocks per step
    auto tile_of = [&](long qq) { return ((qq / RUNL) * G + bx) * RUNL + (qq % RUNL); };
    long tile = tile_of(q);
    load_meta(tile, cur);
    for (; tile < ntiles; ++q, tile = tile_of(q)) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = cur.p_lo, p_hi = cur.p_hi, my_lo = cur.my_lo;
        const int base = p_lo & ~15, cnt = p_hi - base;
        const double xr = (WSKIP >= 2) ? 1.0 : ((r < rend) ? x[r] : 0.0);
        load_meta(tile_of(q + 1), nxt);
        // ---- everything the tile needs, straight into LDS
        {
            const i4v g = *(const i4v*)(wg + (tile * 4 + wv) * 4);
            const unsigned nvw = wn[tile * 4 + wv];
            const int gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hc = (int)((nvw >> (8 * i)) & 0xffu);
                if (hc > 0 && i < 4 - WSKIP) { const int l2 = min(lane, hc - 1); MK_DMA16(x + gs[i] + 2 * l2, xw + (wv + 4 * i) * 128); }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {                            // slots: 512 per wave-level copy
                const int c0 = (wv + 4 * c) * 512;
                if (c0 < cnt) MK_DMA16(sl + base + c0 + 8 * lane, sslot + c0);
            }
            if (VC) {
                const int c0 = wv * 1024;                            // codes: 1024 per wave-level copy
                if (c0 < cnt) MK_DMA16(vc + base + c0 + 16 * lane, scode + c0);
            } else {
#pragma unroll
                for (int c = 0; c < 5; ++c) {                        // values: 128 per wave-level copy
                    const int c0 = (wv + 4 * c) * 128;
                    if (c0 < cnt) MK_DMA16(dv + base + c0 + 2 * lane, sval + c0);
                }
            }
        }
        sptr[tid] = my_lo;
        if (tid == 0) sptr[BLOCK] = p_hi;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int my_hi = sptr[tid + 1];
        // ---- row phase
        const int lo = my_lo - base, len = my_hi - my_lo;
        double sum = 0.0;
        unsigned short sk[8];
        unsigned char ck[8];
        double vk[8], xk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sk[k] = sslot[lo + k]; if (VC) ck[k] = scode[lo + k]; else vk[k] = sval[lo + k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { xk[k] = xw[sk[k]]; if (VC) vk[k] = sdict[ck[k]]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double t = vk[k] * xk[k]; sum += (k < len) ? t : 0.0; }
        for (int k = 8; k < len; ++k) sum += (VC ? sdict[scode[lo + k]] : sval[lo + k]) * xw[sslot[lo + k]];
        if (r < rend) { y[r] = sum; acc += xr * sum; }
        __syncthreads();
        cur = nxt;
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}


// ================================================================================================= version 7
// w6 with ONE 32-bit word per nonzero {slot:16 | code:8} (one conflict-free ds_read_b32 per nonzero instead of a
// u16 and a u8 read) and, for dictionaries of <= 2 values, the value selected in registers instead of read from LDS.
__global__ void pack_kernel(long nnz, const uint16_t* sl, const uint8_t* vc, unsigned* pk) {
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < nnz; j += (long)gridDim.x * 256) pk[j] = (unsigned)sl[j] | ((unsigned)vc[j] << 16);
}

template <int ND2, int MAP>
__global__ __launch_bounds__(BLOCK, 8) void spmv_w7(const int* __restrict__ ip, const unsigned* __restrict__ pk, const double* __restrict__ dict,
                                                    const int* __restrict__ wg, const unsigned* __restrict__ wn, const double* __restrict__ x,
                                                    double* __restrict__ y, long nrows, long ntiles, int nchunk, double* part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem7[];
    double* xw = (double*)smem7;
    unsigned* spk = (unsigned*)(xw + nchunk * 128 + 2);
    __shared__ double s4[4];
    __shared__ int sptr[BLOCK + 1];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (!ND2) sdict[tid] = dict[tid];
    const double d0 = dict[0], d1 = dict[1];
    double acc = 0.0;
    struct Meta { int p_lo, p_hi, my_lo; };
    auto load_meta = [&](long tile, Meta& m) {
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (tile < ntiles) {
            const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
            m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = ip[(r < rend) ? r : rend];
        }
    };
    Meta cur, nxt;
    const long G = gridDim.x;
    const long bx = (MAP && G % 8 == 0) ? (long)(blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    long tile = bx;
    load_meta(tile, cur);
    for (; tile < ntiles; tile += G) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = cur.p_lo, p_hi = cur.p_hi, my_lo = cur.my_lo;
        const int base = p_lo & ~3, cnt = p_hi - base;
        const double xr = (r < rend) ? x[r] : 0.0;
        load_meta(tile + G, nxt);
        {
            const i4v g = *(const i4v*)(wg + (tile * 4 + wv) * 4);
            const unsigned nvw = wn[tile * 4 + wv];
            const int gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hc = (int)((nvw >> (8 * i)) & 0xffu);
                if (hc > 0) { const int l2 = min(lane, hc - 1); MK_DMA16(x + gs[i] + 2 * l2, xw + (wv + 4 * i) * 128); }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {                            // packed words: 256 per wave-level copy
                const int c0 = (wv + 4 * c) * 256;
                if (c0 < cnt) MK_DMA16(pk + base + c0 + 4 * lane, spk + c0);
            }
        }
        sptr[tid] = my_lo;
        if (tid == 0) sptr[BLOCK] = p_hi;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int my_hi = sptr[tid + 1];
        const int lo = my_lo - base, len = my_hi - my_lo;
        double sum = 0.0;
        unsigned wk[8];
        double xk[8], vk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wk[k] = spk[lo + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { xk[k] = xw[wk[k] & 0xffffu]; if (ND2) vk[k] = (wk[k] >> 16) ? d1 : d0; else vk[k] = sdict[wk[k] >> 16]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double t = vk[k] * xk[k]; sum += (k < len) ? t : 0.0; }
        for (int k = 8; k < len; ++k) { const unsigned w = spk[lo + k]; sum += (ND2 ? ((w >> 16) ? d1 : d0) : sdict[w >> 16]) * xw[w & 0xffffu]; }
        if (r < rend) { y[r] = sum; acc += xr * sum; }
        __syncthreads();
        cur = nxt;
    }
    const double tot = block_sum(acc, s4);
    if (tid == 0) part[blockIdx.x] = tot;
}


// ================================================================================================= version 8
// w7 double-buffered: the next tile's DMA (packed words + windows) is issued into the other LDS buffer right after the
// barrier that publishes the current one, so it lands during the row phase; ONE barrier per tile.
template <int ND2, int MAP>
__global__ __launch_bounds__(BLOCK, 4) void spmv_w8(const int* __restrict__ ip, const unsigned* __restrict__ pk, const double* __restrict__ dict,
                                                    const int* __restrict__ wg, const unsigned* __restrict__ wn, const double* __restrict__ x,
                                                    double* __restrict__ y, long nrows, long ntiles, int nchunk, double* part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    const int xwsz = nchunk * 128 + 2;                               // doubles per window buffer
    const int bufbytes = xwsz * 8 + (TILE + 16) * 4;
    __shared__ double s4[4];
    __shared__ int sptr[2][BLOCK + 1];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (!ND2) sdict[tid] = dict[tid];
    const double d0 = dict[0], d1 = dict[1];
    double acc = 0.0;
    struct Meta { int p_lo, p_hi, my_lo; };
    auto load_meta = [&](long tile, Meta& m) {
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (tile < ntiles) {
            const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
            m.p_lo = ip[r0]; m.p_hi = ip[rend]; m.my_lo = ip[(r < rend) ? r : rend];
        }
    };
    auto issue = [&](long tile, const Meta& m, int b) {              // all of a tile's input -> LDS buffer b
        if (tile >= ntiles) return;
        double* xw = (double*)(smem8 + b * bufbytes);
        unsigned* spk = (unsigned*)(xw + xwsz);
        const i4v g = *(const i4v*)(wg + (tile * 4 + wv) * 4);
        const unsigned nvw = wn[tile * 4 + wv];
        const int gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hc = (int)((nvw >> (8 * i)) & 0xffu);
            if (hc > 0) { const int l2 = min(lane, hc - 1); MK_DMA16(x + gs[i] + 2 * l2, xw + (wv + 4 * i) * 128); }
        }
        const int base = m.p_lo & ~3, cnt = m.p_hi - base;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int c0 = (wv + 4 * c) * 256;
            if (c0 < cnt) MK_DMA16(pk + base + c0 + 4 * lane, spk + c0);
        }
        sptr[b][tid] = m.my_lo;
        if (tid == 0) sptr[b][BLOCK] = m.p_hi;
    };
    Meta cur, nxt, nx2;
    const long G = gridDim.x;
    const long bx = (MAP && G % 8 == 0) ? (long)(blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    long tile = bx;
    load_meta(tile, cur); load_meta(tile + G, nxt);
    issue(tile, cur, 0);
    int b = 0;
    for (; tile < ntiles; tile += G, b ^= 1) {
        const long r0 = tile * ROWS, rend = min(r0 + ROWS, nrows), r = r0 + tid;
        const int p_lo = cur.p_lo, my_lo = cur.my_lo;
        const int base = p_lo & ~3;
        const double xr = (r < rend) ? x[r] : 0.0;
        load_meta(tile + 2 * G, nx2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this tile's DMA (and the meta of the next) landed
        __syncthreads();
        issue(tile + G, nxt, b ^ 1);                                 // next tile -> other buffer, lands during the row phase
        const double* xw = (const double*)(smem8 + b * bufbytes);
        const unsigned* spk = (const unsigned*)(xw + xwsz);
        const int my_hi = sptr[b][tid + 1];
        const int lo = my_lo - base, len = my_hi - my_lo;
        double sum = 0.0;
        unsigned wk[8];
        double xk[8], vk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wk[k] = spk[lo + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { xk[k] = xw[wk[k] & 0xffffu]; if (ND2) vk[k] = (wk[k] >> 16) ? d1 : d0; else vk[k] = sdict[wk[k] >> 16]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double t = vk[k] * xk[k]; sum += (k < len) ? t : 0.0; }
        for (int k = 8; k < len; ++k) { const unsigned w = spk[lo + k]; sum += (ND2 ? ((w >> 16) ? d1 : d0) : sdict[w >> 16]) * xw[w & 0xffffu]; }
        if (r < rend) { y[r] = sum; acc += xr * sum; }
        cur = nxt; nxt = nx2;
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
    int *ip, *ix, *stats; double *dv, *x, *y, *y0, *part, *dict; uint16_t* sl; uint8_t* vc; WDesc* wd;
    CK(hipMalloc(&ip, (n + 1) * 4)); CK(hipMalloc(&ix, (nnz + 16) * 4)); CK(hipMalloc(&dv, (nnz + 4096) * 8));
    CK(hipMalloc(&sl, (nnz + 16) * 2)); CK(hipMalloc(&vc, nnz + 4096)); CK(hipMalloc(&wd, ntiles * sizeof(WDesc)));
    CK(hipMemset(sl, 0, (nnz + 16) * 2)); CK(hipMemset(vc, 0, nnz + 4096)); CK(hipMemset(dv, 0, (nnz + 4096) * 8)); CK(hipMemset(ix, 0, (nnz + 16) * 4));
    CK(hipMalloc(&x, (n + 2) * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&y0, n * 8)); CK(hipMalloc(&part, 8192 * 8));
    CK(hipMalloc(&dict, 256 * 8)); CK(hipMalloc(&stats, 8)); CK(hipMemset(stats, 0, 8));
    std::vector<double> hd(256, 0.0); hd[0] = -1.0; hd[1] = 6.0; CK(hipMemcpy(dict, hd.data(), 256 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gen3, dim3(4096), dim3(256), 0, 0, nx, ny, nz, ip, ix, dv, vc);
    const int xmode = argc > 5 ? atoi(argv[5]) : 0;       // 0: smooth x, 1: random bits (data-dependent clocks)
    std::vector<double> hx(n);
    if (xmode == 0) for (long i = 0; i < n; ++i) hx[i] = 1.0 + (double)(i % 977) * 1e-3;
    else { unsigned long long st = 88172645463325252ULL; for (long i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; hx[i] = ((double)(st >> 11) / 9007199254740992.0 - 0.5) * 1e3; } }
    CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice)); CK(hipDeviceSynchronize());
    float cms = timeit([&] { hipLaunchKernelGGL(cover_kernel, dim3(2048), dim3(BLOCK), 0, 0, ip, ix, n, ntiles, 8192, wd, sl, stats); }, 1);
    int hs[2]; CK(hipMemcpy(hs, stats, 8, hipMemcpyDeviceToHost));
    const double bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    printf("grid %ldx%ldx%ld: n=%ld nnz=%ld, B_spmv=%.1f MB; cover %.1f ms, max L=%d, covered tiles=%d/%ld (x2 runs)\n", nx, ny, nz, n, nnz, bytes / 1e6, cms, hs[0], hs[1], ntiles);
    std::vector<double> h0(n), h1(n);
    auto check = [&](double* yy) { CK(hipMemcpy(h1.data(), yy, n * 8, hipMemcpyDeviceToHost)); for (long i = 0; i < n; ++i) if (h0[i] != h1[i]) return "MISMATCH"; return "bit-exact"; };
    for (int g : {1024, 2048}) {
        float ms = timeit([&] { hipLaunchKernelGGL(spmv_base, dim3(g), dim3(BLOCK), 0, 0, ip, ix, dv, x, y0, n, ntiles, part); }, reps);
        printf("base            grid=%4d : %9.1f us  %.2f TB/s\n", g, ms * 1e3, bytes / ms / 1e9);
    }
    CK(hipMemcpy(h0.data(), y0, n * 8, hipMemcpyDeviceToHost));
#define RUNW(VC, WL, DMA) if (hs[0] <= WL * 512) for (int g : {1024, 1536, 2048}) { const size_t lds = (TILE + WL * 512) * 8; \
        CK(hipMemset(y, 0, n * 8)); \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_win<VC, WL, DMA>), dim3(g), dim3(BLOCK), lds, 0, ip, sl, dv, vc, dict, wd, x, y, n, ntiles, part); }, reps); \
        printf("win vc=%d WL=%d dma=%d grid=%4d : %9.1f us  %.2f TB/s  %s\n", VC, WL, DMA, g, ms * 1e3, bytes / ms / 1e9, check(y)); }
    int *wg; unsigned* wn; uint16_t* sl2;
    CK(hipMalloc(&wg, ntiles * CMAX * 4)); CK(hipMalloc(&wn, ntiles * 16)); CK(hipMalloc(&sl2, (nnz + 4096) * 2)); CK(hipMemset(sl2, 0, (nnz + 4096) * 2));
    CK(hipMemset(stats, 0, 8));
    cms = timeit([&] { hipLaunchKernelGGL(cover2_kernel, dim3(2048), dim3(BLOCK), 0, 0, ip, ix, n, ntiles, CMAX, wg, wn, sl2, stats); }, 1);
    CK(hipMemcpy(hs, stats, 8, hipMemcpyDeviceToHost));
    printf("cover2 %.1f ms, max chunks=%d, covered tiles=%d/%ld (x2 runs)\n", cms, hs[0], hs[1], ntiles);
#define RUN2(VC, PF) for (int g : {1024, 1280, 2048}) { const size_t lds = (TILE + hs[0] * 128) * 8; \
        CK(hipMemset(y, 0, n * 8)); \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_w2<VC, PF>), dim3(g), dim3(BLOCK), lds, 0, ip, sl2, dv, vc, dict, wg, wn, x, y, n, ntiles, hs[0], part); }, reps); \
        printf("w2 vc=%d pf=%d grid=%4d lds=%zu : %9.1f us  %.2f TB/s  %s\n", VC, PF, g, lds, ms * 1e3, bytes / ms / 1e9, check(y)); }
#define RUN3(VC, MAP, NT) for (int g : {1024, 1280, 2048}) { const size_t lds = (TILE + hs[0] * 128) * 8; \
        CK(hipMemset(y, 0, n * 8)); \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_w3<VC, MAP, NT>), dim3(g), dim3(BLOCK), lds, 0, ip, sl2, dv, vc, dict, wg, wn, x, y, n, ntiles, hs[0], part); }, reps); \
        printf("w3 vc=%d map=%d nt=%d grid=%4d lds=%zu : %9.1f us  %.2f TB/s  %s\n", VC, MAP, NT, g, lds, ms * 1e3, bytes / ms / 1e9, check(y)); }
#define RUN4(VC) for (int g : {1024, 1280, 2048}) { const size_t lds = (TILE + hs[0] * 128) * 8; \
        CK(hipMemset(y, 0, n * 8)); \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_w4<VC>), dim3(g), dim3(BLOCK), lds, 0, ip, sl2, dv, vc, dict, wg, wn, x, y, n, ntiles, hs[0], part); }, reps); \
        printf("w4 vc=%d grid=%4d lds=%zu : %9.1f us  %.2f TB/s  %s\n", VC, g, lds, ms * 1e3, bytes / ms / 1e9, check(y)); }
    { QDesc* qd; uint16_t* sl5; const long nq = ntiles * 4;
      CK(hipMalloc(&qd, nq * sizeof(QDesc))); CK(hipMalloc(&sl5, (nnz + 16) * 2)); CK(hipMemset(sl5, 0, (nnz + 16) * 2));
      CK(hipMemset(stats, 0, 8));
      float c5 = timeit([&] { hipLaunchKernelGGL(cover5_kernel, dim3(8192), dim3(BLOCK), 0, 0, ip, ix, n, nq, qd, sl5, stats); }, 1);
      int h5[2]; CK(hipMemcpy(h5, stats, 8, hipMemcpyDeviceToHost));
      const int wcap = (h5[0] + 1) & ~1;
      printf("cover5 %.1f ms, max L=%d, covered quarters=%d/%ld (x2 runs)\n", c5, h5[0], h5[1], nq);
#define RUN5(VC) for (int g : {1024, 1280, 1536, 2048}) { const size_t lds = 4 * (8 * 65 + wcap) * 8; \
        CK(hipMemset(y, 0, n * 8)); \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_w5<VC>), dim3(g), dim3(BLOCK), lds, 0, ip, sl5, dv, vc, dict, qd, x, y, n, ntiles, wcap, part); }, reps); \
        printf("w5 vc=%d grid=%4d lds=%zu : %9.1f us  %.2f TB/s  %s\n", VC, g, lds, ms * 1e3, bytes / ms / 1e9, check(y)); }
    }
#define RUN6(VC, RL, MP) for (int g : {1024, 1536, 2048}) { const size_t lds = (hs[0] * 128 + 2) * 8 + T6 * 2 + (VC ? T6 : T6 * 8) + 64; \
        CK(hipMemset(y, 0, n * 8)); \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_w6<VC, RL, MP>), dim3(g), dim3(BLOCK), lds, 0, ip, sl2, dv, vc, dict, wg, wn, x, y, n, ntiles, hs[0], part); }, reps); \
        printf("w6 vc=%d run=%d map=%d grid=%4d lds=%zu : %9.1f us  %.2f TB/s  %s\n", VC, RL, MP, g, lds, ms * 1e3, bytes / ms / 1e9, check(y)); }
    RUN6(1, 1, 0)
    { unsigned* pk; CK(hipMalloc(&pk, (nnz + 4096) * 4)); CK(hipMemset(pk, 0, (nnz + 4096) * 4));
      hipLaunchKernelGGL(pack_kernel, dim3(4096), dim3(256), 0, 0, nnz, sl2, vc, pk); CK(hipDeviceSynchronize());
#define RUN7(ND2, MP) for (int g : {1024}) { const size_t lds = (hs[0] * 128 + 2) * 8 + (TILE + 16) * 4 + 64; \
        CK(hipMemset(y, 0, n * 8)); \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_w7<ND2, MP>), dim3(g), dim3(BLOCK), lds, 0, ip, pk, dict, wg, wn, x, y, n, ntiles, hs[0], part); }, reps); \
        printf("w7 nd2=%d map=%d grid=%4d lds=%zu : %9.1f us  %.2f TB/s  %s\n", ND2, MP, g, lds, ms * 1e3, bytes / ms / 1e9, check(y)); }
      RUN7(1, 0)
#define RUN8(ND2, MP) for (int g : {768, 1024, 1280, 1536}) { const size_t lds = 2 * ((hs[0] * 128 + 2) * 8 + (TILE + 16) * 4); \
        CK(hipMemset(y, 0, n * 8)); \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_w8<ND2, MP>), dim3(g), dim3(BLOCK), lds, 0, ip, pk, dict, wg, wn, x, y, n, ntiles, hs[0], part); }, reps); \
        printf("w8 nd2=%d map=%d grid=%4d lds=%zu : %9.1f us  %.2f TB/s  %s\n", ND2, MP, g, lds, ms * 1e3, bytes / ms / 1e9, check(y)); }
      RUN8(1, 0) RUN8(1, 1) RUN8(0, 1)
    }
#define RUN6S(VC, WS) for (int g : {1024, 2048}) { const size_t lds = (hs[0] * 128 + 2) * 8 + T6 * 2 + (VC ? T6 : T6 * 8) + 64; \
        float ms = timeit([&] { hipLaunchKernelGGL((spmv_w6<VC, 1, 0, WS>), dim3(g), dim3(BLOCK), lds, 0, ip, sl2, dv, vc, dict, wg, wn, x, y, n, ntiles, hs[0], part); }, reps); \
        printf("w6 vc=%d WSKIP=%d grid=%4d : %9.1f us  (timing only: window chunks i >= %d of each wave not loaded%s)\n", VC, WS, g, ms * 1e3, 4 - WS, WS >= 2 ? ", no x[r] load" : ""); }
    RUN6S(1, 4)
    RUN3(0, 0, 0) RUN3(1, 0, 0)
    return 0;
}
