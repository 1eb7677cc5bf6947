// spmv_cb -- micro-benchmark: column-blocked tiles for matrices whose x does not fit an XCD's L2 (BASELINE config 3:
// 1M rows, diagonal + 4 random columns per row, x = 8 MB against 4 MiB of L2 per XCD).
//
// Today's gather kernel pulls one 64-byte sector through the fabric per nonzero (rocprof: 371 MB of fabric reads for
// 80 MB of algorithmic bytes).  Here every 256-row tile stores its nonzeros grouped by column block (K blocks of
// n/K columns), and all workgroups walk the blocks in the same order with their row sums in registers, so at any time
// the whole chip gathers from ONE slice of x that fits L2.  Row sums still run left to right (columns are sorted, the
// blocks ascend): same bits as plain CSR.
//   lane : one row per lane, no LDS, no barriers
//   lds  : nonzeros loaded coalesced, products to LDS, row sums from LDS (two barriers per tile-block)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256, ROWS = 256, PCAP = 4096;

template <int TW>
__global__ __launch_bounds__(BLOCK) void cb_lane(int T, int K, int n, const int* __restrict__ seg, const uint16_t* __restrict__ rowoff,
                                                  const int* __restrict__ cols, const double* __restrict__ vals,
                                                  const double* __restrict__ x, double* __restrict__ y) {
    double sum[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) sum[i] = 0.0;
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            const int t = blockIdx.x + i * gridDim.x;
            if (t < T) {
                const int s = t * K + k, base = seg[s];
                const uint16_t* ro = rowoff + (size_t)s * (ROWS + 1);
                const int lo = ro[threadIdx.x], hi = ro[threadIdx.x + 1];
                for (int j = lo; j < hi; ++j) sum[i] += vals[base + j] * x[cols[base + j]];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const int t = blockIdx.x + i * gridDim.x;
        const long r = (long)t * ROWS + threadIdx.x;
        if (t < T && r < n) y[r] = sum[i];
    }
}

template <int TW>
__global__ __launch_bounds__(BLOCK) void cb_lds(int T, int K, int n, const int* __restrict__ seg, const uint16_t* __restrict__ rowoff,
                                                 const int* __restrict__ cols, const double* __restrict__ vals,
                                                 const double* __restrict__ x, double* __restrict__ y) {
    __shared__ double prod[2][PCAP];
    double sum[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) sum[i] = 0.0;
    int buf = 0;
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            const int t = blockIdx.x + i * gridDim.x;
            if (t < T) {                                     // (uniform per workgroup)
                const int s = t * K + k, base = seg[s], len = seg[s + 1] - base;
                const uint16_t* ro = rowoff + (size_t)s * (ROWS + 1);
                const int lo = ro[threadIdx.x], hi = ro[threadIdx.x + 1];
                double* p = prod[buf];
                for (int j = threadIdx.x; j < len; j += BLOCK) p[j] = vals[base + j] * x[cols[base + j]];
                __syncthreads();                             // one barrier per step: the buffers alternate
                for (int j = lo; j < hi; ++j) sum[i] += p[j];
                buf ^= 1;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const int t = blockIdx.x + i * gridDim.x;
        const long r = (long)t * ROWS + threadIdx.x;
        if (t < T && r < n) y[r] = sum[i];
    }
}

// wave : every wave owns 64-row sub-tiles; nonzeros loaded coalesced within the wave, products to a wave-private LDS
//        region, row sums from there -- coalesced matrix reads without any workgroup barrier
constexpr int WCAP = 1024;
template <int SW>
__global__ __launch_bounds__(BLOCK) void cb_wave(int S, int K, int n, const int* __restrict__ seg, const uint16_t* __restrict__ rowoff,
                                                  const int* __restrict__ cols, const double* __restrict__ vals,
                                                  const double* __restrict__ x, double* __restrict__ y) {
    __shared__ double prod[4][2][WCAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
    double sum[SW];
#pragma unroll
    for (int i = 0; i < SW; ++i) sum[i] = 0.0;
    int buf = 0;
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int i = 0; i < SW; ++i) {
            const int st = gw + i * nw;
            if (st < S) {
                const int s = st * K + k, base = seg[s], len = seg[s + 1] - base;
                const uint16_t* ro = rowoff + (size_t)s * 65;
                const int lo = ro[lane], hi = ro[lane + 1];
                double* p = prod[wv][buf];
                for (int j = lane; j < len; j += 64) p[j] = vals[base + j] * x[cols[base + j]];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int j = lo; j < hi; ++j) sum[i] += p[j];
                buf ^= 1;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < SW; ++i) {
        const int st = gw + i * nw;
        const long r = (long)st * 64 + lane;
        if (st < S && r < n) y[r] = sum[i];
    }
}

// res  : plain CSR, no format change.  The workgroup's tiles are loaded once into LDS (coalesced), then the K column
//        blocks are walked with a cursor per row: only the x gathers go to memory during the phases, and at any
//        time they all fall into one slice of x
constexpr int RCAP = 1536;
template <int TW, int U>
__global__ __launch_bounds__(BLOCK) void cb_res(int T, int K, int n, int W, const int* __restrict__ ip,
                                                 const int* __restrict__ cols, const double* __restrict__ vals,
                                                 const double* __restrict__ x, double* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* lv = (double*)smem;                              // [TW][RCAP]
    int* lc = (int*)(lv + TW * RCAP);                        // [TW][RCAP]
    int cur[TW], end[TW];
    double sum[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const int t = blockIdx.x + i * gridDim.x;
        sum[i] = 0.0;
        cur[i] = end[i] = 0;
        if (t < T) {
            const int row0 = t * ROWS, rowe = min(n, row0 + ROWS);
            const int e0 = ip[row0], e1 = ip[rowe];
            for (int j = threadIdx.x; j < e1 - e0; j += BLOCK) {
                lc[i * RCAP + j] = cols[e0 + j];
                lv[i * RCAP + j] = vals[e0 + j];
            }
            const int r = row0 + threadIdx.x;
            if (r < n) { cur[i] = ip[r] - e0; end[i] = ip[r + 1] - e0; }
        }
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const int c1 = min(n, (k + 1) * W);
        for (;;) {
            bool more = false;
            double xv[TW][U], vv[TW][U];
            bool ok[TW][U];
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = cur[i] + u;
                    int c = 0x7fffffff;
                    if (p < end[i]) c = lc[i * RCAP + p];
                    ok[i][u] = c < c1;
                    xv[i][u] = 0.0; vv[i][u] = 0.0;
                    if (ok[i][u]) { xv[i][u] = x[c]; vv[i][u] = lv[i * RCAP + p]; }
                }
#pragma unroll
            for (int i = 0; i < TW; ++i) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ok[i][u]) { sum[i] += vv[i][u] * xv[i][u]; cur[i] += 1; }
                more |= ok[i][U - 1];
            }
            if (!__any(more)) break;
        }
    }
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const int t = blockIdx.x + i * gridDim.x;
        const long r = (long)t * ROWS + threadIdx.x;
        if (t < T && r < n) y[r] = sum[i];
    }
}

// reg  : round 4.  Rows of <= RW entries held in REGISTERS, RT tiles per workgroup at once; the phase loop is static: in
//        phase k a lane issues the gathers of all its entries whose column lies in slice k at once, waits once, adds in
//        column order.  ING = 0: a lane loads its own row straight from the CSR arrays (strided, cached); 1: through LDS.
// soft XCD-local barrier: performance hint only (bounded spin), counters only grow: target = epoch * (workgroups per XCD)
__device__ __forceinline__ void soft_barrier(unsigned* cnt, int which, unsigned target) {
    if (threadIdx.x == 0) {
        unsigned* c = cnt + (which * 8 + (blockIdx.x & 7)) * 32;           // one 128-byte line per counter
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && wall_clock64() - t0 < 3000) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
}
template <int RT, int RW, int OCC, int ING = 0>
__global__ __launch_bounds__(BLOCK, OCC) void cb_reg(int T, int K, int n, int W, const int* __restrict__ ip,
                                                 const int* __restrict__ cols, const double* __restrict__ vals,
                                                 const double* __restrict__ x, double* __restrict__ y, long long* __restrict__ ts,
                                                 unsigned* sb = nullptr, unsigned epoch = 0, int sbmode = 0) {
    unsigned off[RT][RW];
    if (ts && threadIdx.x == 0) ts[blockIdx.x] = wall_clock64();
    double v[RT][RW], sum[RT];
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(x), 0, 0x7fffffff, 0x00020000);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (ING == 0) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int t = blockIdx.x + i * gridDim.x;
        const long r = (long)t * ROWS + threadIdx.x;
        sum[i] = 0.0;
        int cur = 0, fin = 0;
        if (t < T && r < n) { cur = ip[r]; fin = ip[r + 1]; }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const bool has = cur + j < fin;
            const int idx = has ? cur + j : 0;
            const unsigned cc = (unsigned)cols[idx] << 3;
            const double vv = vals[idx];
            off[i][j] = has ? cc : 0xffffffffu;
            v[i][j] = has ? vv : 0.0;
        }
    }
    } else {
    // staged as the library does (mk_spmv_fmt3r.h): columns of the RT tiles stay in LDS, values pass through a staging area
    int* lc = (int*)smem;                                    // [RT][RCAP]
    double* lv = (double*)(lc + (RT - 1) * RCAP);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int cur[RT], len[RT], base[RT], cnt[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int t = blockIdx.x + i * gridDim.x;
        sum[i] = 0.0; cur[i] = len[i] = base[i] = cnt[i] = 0;
#pragma unroll
        for (int j = 0; j < RW; ++j) v[i][j] = 0.0;
        if (t < T) {
            const int row0 = t * ROWS, rowe = min(n, row0 + ROWS);
            const int e0 = ip[row0], e1 = ip[rowe];
            base[i] = e0 & ~3; cnt[i] = e1 - base[i];
            const int lastv = (cnt[i] > 0) ? ((cnt[i] - 1) & ~1) : 0;
            for (int c0 = wv * 128; c0 < cnt[i]; c0 += 4 * 128) {
                int j = c0 + 2 * lane; j = j < lastv ? j : lastv;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vals + base[i] + j),
                                                 (__attribute__((address_space(3))) void*)(lv + c0), 16, 0, 0);
            }
            const int r = row0 + threadIdx.x;
            if (r < n) { cur[i] = ip[r] - base[i]; len[i] = ip[r + 1] - base[i] - cur[i]; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int j = 0; j < RW; ++j) { const double vv = lv[(j < len[i]) ? cur[i] + j : 0]; v[i][j] = (j < len[i]) ? vv : 0.0; }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int last = (cnt[i] > 0) ? ((cnt[i] - 1) & ~3) : 0;
        for (int c0 = wv * 256; c0 < cnt[i]; c0 += 4 * 256) {
            int j = c0 + 4 * lane; j = j < last ? j : last;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cols + base[i] + j),
                                             (__attribute__((address_space(3))) void*)(lc + i * RCAP + c0), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < RW; ++j)
            off[i][j] = (j < len[i]) ? (unsigned)lc[i * RCAP + cur[i] + j] << 3 : 0xffffffffu;
    }
    for (int k = 0; k < K; ++k) {
        if (sb && (k == 0 || sbmode == 2)) soft_barrier(sb, k, epoch * (gridDim.x / 8));
        if (ts && threadIdx.x == 0) ts[(size_t)(k + 1) * gridDim.x + blockIdx.x] = wall_clock64();
        const unsigned o_lo = (unsigned)(k * W) << 3;
        const unsigned o_hi = (k + 1 < K) ? (unsigned)((k + 1) * W) << 3 : 0xffffffffu;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            double xv[RW];
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const bool in = off[i][j] >= o_lo && off[i][j] < o_hi;
                xv[j] = 0.0;
                if (in) {
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    const u2 w = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(xres, (int)off[i][j], 0, 0));
                    xv[j] = __builtin_bit_cast(double, w);
                }
            }
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const bool in = off[i][j] >= o_lo && off[i][j] < o_hi;
                const double tt = sum[i] + v[i][j] * xv[j];
                sum[i] = in ? tt : sum[i];
            }
        }
    }
    if (ts && threadIdx.x == 0) ts[(size_t)(K + 1) * gridDim.x + blockIdx.x] = wall_clock64();
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int t = blockIdx.x + i * gridDim.x;
        const long r = (long)t * ROWS + threadIdx.x;
        if (t < T && r < n) y[r] = sum[i];
    }
}

// pair : the library's round-4 kernel (mk_spmv_fmt3r.h): tile A in LDS walked with a cursor, tile B (the workgroup's next
//        tile) in registers with static exec-masked gathers issued before A's walk.  PAIR = 0: tile A only (= the library's
//        format-3 kernel, one tile per step).  Stamps: 0 start, 1 after B's ingest, 2 after A's ingest, 3 .. K + 2 phase ends.
template <int PAIR, int AUX = 0>
__global__ __launch_bounds__(BLOCK, 8) void cb_pair(int T, int K, int n, int W, const int* __restrict__ ip,
                                                    const int* __restrict__ cols, const double* __restrict__ vals,
                                                    const double* __restrict__ x, double* __restrict__ y, long long* __restrict__ ts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* lv = (double*)smem;
    int* lc = (int*)(lv + RCAP);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(x), 0, 0x7fffffff, 0x00020000);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    auto ingest = [&](int t, int& cur, int& fin) {
        const int row0 = t * ROWS, rowe = min(n, row0 + ROWS);
        const int e0 = ip[row0], e1 = ip[rowe];
        const int base = e0 & ~3, cnt = e1 - base;
        const int last = (cnt > 0) ? ((cnt - 1) & ~3) : 0;
        for (int c0 = wv * 256; c0 < cnt; c0 += 4 * 256) {
            int j = c0 + 4 * lane; j = j < last ? j : last;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cols + base + j),
                                             (__attribute__((address_space(3))) void*)(lc + c0), 16, 0, 0);
        }
        const int lastv = (cnt > 0) ? ((cnt - 1) & ~1) : 0;
        for (int c0 = wv * 128; c0 < cnt; c0 += 4 * 128) {
            int j = c0 + 2 * lane; j = j < lastv ? j : lastv;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vals + base + j),
                                             (__attribute__((address_space(3))) void*)(lv + c0), 16, 0, 0);
        }
        cur = fin = 0;
        const int r = row0 + threadIdx.x;
        if (r < rowe) { cur = ip[r] - base; fin = ip[r + 1] - base; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    int stampi = 0;
    auto stamp = [&]() { if (ts && threadIdx.x == 0 && stampi < 16) ts[(size_t)stampi * gridDim.x + blockIdx.x] = wall_clock64(); ++stampi; };
    for (int pos = blockIdx.x; pos < T; pos += (PAIR ? 2 : 1) * gridDim.x) {
        const bool first = (pos == (int)blockIdx.x);
        if (first) stamp();
        unsigned ob[5]; double vb[5]; double sumb = 0.0;
#pragma unroll
        for (int j = 0; j < 5; ++j) { ob[j] = 0xffffffffu; vb[j] = 0.0; }
        const int posb = pos + gridDim.x;
        const bool has_b = PAIR && posb < T;
        if (has_b) {
            int cur, fin; ingest(posb, cur, fin);
#pragma unroll
            for (int j = 0; j < 5; ++j) { const bool has = cur + j < fin; const int idx = has ? cur + j : 0; const unsigned cc = (unsigned)lc[idx] << 3; const double vv = lv[idx]; ob[j] = has ? cc : 0xffffffffu; vb[j] = has ? vv : 0.0; }
            __syncthreads();
        }
        if (first) stamp();
        int cur, fin; double sum = 0.0;
        ingest(pos, cur, fin);
        if (first) stamp();
        for (int k = 0; k < K; ++k) {
            const int c1 = (k + 1 < K) ? (k + 1) * W : 0x7fffffff;
            const unsigned o_lo = (unsigned)(k * W) << 3, o_hi = (k + 1 < K) ? (unsigned)c1 << 3 : 0xffffffffu;
            double xb[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) { xb[j] = 0.0; if (PAIR && ob[j] >= o_lo && ob[j] < o_hi) { const u2 w = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(xres, (int)ob[j], 0, AUX)); xb[j] = __builtin_bit_cast(double, w); } }
            for (;;) {
                int ca = 0x7fffffff;
                if (cur < fin) ca = lc[cur];
                const bool oa = ca < c1;
                if (oa) { const u2 w = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(xres, ca << 3, 0, AUX)); sum += lv[cur] * __builtin_bit_cast(double, w); cur += 1; }
                if (!__any(oa)) break;
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) { const bool in = ob[j] >= o_lo && ob[j] < o_hi; const double t = sumb + vb[j] * xb[j]; sumb = in ? t : sumb; }
            if (first) stamp();
        }
        { const long r = (long)pos * ROWS + threadIdx.x; if (r < n) y[r] = sum; }
        if (has_b) { const long r = (long)posb * ROWS + threadIdx.x; if (r < n) y[r] = sumb; }
        __syncthreads();
    }
}

// pair2 : VERDICT r4 item 3 priced -- the two ingests of a pair OVERLAPPED: tile B's stream and tile A's stream are DMA'd
//         into TWO LDS buffers back to back, one wait, B's rows go to registers, then the phases walk A in the second
//         buffer.  Costs twice the LDS per workgroup (36 KB: 4 resident workgroups per CU instead of 8).  HALF = 1 prices
//         the occupancy loss alone: the production order of ingests, but with the doubled LDS allocation.
template <int HALF>
__global__ __launch_bounds__(BLOCK, 4) void cb_pair2(int T, int K, int n, int W, const int* __restrict__ ip,
                                                     const int* __restrict__ cols, const double* __restrict__ vals,
                                                     const double* __restrict__ x, double* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(x), 0, 0x7fffffff, 0x00020000);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    auto bufv = [&](int b) { return (double*)(smem + (size_t)b * RCAP * 12); };
    auto bufc = [&](int b) { return (int*)(bufv(b) + RCAP); };
    auto issue = [&](int t, int b, int& cur, int& fin) {
        double* lv = bufv(b); int* lc = bufc(b);
        const int row0 = t * ROWS, rowe = min(n, row0 + ROWS);
        const int e0 = ip[row0], e1 = ip[rowe];
        const int base = e0 & ~3, cnt = e1 - base;
        const int last = (cnt > 0) ? ((cnt - 1) & ~3) : 0;
        for (int c0 = wv * 256; c0 < cnt; c0 += 4 * 256) {
            int j = c0 + 4 * lane; j = j < last ? j : last;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cols + base + j),
                                             (__attribute__((address_space(3))) void*)(lc + c0), 16, 0, 0);
        }
        const int lastv = (cnt > 0) ? ((cnt - 1) & ~1) : 0;
        for (int c0 = wv * 128; c0 < cnt; c0 += 4 * 128) {
            int j = c0 + 2 * lane; j = j < lastv ? j : lastv;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vals + base + j),
                                             (__attribute__((address_space(3))) void*)(lv + c0), 16, 0, 0);
        }
        cur = fin = 0;
        const int r = row0 + threadIdx.x;
        if (r < rowe) { cur = ip[r] - base; fin = ip[r + 1] - base; }
    };
    for (int pos = blockIdx.x; pos < T; pos += 2 * gridDim.x) {
        unsigned ob[5]; double vb[5]; double sumb = 0.0;
#pragma unroll
        for (int j = 0; j < 5; ++j) { ob[j] = 0xffffffffu; vb[j] = 0.0; }
        const int posb = pos + gridDim.x;
        const bool has_b = posb < T;
        int curb = 0, finb = 0, cur, fin; double sum = 0.0;
        if (HALF) {                                          // production order, doubled allocation
            if (has_b) { issue(posb, 0, curb, finb); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        } else {
            if (has_b) issue(posb, 0, curb, finb);
            issue(pos, 1, cur, fin);                         // both streams in flight together
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (has_b) {
            const double* lv = bufv(0); const int* lc = bufc(0);
#pragma unroll
            for (int j = 0; j < 5; ++j) { const bool has = curb + j < finb; const int idx = has ? curb + j : 0; const unsigned cc = (unsigned)lc[idx] << 3; const double vv = lv[idx]; ob[j] = has ? cc : 0xffffffffu; vb[j] = has ? vv : 0.0; }
        }
        if (HALF) { __syncthreads(); issue(pos, 1, cur, fin); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        const double* lv = bufv(1); const int* lc = bufc(1);
        for (int k = 0; k < K; ++k) {
            const int c1 = (k + 1 < K) ? (k + 1) * W : 0x7fffffff;
            const unsigned o_lo = (unsigned)(k * W) << 3, o_hi = (k + 1 < K) ? (unsigned)c1 << 3 : 0xffffffffu;
            double xb[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) { xb[j] = 0.0; if (ob[j] >= o_lo && ob[j] < o_hi) { const u2 w = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(xres, (int)ob[j], 0, 0)); xb[j] = __builtin_bit_cast(double, w); } }
            for (;;) {
                int ca = 0x7fffffff;
                if (cur < fin) ca = lc[cur];
                const bool oa = ca < c1;
                if (oa) { const u2 w = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(xres, ca << 3, 0, 0)); sum += lv[cur] * __builtin_bit_cast(double, w); cur += 1; }
                if (!__any(oa)) break;
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) { const bool in = ob[j] >= o_lo && ob[j] < o_hi; const double t = sumb + vb[j] * xb[j]; sumb = in ? t : sumb; }
        }
        { const long r = (long)pos * ROWS + threadIdx.x; if (r < n) y[r] = sum; }
        if (has_b) { const long r = (long)posb * ROWS + threadIdx.x; if (r < n) y[r] = sumb; }
        __syncthreads();
    }
}

// gath : the raw cost of the gathers alone: out[lane] = sum of x[idx[j]] over a grid-stride range (no values, no rows)
template <int U>
__global__ __launch_bounds__(BLOCK) void gath(long nnz, const int* __restrict__ idx, const double* __restrict__ x, double* __restrict__ out) {
    const long S = (long)gridDim.x * BLOCK;
    double s = 0.0;
    long j = (long)blockIdx.x * BLOCK + threadIdx.x;
    for (; j + (U - 1) * S < nnz; j += U * S) {
        int c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = idx[j + u * S];
#pragma unroll
        for (int u = 0; u < U; ++u) s += x[c[u]];
    }
    for (; j < nnz; j += S) s += x[idx[j]];
    out[(long)blockIdx.x * BLOCK + threadIdx.x] = s;
}

struct Csr { int n; std::vector<int> ip, ix; std::vector<double> dv; };

static Csr make_random(int n, int k, uint64_t seed) {
    Csr A; A.n = n; A.ip.assign(n + 1, 0);
    uint64_t s = seed * 6364136223846793005ULL + 1442695040888963407ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    std::vector<std::pair<int, double>> row;
    for (int r = 0; r < n; ++r) {
        row.clear();
        double dsum = 1.0;
        for (int q = 0; q < k; ++q) {
            int c = (int)(rnd() % (uint64_t)n);
            double v = ((double)(rnd() % 2000001) - 1000000.0) / 500000.0;
            if (c == r || v == 0.0) continue;
            bool dup = false;
            for (auto& e : row) if (e.first == c) { e.second += v; dup = true; }
            if (!dup) row.push_back({c, v});
        }
        for (auto& e : row) dsum += e.second < 0 ? -e.second : e.second;
        row.push_back({r, dsum});
        std::sort(row.begin(), row.end());
        for (auto& e : row) { A.ix.push_back(e.first); A.dv.push_back(e.second); }
        A.ip[r + 1] = (int)A.ix.size();
    }
    return A;
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    Csr A = make_random(n, 4, 7);
    const long nnz = (long)A.ix.size();
    const int T = (n + ROWS - 1) / ROWS;
    std::vector<double> x(n), yref(n);
    for (int i = 0; i < n; ++i) x[i] = 1.0 + (double)(i % 977) / 977.0;
    for (int r = 0; r < n; ++r) { double s = 0; for (int j = A.ip[r]; j < A.ip[r + 1]; ++j) s += A.dv[j] * x[A.ix[j]]; yref[r] = s; }
    const double alg = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    printf("n = %d nnz = %ld tiles = %d algorithmic bytes = %.1f MB\n", n, nnz, T, alg / 1e6);
    double *dx, *dy; CK(hipMalloc(&dx, 8L * n)); CK(hipMalloc(&dy, 8L * n));
    CK(hipMemcpy(dx, x.data(), 8L * n, hipMemcpyHostToDevice));
    int *dip, *dix; double* ddv;
    CK(hipMalloc(&dip, 4L * (n + 1))); CK(hipMalloc(&dix, 4 * nnz)); CK(hipMalloc(&ddv, 8 * nnz));
    CK(hipMemcpy(dip, A.ip.data(), 4L * (n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(dix, A.ix.data(), 4 * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddv, A.dv.data(), 8 * nnz, hipMemcpyHostToDevice));
    {
        double* dout; CK(hipMalloc(&dout, 8L * 4096 * BLOCK));
        for (int G : {1024, 2048, 4096}) {
            float m1 = timeit([&] { hipLaunchKernelGGL((gath<1>), dim3(G), dim3(BLOCK), 0, 0, nnz, dix, dx, dout); }, reps);
            float m5 = timeit([&] { hipLaunchKernelGGL((gath<5>), dim3(G), dim3(BLOCK), 0, 0, nnz, dix, dx, dout); }, reps);
            printf("gather only grid=%4d : U=1 %6.1f us  U=5 %6.1f us  (%ld gathers of 8 B from %.1f MB)\n", G, m1 * 1e3, m5 * 1e3, nnz, 8.0 * n / 1e6);
        }
        CK(hipFree(dout));
    }
    std::vector<int> Ks;
    for (int a = 3; a < argc; ++a) Ks.push_back(atoi(argv[a]));
    if (Ks.empty()) Ks = {1, 2, 4, 8};
    for (int K : Ks) {
        const long W = ((long)n + K - 1) / K;
        std::vector<int> seg((size_t)T * K + 1), cols(nnz);
        std::vector<double> vals(nnz);
        std::vector<uint16_t> ro((size_t)T * K * (ROWS + 1));
        long p = 0; int maxlen = 0;
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < K; ++k) {
                const size_t s = (size_t)t * K + k;
                seg[s] = (int)p;
                const long c0 = k * W, c1 = std::min<long>(n, c0 + W);
                for (int q = 0; q <= ROWS; ++q) {
                    ro[s * (ROWS + 1) + q] = (uint16_t)(p - seg[s]);
                    const long r = (long)t * ROWS + q;
                    if (q == ROWS || r >= n) continue;
                    for (int j = A.ip[r]; j < A.ip[r + 1]; ++j)
                        if (A.ix[j] >= c0 && A.ix[j] < c1) { cols[p] = A.ix[j]; vals[p] = A.dv[j]; ++p; }
                }
                maxlen = std::max(maxlen, (int)(p - seg[s]));
            }
        seg[(size_t)T * K] = (int)p;
        if (p != nnz || maxlen > PCAP) { printf("format error %ld %ld %d\n", p, nnz, maxlen); return 1; }
        int *dseg, *dcols; double* dvals; uint16_t* dro;
        CK(hipMalloc(&dseg, 4 * seg.size())); CK(hipMalloc(&dcols, 4 * nnz)); CK(hipMalloc(&dvals, 8 * nnz)); CK(hipMalloc(&dro, 2 * ro.size()));
        CK(hipMemcpy(dseg, seg.data(), 4 * seg.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dcols, cols.data(), 4 * nnz, hipMemcpyHostToDevice));
        CK(hipMemcpy(dvals, vals.data(), 8 * nnz, hipMemcpyHostToDevice));
        CK(hipMemcpy(dro, ro.data(), 2 * ro.size(), hipMemcpyHostToDevice));
        auto check = [&](const char* nm) {
            std::vector<double> y(n); CK(hipMemcpy(y.data(), dy, 8L * n, hipMemcpyDeviceToHost));
            long bad = 0; for (int i = 0; i < n; ++i) bad += (y[i] != yref[i]);
            if (bad) printf("   !! %s: %ld rows differ\n", nm, bad);
        };
#define RUN(KERN, TW, G, NAME) { CK(hipMemset(dy, 0, 8L * n)); float ms = timeit([&] { hipLaunchKernelGGL((KERN<TW>), dim3(G), dim3(BLOCK), 0, 0, T, K, n, dseg, dro, dcols, dvals, dx, dy); }, reps); \
            CK(hipGetLastError()); check(NAME); printf("K=%2d %-5s grid=%4d tiles/wg=%d : %7.1f us  %5.2f TB/s (frac %.3f)  maxseg=%d\n", K, NAME, G, TW, ms * 1e3, alg / ms / 1e9, alg / ms / 1e9 / 8.0, maxlen); }
        const int g1 = T, g2 = (T + 1) / 2, g4 = (T + 3) / 4;
#define RUNRES(TW, U, G) { CK(hipMemset(dy, 0, 8L * n)); const size_t lds = (size_t)TW * RCAP * 12; \
            CK(hipFuncSetAttribute((const void*)cb_res<TW, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            float ms = timeit([&] { hipLaunchKernelGGL((cb_res<TW, U>), dim3(G), dim3(BLOCK), lds, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy); }, reps); \
            CK(hipGetLastError()); check("res"); printf("K=%2d res%d  grid=%4d tiles/wg=%d : %7.1f us  %5.2f TB/s (frac %.3f)\n", K, U, G, TW, ms * 1e3, alg / ms / 1e9, alg / ms / 1e9 / 8.0); }
#define RUNREG(RT, OCC, G) { CK(hipMemset(dy, 0, 8L * n)); \
            float ms = timeit([&] { hipLaunchKernelGGL((cb_reg<RT, 5, OCC>), dim3(G), dim3(BLOCK), 0, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, (long long*)nullptr); }, reps); \
            CK(hipGetLastError()); check("reg"); printf("K=%2d reg   grid=%4d tiles/wg=%d occ=%d : %7.1f us  %5.2f TB/s (frac %.3f)\n", K, G, RT, OCC, ms * 1e3, alg / ms / 1e9, alg / ms / 1e9 / 8.0); }
        if (getenv("REG_TS")) {   // per-workgroup wall-clock stamps (100 MHz) at kernel start, at every phase start and at the end
            const int G = g2; std::vector<long long> h((size_t)(K + 2) * G); long long* dts; CK(hipMalloc(&dts, 8 * h.size()));
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((cb_reg<2, 5, 8>), dim3(G), dim3(BLOCK), 0, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, dts);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), dts, 8 * h.size(), hipMemcpyDeviceToHost));
            long long t0 = h[0]; for (int b = 0; b < G; ++b) t0 = std::min(t0, h[b]);
            for (int k = 0; k < K + 2; ++k) { long long lo = 1LL << 62, hi = 0; double av = 0; for (int b = 0; b < G; ++b) { long long v = h[(size_t)k * G + b] - t0; lo = std::min(lo, v); hi = std::max(hi, v); av += v; }
                printf("  stamp %d: min %.2f us  mean %.2f us  max %.2f us\n", k, lo / 100.0, av / G / 100.0, hi / 100.0); }
            CK(hipFree(dts)); }
#define RUNREGL(RT, OCC, G) { CK(hipMemset(dy, 0, 8L * n)); const size_t lds = (size_t)(RT + 1) * RCAP * 4; \
            CK(hipFuncSetAttribute((const void*)cb_reg<RT, 5, OCC, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            float ms = timeit([&] { hipLaunchKernelGGL((cb_reg<RT, 5, OCC, 1>), dim3(G), dim3(BLOCK), lds, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, (long long*)nullptr); }, reps); \
            CK(hipGetLastError()); check("regL"); printf("K=%2d regL  grid=%4d tiles/wg=%d occ=%d : %7.1f us  %5.2f TB/s (frac %.3f)\n", K, G, RT, OCC, ms * 1e3, alg / ms / 1e9, alg / ms / 1e9 / 8.0); }
        if (getenv("REG_TS")) {
            const int G = g2; std::vector<long long> h((size_t)(K + 2) * G); long long* dts; CK(hipMalloc(&dts, 8 * h.size()));
            const size_t lds = (size_t)3 * RCAP * 4;
            CK(hipFuncSetAttribute((const void*)cb_reg<2, 5, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((cb_reg<2, 5, 8, 1>), dim3(G), dim3(BLOCK), lds, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, dts);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), dts, 8 * h.size(), hipMemcpyDeviceToHost));
            long long t0 = h[0]; for (int b = 0; b < G; ++b) t0 = std::min(t0, h[b]);
            for (int k = 0; k < K + 2; ++k) { long long lo = 1LL << 62, hi = 0; double av = 0; for (int b = 0; b < G; ++b) { long long v = h[(size_t)k * G + b] - t0; lo = std::min(lo, v); hi = std::max(hi, v); av += v; }
                printf("  regL stamp %d: min %.2f us  mean %.2f us  max %.2f us\n", k, lo / 100.0, av / G / 100.0, hi / 100.0); }
            CK(hipFree(dts)); }
#define RUNREGS(RT, OCC, G0, MODE) { const int G = (G0 + 7) / 8 * 8; CK(hipMemset(dy, 0, 8L * n)); const size_t lds = (size_t)(RT + 1) * RCAP * 4; \
            CK(hipFuncSetAttribute((const void*)cb_reg<RT, 5, OCC, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            unsigned* dsb; CK(hipMalloc(&dsb, 4 * 32 * 8 * 64)); CK(hipMemset(dsb, 0, 4 * 32 * 8 * 64)); unsigned ep = 0; \
            float ms = timeit([&] { ++ep; hipLaunchKernelGGL((cb_reg<RT, 5, OCC, 1>), dim3(G), dim3(BLOCK), lds, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, (long long*)nullptr, dsb, ep, MODE); }, reps); \
            CK(hipGetLastError()); check("regS"); printf("K=%2d regS%d grid=%4d tiles/wg=%d occ=%d : %7.1f us  %5.2f TB/s (frac %.3f)\n", K, MODE, G, RT, OCC, ms * 1e3, alg / ms / 1e9, alg / ms / 1e9 / 8.0); CK(hipFree(dsb)); }
        if (getenv("PAIR_AUX")) {   // cache policy of the gathers: aux bits of the buffer load (1 sc0, 2 nt, 16 sc1)
            const int G = 2048; const size_t lds = (size_t)RCAP * 12;
#define RUNAUX(P, AX) { CK(hipMemset(dy, 0, 8L * n)); float ms = timeit([&] { hipLaunchKernelGGL((cb_pair<P, AX>), dim3(G), dim3(BLOCK), lds, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, (long long*)nullptr); }, reps); \
                check("aux"); printf("K=%2d pair=%d aux=%2d : %7.1f us\n", K, P, AX, ms * 1e3); }
            RUNAUX(0, 0) RUNAUX(0, 1) RUNAUX(0, 2) RUNAUX(0, 3) RUNAUX(0, 16) RUNAUX(0, 17) RUNAUX(0, 18)
            RUNAUX(1, 0) RUNAUX(1, 1) RUNAUX(1, 2) RUNAUX(1, 3) RUNAUX(1, 16) RUNAUX(1, 17) RUNAUX(1, 18)
        }
        if (getenv("PAIR")) for (int pair = 0; pair <= 1; ++pair) {
            const int G = 2048; const size_t lds = (size_t)RCAP * 12; const int NS = K + 3;
            std::vector<long long> h((size_t)16 * G); long long* dts; CK(hipMalloc(&dts, 8 * h.size())); CK(hipMemset(dts, 0, 8 * h.size()));
            auto launch = [&](long long* t) { if (pair) hipLaunchKernelGGL((cb_pair<1>), dim3(G), dim3(BLOCK), lds, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, t);
                                              else hipLaunchKernelGGL((cb_pair<0>), dim3(G), dim3(BLOCK), lds, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, t); };
            CK(hipMemset(dy, 0, 8L * n));
            float ms = timeit([&] { launch(nullptr); }, reps);
            check("pair");
            printf("K=%2d pair=%d grid=%4d : %7.1f us\n", K, pair, G, ms * 1e3);
            for (int rep = 0; rep < 3; ++rep) launch(dts);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), dts, 8 * h.size(), hipMemcpyDeviceToHost));
            long long t0 = h[0]; for (int b = 0; b < G; ++b) t0 = std::min(t0, h[b]);
            for (int k = 0; k < NS; ++k) { long long lo = 1LL << 62, hi = 0; double av = 0; for (int b = 0; b < G; ++b) { long long v = h[(size_t)k * G + b] - t0; lo = std::min(lo, v); hi = std::max(hi, v); av += v; }
                printf("  pair=%d stamp %d: min %.2f us  mean %.2f us  max %.2f us\n", pair, k, lo / 100.0, av / G / 100.0, hi / 100.0); }
            CK(hipFree(dts)); }
        if (getenv("PAIR_OVL")) {                          // item 3 priced: overlapped ingests at twice the LDS / half the occupancy
            const size_t lds2 = (size_t)RCAP * 24;
            for (int G : {2048, 1024}) {
                CK(hipMemset(dy, 0, 8L * n));
                float m0 = timeit([&] { hipLaunchKernelGGL((cb_pair2<0>), dim3(G), dim3(BLOCK), lds2, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy); }, reps);
                check("pair2 overlapped");
                CK(hipMemset(dy, 0, 8L * n));
                float m1 = timeit([&] { hipLaunchKernelGGL((cb_pair2<1>), dim3(G), dim3(BLOCK), lds2, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy); }, reps);
                check("pair2 serial");
                printf("K=%2d pair2 grid=%4d (LDS %zu B per workgroup: 4 per CU): ingests overlapped %7.1f us, ingests in series %7.1f us\n", K, G, lds2, m0 * 1e3, m1 * 1e3);
            }
        }
        if (getenv("PAIR_ONLY")) { CK(hipFree(dseg)); CK(hipFree(dcols)); CK(hipFree(dvals)); CK(hipFree(dro)); continue; }
        if (getenv("REG_TS")) for (int mode = 1; mode <= 2; ++mode) {
            const int G = (g2 + 7) / 8 * 8; std::vector<long long> h((size_t)(K + 2) * G); long long* dts; CK(hipMalloc(&dts, 8 * h.size()));
            const size_t lds = (size_t)3 * RCAP * 4;
            unsigned* dsb; CK(hipMalloc(&dsb, 4 * 32 * 8 * 64)); CK(hipMemset(dsb, 0, 4 * 32 * 8 * 64));
            for (unsigned rep = 1; rep <= 3; ++rep) hipLaunchKernelGGL((cb_reg<2, 5, 8, 1>), dim3(G), dim3(BLOCK), lds, 0, T, K, n, (int)W, dip, dix, ddv, dx, dy, dts, dsb, rep, mode);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), dts, 8 * h.size(), hipMemcpyDeviceToHost));
            long long t0 = h[0]; for (int b = 0; b < G; ++b) t0 = std::min(t0, h[b]);
            for (int k = 0; k < K + 2; ++k) { long long lo = 1LL << 62, hi = 0; double av = 0; for (int b = 0; b < G; ++b) { long long v = h[(size_t)k * G + b] - t0; lo = std::min(lo, v); hi = std::max(hi, v); av += v; }
                printf("  regS%d stamp %d: min %.2f us  mean %.2f us  max %.2f us\n", mode, k, lo / 100.0, av / G / 100.0, hi / 100.0); }
            CK(hipFree(dts)); CK(hipFree(dsb)); }
        if (getenv("REG_ONLY") || getenv("REG")) { RUNREGS(2, 8, g2, 1) RUNREGS(2, 8, g2, 2) RUNREGS(4, 4, g4, 1) RUNREGS(4, 4, g4, 2) RUNREGS(1, 8, 2048, 1) RUNREGS(1, 8, 2048, 2)
        RUNREGL(1, 8, g1) RUNREGL(2, 8, g2) RUNREGL(4, 4, g4) RUNREG(1, 8, g1) RUNREG(2, 8, g2) RUNREG(2, 4, g2) RUNREG(4, 4, g4) RUNREG(4, 2, g4) }
        if (getenv("REG_ONLY")) { CK(hipFree(dseg)); CK(hipFree(dcols)); CK(hipFree(dvals)); CK(hipFree(dro)); continue; }
        RUNRES(1, 1, g1) RUNRES(1, 2, g1) RUNRES(2, 1, g2) RUNRES(2, 2, g2) RUNRES(4, 1, g4) RUNRES(4, 2, g4)
        if (getenv("RES_ONLY")) { CK(hipFree(dseg)); CK(hipFree(dcols)); CK(hipFree(dvals)); CK(hipFree(dro)); continue; }
        RUN(cb_lane, 1, g1, "lane") RUN(cb_lane, 2, g2, "lane") RUN(cb_lane, 4, g4, "lane")
        RUN(cb_lds, 1, g1, "lds") RUN(cb_lds, 2, g2, "lds") RUN(cb_lds, 4, g4, "lds")
        CK(hipFree(dseg)); CK(hipFree(dro));
        {   // sub-tile (64 rows) format for the wave variant
            const int S = (n + 63) / 64;
            std::vector<int> seg2((size_t)S * K + 1);
            std::vector<uint16_t> ro2((size_t)S * K * 65);
            long p2 = 0; int maxl = 0;
            for (int st = 0; st < S; ++st)
                for (int k = 0; k < K; ++k) {
                    const size_t s2 = (size_t)st * K + k;
                    seg2[s2] = (int)p2;
                    const long c0 = k * W, c1 = std::min<long>(n, c0 + W);
                    for (int q = 0; q <= 64; ++q) {
                        ro2[s2 * 65 + q] = (uint16_t)(p2 - seg2[s2]);
                        const long r = (long)st * 64 + q;
                        if (q == 64 || r >= n) continue;
                        for (int j = A.ip[r]; j < A.ip[r + 1]; ++j)
                            if (A.ix[j] >= c0 && A.ix[j] < c1) { cols[p2] = A.ix[j]; vals[p2] = A.dv[j]; ++p2; }
                    }
                    maxl = std::max(maxl, (int)(p2 - seg2[s2]));
                }
            seg2[(size_t)S * K] = (int)p2;
            if (p2 != nnz || maxl > WCAP) { printf("format error %ld %ld %d\n", p2, nnz, maxl); return 1; }
            CK(hipMalloc(&dseg, 4 * seg2.size())); CK(hipMalloc(&dro, 2 * ro2.size()));
            CK(hipMemcpy(dseg, seg2.data(), 4 * seg2.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dro, ro2.data(), 2 * ro2.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dcols, cols.data(), 4 * nnz, hipMemcpyHostToDevice));
            CK(hipMemcpy(dvals, vals.data(), 8 * nnz, hipMemcpyHostToDevice));
            const int T = S; const int maxlen = maxl;    // (RUN prints these)
            const int w1 = (S + 3) / 4, w2 = (S + 7) / 8, w4 = (S + 15) / 16;
            RUN(cb_wave, 1, w1, "wave") RUN(cb_wave, 2, w2, "wave") RUN(cb_wave, 4, w4, "wave")
        }
        CK(hipFree(dseg)); CK(hipFree(dcols)); CK(hipFree(dvals)); CK(hipFree(dro));
    }
    return 0;
}
