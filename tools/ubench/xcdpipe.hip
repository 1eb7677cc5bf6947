// xcdpipe -- micro-benchmark (round 6): XCD-owned column blocks INSIDE ONE LAUNCH for the scattered
// product whose x is several L2s long -- the least-squares loops' A' u: A' is 1e6 x 4e6 with ~20 entries per row, u = 32 MB =
// eight 4 MiB slices, one per XCD's L2.
//
// Today (csrc/mk_format.hip cblocks_build): four column-block LAUNCHES, every block a resident tile of format 3 whose
// column phases walk the block's 8 MiB slice; 175 us for 284 MB, 0.88 GB through the fabric (3.1 x), against a gather floor
// of 94 us (2.9 clocks per gathered entry and CU).
//
// Here: workgroup b works for XCD b % 8 (observed dispatch order) on column block k = b % 8 ONLY, for the row tiles
// w, w + W, ... (w = b / 8).  A row's sum must run left to right over the column blocks, so stage k continues where stage
// k - 1 left off: the partial sums of a tile travel through device memory (uncached allocation: no L2 invalidate is needed on
// the reader's side, which would throw the u slice out) and a per-tile flag word says how far a tile has come.  A consumer's
// producer always has a lower workgroup number (same w, stage k - 1), workgroups are dispatched in order, every workgroup
// walks its tiles in ascending order and stage 0 never waits: no deadlock whatever the residency.
//
//   hipcc --offload-arch=gfx950 -O3 -o xcdpipe xcdpipe.hip && ./xcdpipe [rows=1000000] [cols=4000000] [per_col_row=5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <random>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256, ROWS = 256, K = 8;

// segment (tile t, block k): entries seg[t * K + k] .. seg[t * K + k + 1] of cols / vals, row by row; rowoff[(t * K + k) * 257 + i]
// = start of row i inside the segment
// MODE 0: release = __threadfence() (agent-scope: writes the L2's dirty lines back); 1: the sums are write-through (uncached
// memory): wait for the stores' acknowledgements, then raise the flag; 2: no synchronisation at all (WRONG results: the cost of
// the layout's loads and gathers alone)
template <int LDSCAP, bool NT, int MODE>
__global__ __launch_bounds__(BLOCK, 8) void pipe_kernel(int T, int n, const int *__restrict__ seg, const uint16_t *__restrict__ rowoff,
                                                       const int *__restrict__ cols, const double *__restrict__ vals,
                                                       const double *__restrict__ u, double *__restrict__ y,
                                                       double *psum, int *flag, int epoch, int *err) {
    __shared__ double lv[LDSCAP];
    __shared__ int lc[LDSCAP];
    const int k = blockIdx.x % K, w = blockIdx.x / K, W = gridDim.x / K;
    const int tid = threadIdx.x;
    for (int t = w; t < T; t += W) {
        const int s = t * K + k, base = seg[s], len = seg[s + 1] - base;
        const uint16_t *ro = rowoff + (size_t)s * (ROWS + 1);
        const int lo = ro[tid], hi = ro[tid + 1];
        for (int j = tid; j < len; j += BLOCK) {              // the segment's stream, coalesced, past the caches
            if constexpr (NT) {
                lc[j] = __builtin_nontemporal_load(cols + base + j);
                lv[j] = __builtin_nontemporal_load(vals + base + j);
            } else {
                lc[j] = cols[base + j];
                lv[j] = vals[base + j];
            }
        }
        double xg[8];
        double sum = 0.0;
        if (k > 0 && MODE != 2) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(flag + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * K + k) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 22)) {                // (a bug must not hang the box)
                        *err = 1;
                        break;
                    }
                }
            }
        }
        __syncthreads();                                      // the segment is in LDS, the predecessor's sums are out
        if (k > 0) sum = __builtin_nontemporal_load(psum + (size_t)t * ROWS + tid);
#pragma unroll
        for (int j = 0; j < 8; ++j) {                         // all gathers of the row's run in flight together
            const int p = lo + j;
            xg[j] = (p < hi) ? u[lc[p]] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = lo + j;
            if (p < hi) sum += lv[p] * xg[j];
        }
        for (int p = lo + 8; p < hi; ++p) sum += lv[p] * u[lc[p]];
        const long r = (long)t * ROWS + tid;
        if (k + 1 < K) {
            __builtin_nontemporal_store(sum, psum + (size_t)t * ROWS + tid);
            if constexpr (MODE == 0) __threadfence();         // (release: the sums are out before the flag says so)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag + t, epoch * K + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (r < n) y[r] = sum;
            __syncthreads();
        }
    }
}

// reference structure on the device: plain CSR, one row per lane (what the gather path does, for the bits)
__global__ void csr_rows(int n, const int *__restrict__ ip, const int *__restrict__ ix, const double *__restrict__ v,
                         const double *__restrict__ u, double *__restrict__ y) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s = 0.0;
    for (int j = ip[r]; j < ip[r + 1]; ++j) s += v[j] * u[ix[j]];
    y[r] = s;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000;         // rows of A' (columns of A)
    const int m = argc > 2 ? atoi(argv[2]) : 4000000;         // columns of A' (rows of A)
    const int kk = argc > 3 ? atoi(argv[3]) : 5;              // entries per row of A
    // A: m x n, entry j of a row in the j-th of kk equal column ranges (bench.py random_tall_csr); A' by counting sort
    std::mt19937_64 rng(11);
    const int wcol = n / kk;
    std::vector<int> acol((size_t)m * kk);
    std::vector<double> aval((size_t)m * kk);
    std::normal_distribution<double> nd;
    for (size_t i = 0; i < acol.size(); ++i) {
        acol[i] = (int)(rng() % wcol) + (int)(i % kk) * wcol;
        aval[i] = nd(rng);
    }
    std::vector<int> ip(n + 1, 0);
    for (int c : acol) ip[c + 1]++;
    for (int i = 0; i < n; ++i) ip[i + 1] += ip[i];
    const int nnz = ip[n];
    std::vector<int> ix(nnz), fill(ip.begin(), ip.end() - 1);
    std::vector<double> vv(nnz);
    for (int r = 0; r < m; ++r)
        for (int j = 0; j < kk; ++j) {
            const size_t e = (size_t)r * kk + j;
            const int p = fill[acol[e]]++;
            ix[p] = r;                                        // (rows of A ascend: the columns of a row of A' are sorted)
            vv[p] = aval[e];
        }
    // column-block segments
    const int T = (n + ROWS - 1) / ROWS;
    const int cw = (m + K - 1) / K;
    std::vector<int> seg((size_t)T * K + 1, 0);
    std::vector<uint16_t> rowoff((size_t)T * K * (ROWS + 1), 0);
    std::vector<int> scol(nnz);
    std::vector<double> sval(nnz);
    int pos = 0, maxseg = 0, maxrun = 0;
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < K; ++k) {
            const size_t s = (size_t)t * K + k;
            seg[s] = pos;
            for (int i = 0; i < ROWS; ++i) {
                const int r = t * ROWS + i;
                rowoff[s * (ROWS + 1) + i] = (uint16_t)(pos - seg[s]);
                if (r < n) {
                    int run = 0;
                    for (int j = ip[r]; j < ip[r + 1]; ++j)
                        if (ix[j] / cw == k) {
                            scol[pos] = ix[j];
                            sval[pos] = vv[j];
                            ++pos;
                            ++run;
                        }
                    maxrun = std::max(maxrun, run);
                }
            }
            rowoff[s * (ROWS + 1) + ROWS] = (uint16_t)(pos - seg[s]);
            maxseg = std::max(maxseg, pos - seg[s]);
        }
    seg[(size_t)T * K] = pos;
    printf("A' %d x %d, %d entries (%.1f per row), %d tiles x %d column blocks of %d columns (%.1f MB of u each); longest segment %d, "
           "longest run %d\n", n, m, nnz, (double)nnz / n, T, K, cw, cw * 8e-6, maxseg, maxrun);
    if (maxseg > 1536) { printf("segment too long for the LDS buffer\n"); return 1; }
    std::vector<double> hu(m);
    for (auto &x : hu) x = nd(rng);
    int *d_ip, *d_ix, *d_seg, *d_scol, *d_flag, *d_err;
    uint16_t *d_ro;
    double *d_v, *d_sval, *d_u, *d_y, *d_yref, *d_psum;
    CK(hipMalloc(&d_ip, sizeof(int) * (n + 1)));
    CK(hipMalloc(&d_ix, sizeof(int) * nnz));
    CK(hipMalloc(&d_v, sizeof(double) * nnz));
    CK(hipMalloc(&d_seg, sizeof(int) * seg.size()));
    CK(hipMalloc(&d_ro, sizeof(uint16_t) * rowoff.size()));
    CK(hipMalloc(&d_scol, sizeof(int) * nnz));
    CK(hipMalloc(&d_sval, sizeof(double) * nnz));
    CK(hipMalloc(&d_u, sizeof(double) * m));
    CK(hipMalloc(&d_y, sizeof(double) * (size_t)T * ROWS));
    CK(hipMalloc(&d_yref, sizeof(double) * (size_t)T * ROWS));
    CK(hipMalloc(&d_err, sizeof(int)));
    // partial sums and flags: uncached device memory (coherent across the XCDs' L2s without invalidates)
    const char *plain = getenv("XP_PLAIN_MEM");
    if (plain) {
        CK(hipMalloc(&d_psum, sizeof(double) * (size_t)T * ROWS));
        CK(hipMalloc(&d_flag, sizeof(int) * T));
    } else {
        CK(hipExtMallocWithFlags((void **)&d_psum, sizeof(double) * (size_t)T * ROWS, hipDeviceMallocUncached));
        CK(hipExtMallocWithFlags((void **)&d_flag, sizeof(int) * T, hipDeviceMallocUncached));
    }
    CK(hipMemset(d_flag, 0, sizeof(int) * T));
    CK(hipMemset(d_err, 0, sizeof(int)));
    CK(hipMemcpy(d_ip, ip.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ix, ix.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_v, vv.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_seg, seg.data(), sizeof(int) * seg.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ro, rowoff.data(), sizeof(uint16_t) * rowoff.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_scol, scol.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sval, sval.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_u, hu.data(), sizeof(double) * m, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(csr_rows, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_ip, d_ix, d_v, d_u, d_yref);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int epoch = 0;
    std::vector<double> ya((size_t)T * ROWS), yb((size_t)T * ROWS);
    CK(hipMemcpy(yb.data(), d_yref, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (int mode : {2, 1, 0})
        for (int per_cu : {2, 4, 8})
            for (int nt = 0; nt < 2; ++nt) {
                const int grid = 256 * per_cu;                    // W = 32 per_cu workgroups per XCD
                float best = 1e30f, sum_ms = 0.f;
                const int reps = 12;
                for (int it = 0; it < reps + 2; ++it) {
                    ++epoch;
                    CK(hipEventRecord(e0));
#define LAUNCH(NTV, MV) hipLaunchKernelGGL((pipe_kernel<1536, NTV, MV>), dim3(grid), dim3(BLOCK), 0, 0, T, n, d_seg, d_ro, d_scol, d_sval, d_u, d_y, d_psum, d_flag, epoch, d_err)
                    if (mode == 0) { if (nt) LAUNCH(true, 0); else LAUNCH(false, 0); }
                    else if (mode == 1) { if (nt) LAUNCH(true, 1); else LAUNCH(false, 1); }
                    else { if (nt) LAUNCH(true, 2); else LAUNCH(false, 2); }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (it >= 2) {
                        best = std::min(best, ms);
                        sum_ms += ms;
                    }
                }
                int herr = 0;
                CK(hipMemcpy(&herr, d_err, sizeof(int), hipMemcpyDeviceToHost));
                CK(hipMemcpy(ya.data(), d_y, sizeof(double) * n, hipMemcpyDeviceToHost));
                const bool same = memcmp(ya.data(), yb.data(), sizeof(double) * n) == 0;
                printf("%-26s %4d workgroups (%d per CU), %s stream loads: avg %7.1f us  best %7.1f us   bits %s%s\n",
                       mode == 0 ? "pipeline, __threadfence" : (mode == 1 ? "pipeline, waitcnt release" : "NO SYNC (layout cost only)"), grid, per_cu,
                       nt ? "non-temporal" : "plain       ", 1e3 * sum_ms / reps, 1e3 * best,
                       mode == 2 ? "(not compared)" : (same ? "equal to the CSR loop" : "DIFFER"), herr ? "   SPIN LIMIT HIT" : "");
                if (herr) return 2;
            }
    return 0;
}
