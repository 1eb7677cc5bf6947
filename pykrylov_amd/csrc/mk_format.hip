// mk_format.hip -- the windowed tile format of a CSR matrix (built on first use, owned by the matrix).
//
// The reference has no sparse format at all (the operator wraps a user callable, linop/linop.py:114,289); CSR is
// this library's own layout and the format below is an acceleration structure BESIDE the CSR arrays, in the spirit
// of north_star's "LDS staging of x-vector tiles": the SpMV kernel (mk_device.h) reads it instead of `indices`
// (and, with a value dictionary, instead of `data`) for every tile the builder could cover.
//
//  * cover:  per 256-row tile the sorted set of referenced column PAIRS is cut into windows wherever two
//    neighbours are more than G pairs apart (G = 4, 16, ... until the cover fits); a window is stored as chunks
//    of 128 doubles, chunk c of a tile belonging to wave c % 4.  A tile is eligible when it has 1..2048 nonzeros
//    (counted from the 8-aligned start of its stream) and its cover needs <= 16 chunks.  Per nonzero the builder
//    stores the uint16 position of its x entry in the tile's LDS window buffer.  Matrices this cover does not
//    reach (rows of more than 8 entries) get the WIDE cover -- 32 chunks, 8192 nonzeros per tile -- and one of the
//    row-walk formats 6 / 7 / 8 (slots + values, row patterns + values, row patterns + dictionary; plan_build).
//  * value dictionary: when the whole matrix holds <= 256 distinct values (bit patterns), `data` is replaced
//    by an 8-bit index into the sorted dictionary, packed with the slot into ONE 32-bit word per nonzero --
//    constant-coefficient stencils, graph Laplacians,
//    incidence matrices.  Lossless: the product multiplies exactly the same doubles.
//
// Everything is integer work on the device; the row sums are still formed left to right from the same products,
// so the results do not change by a bit whichever format a tile is in.
#include "mk_device.h"

namespace {

constexpr unsigned long long DICT_EMPTY = 0x7ff8dead00c0ffeeULL;
constexpr int DICT_SLOTS = 512;

__device__ inline void bitonic_sort_i32(int *a, int n2) {
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += MK_BLOCK) {
                const int q = i ^ j;
                if (q > i) {
                    const int u = a[i], v = a[q];
                    const bool up = ((i & k) == 0);
                    if ((u > v) == up) {
                        a[i] = v;
                        a[q] = u;
                    }
                }
            }
            __syncthreads();
        }
}

// stats: [0] max chunks of a tile, [1] eligible tiles
// <16, 2048>: the cover of fmt 1, 2, 4, 5 (tiles the per-nonzero kernels can stage); <32, 8192>: the wide cover of
// fmt 6, 7, 8 (row-walk kernels only: no staging buffer bounds the tile).  Chunk c of a tile belongs to wave c % 4:
// wg[(tile * 4 + wave) * (CMAX / 4) + c / 4], half lengths one byte each in wn[(tile * 4 + wave) * (CMAX / 16) + c / 16].
template <int CMAX, int COVER_CAP>
__global__ __launch_bounds__(MK_BLOCK) void cover_kernel(const int32_t *__restrict__ ip, const int32_t *__restrict__ ix,
                                                         int64_t nrows, int64_t ntiles, int64_t xlen,
                                                         int32_t *__restrict__ wg, uint32_t *__restrict__ wn,
                                                         uint16_t *__restrict__ sl, int *__restrict__ stats) {
    constexpr int CPW = CMAX / 4;
    __shared__ int key[COVER_CAP];
    __shared__ int heads[CMAX + 1];
    __shared__ int wst[CMAX], wof[CMAX];
    __shared__ int nh, s_ok, s_nw;
    const int tid = threadIdx.x;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * MK_ROWS_PER_TILE;
        const int64_t rend = (r0 + MK_ROWS_PER_TILE < nrows) ? r0 + MK_ROWS_PER_TILE : nrows;
        const int p_lo = ip[r0], p_hi = ip[rend], cnt = p_hi - p_lo;
        __syncthreads();                                     // previous tile's shared state is no longer read
        if (tid < CMAX) wg[tile * CMAX + tid] = 0;
        if (tid < CMAX / 4) wn[tile * (CMAX / 4) + tid] = 0;
        if (cnt <= 0 || p_hi - (p_lo & ~7) > COVER_CAP) continue;
        int n2 = 2;
        while (n2 < cnt) n2 <<= 1;
        for (int i = tid; i < n2; i += MK_BLOCK) key[i] = (i < cnt) ? (ix[p_lo + i] >> 1) : 0x7fffffff;
        __syncthreads();
        bitonic_sort_i32(key, n2);
        bool done = false;
        for (int G = 4; G <= 16384 && !done; G <<= 2) {
            if (tid == 0) nh = 0;
            __syncthreads();
            for (int i = tid; i < cnt; i += MK_BLOCK)
                if (i == 0 || key[i] - key[i - 1] > G) {
                    const int q = atomicAdd(&nh, 1);
                    if (q < CMAX) heads[q] = i;
                }
            __syncthreads();
            if (tid == 0) {
                bool ok = (nh <= CMAX);
                const int nw = nh;
                if (ok) {
                    for (int a = 1; a < nw; ++a) {           // the (<= 16) window heads in ascending order
                        const int v = heads[a];
                        int b = a - 1;
                        while (b >= 0 && heads[b] > v) {
                            heads[b + 1] = heads[b];
                            --b;
                        }
                        heads[b + 1] = v;
                    }
                    heads[nw] = cnt;
                    int C = 0;
                    for (int k = 0; k < nw; ++k) {
                        const int st = key[heads[k]] * 2, en = (key[heads[k + 1] - 1] + 1) * 2;
                        wst[k] = st;
                        wof[k] = C * MK_WCHUNK;
                        C += (en - st + MK_WCHUNK - 1) / MK_WCHUNK;
                        if ((int64_t)en > xlen) ok = false; // the pair load of the last column would leave x
                    }
                    ok = ok && C <= CMAX;                    // (a wider gap may still merge many tiny windows)
                    if (ok) {
                        unsigned lens[4][CPW / 4];
                        for (int w = 0; w < 4; ++w)
                            for (int q = 0; q < CPW / 4; ++q) lens[w][q] = 0;
                        for (int k = 0; k < nw; ++k) {
                            const int en = (key[heads[k + 1] - 1] + 1) * 2;
                            for (int c = wof[k] / MK_WCHUNK, g = wst[k]; g < en; ++c, g += MK_WCHUNK) {
                                const int wv = c & 3, i = c >> 2;
                                const int half = ((en - g < MK_WCHUNK) ? en - g : MK_WCHUNK) / 2;
                                wg[tile * CMAX + wv * CPW + i] = g;
                                lens[wv][i >> 2] |= (unsigned)half << (8 * (i & 3));
                            }
                        }
                        for (int w = 0; w < 4; ++w) {
                            for (int q = 0; q < CPW / 4; ++q) wn[(tile * 4 + w) * (CPW / 4) + q] = lens[w][q];
                            wg[tile * CMAX + w * CPW] |= 1;  // bit 0 of every wave's first start: tile is eligible
                        }
                        atomicMax(&stats[0], C);
                        atomicAdd(&stats[1], 1);
                    }
                }
                s_ok = ok ? 1 : 0;
                s_nw = nw;
            }
            __syncthreads();
            done = (s_ok != 0);
            __syncthreads();
        }
        if (done) {
            const int nw = s_nw;
            for (int j = tid; j < cnt; j += MK_BLOCK) {
                const int col = ix[p_lo + j];
                int k = 0;
                for (int q = 1; q < nw; ++q) k += (wst[q] <= col) ? 1 : 0;
                sl[p_lo + j] = (uint16_t)(wof[k] + col - wst[k]);
            }
        }
    }
}

__device__ inline unsigned dict_hash(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 29;
    return (unsigned)k & (DICT_SLOTS - 1);
}

// state: [0] distinct values so far, [1] gave up (more than 256, or the sentinel pattern occurs in the data)
__global__ __launch_bounds__(MK_BLOCK) void dict_collect(int64_t nnz, const double *__restrict__ data,
                                                         unsigned long long *table, int *state) {
    unsigned long long seen0 = DICT_EMPTY, seen1 = DICT_EMPTY;     // the two values this lane confirmed last
    for (int64_t j = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * MK_BLOCK) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(data[j]);
        if (key == seen0 || key == seen1) continue;
        if (key == DICT_EMPTY || __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            state[1] = 1;
            return;
        }
        unsigned h = dict_hash(key);
        for (int probe = 0; probe < DICT_SLOTS; ++probe) {
            unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == DICT_EMPTY) {
                cur = atomicCAS(&table[h], DICT_EMPTY, key);
                if (cur == DICT_EMPTY) {
                    if (atomicAdd(&state[0], 1) + 1 > 256) state[1] = 1;
                    cur = key;
                }
            }
            if (cur == key) break;
            h = (h + 1) & (DICT_SLOTS - 1);
        }
        seen1 = seen0;
        seen0 = key;
    }
}

// one workgroup: the distinct values in ascending order of their bit patterns -> dict[0 .. count)
__global__ __launch_bounds__(MK_BLOCK) void dict_finalize(const unsigned long long *table, const int *state, double *dict) {
    __shared__ unsigned long long k[DICT_SLOTS];
    for (int i = threadIdx.x; i < DICT_SLOTS; i += MK_BLOCK) k[i] = (table[i] == DICT_EMPTY) ? ~0ULL : table[i];
    __syncthreads();
    for (int kk = 2; kk <= DICT_SLOTS; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < DICT_SLOTS; i += MK_BLOCK) {
                const int q = i ^ j;
                if (q > i) {
                    const unsigned long long u = k[i], v = k[q];
                    const bool up = ((i & kk) == 0);
                    if ((u > v) == up) {
                        k[i] = v;
                        k[q] = u;
                    }
                }
            }
            __syncthreads();
        }
    const int count = state[0];
    if (threadIdx.x < 256) dict[threadIdx.x] = (threadIdx.x < count) ? __longlong_as_double((long long)k[threadIdx.x]) : 0.0;
}

__global__ __launch_bounds__(MK_BLOCK) void dict_encode(int64_t nnz, const double *__restrict__ data,
                                                        const double *__restrict__ dict, int count,
                                                        const uint16_t *__restrict__ slots, uint32_t *__restrict__ pk) {
    __shared__ unsigned long long k[256];
    k[threadIdx.x] = (threadIdx.x < count) ? (unsigned long long)__double_as_longlong(dict[threadIdx.x]) : ~0ULL;
    __syncthreads();
    for (int64_t j = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * MK_BLOCK) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(data[j]);
        int lo = 0;                                          // largest index with k[lo] <= key (the key is present)
#pragma unroll
        for (int step = 128; step >= 1; step >>= 1)
            if (lo + step < 256 && k[lo + step] <= key) lo += step;
        pk[j] = (uint32_t)slots[j] | ((uint32_t)lo << 16);
    }
}

// ------------------------------------------------------------------------------------------------ row patterns
// fmt 4.  In a dictionary matrix the packed word of a nonzero is {LDS slot, value code}; relative to the row's lane t
// the word {slot - t, code} is the same for every row of a regular stencil, so a ROW is described by the sequence of
// its relative words -- its pattern -- and a matrix with few distinct patterns (<= 128 of up to 8 entries; 27 boundary cases of a 7-point
// stencil, times the few window layouts a tile can have) by ONE BYTE per row instead of one word per nonzero.
// Patterns are collected as 64-bit hashes in the same open-addressing set as the values, numbered in ascending hash
// order, written into a table of `pmax` words per pattern and then VERIFIED row by row against the table, so that a
// hash collision can only make the builder give up, never change a product.
constexpr int PAT_WORDS = 1024;                              // table capacity in entries (npat * pmax): <= 16 KB of LDS
constexpr int PAT_WORDS_WIDE = 4096;                         // ... of the wide formats (one 4-byte word per entry: 16 KB)

// (pk == null: a matrix without a value dictionary -- the words are the bare LDS slots, fmt 5)
__device__ inline int pat_row_words(const int32_t *__restrict__ ip, const uint32_t *__restrict__ pk,
                                    const uint16_t *__restrict__ sl, int64_t r, int t, int pmax, uint32_t *w) {
    const int lo = ip[r], len = ip[r + 1] - lo;
    if (len > pmax) return -1;
    for (int k = 0; k < len; ++k) {
        const uint32_t word = pk ? pk[lo + k] : (uint32_t)sl[lo + k];
        w[k] = ((word & 0xffffu) - (uint32_t)t) & 0xffffu;
        w[k] |= (word >> 16) << 16;
    }
    return len;
}

// (kd: position of the diagonal entry in the row, 255 = none -- part of a pattern's identity: rows of mirror-image
// boundary tiles can share their relative slots and differ in where the diagonal sits)
__device__ inline unsigned long long pat_hash(int len, int kd, const uint32_t *w) {
    unsigned long long h = 0xcbf29ce484222325ULL ^ (unsigned long long)len ^ ((unsigned long long)kd << 8);
    for (int k = 0; k < len; ++k) {
        h ^= (unsigned long long)w[k];
        h *= 0x100000001b3ULL;
        h ^= h >> 29;
    }
    return h == DICT_EMPTY ? h ^ 1ULL : h;
}

// state: [0] distinct keys, [1] failure flag
__device__ inline void set_insert(unsigned long long *table, int *state, unsigned long long key, int limit) {
    unsigned h = dict_hash(key);
    for (int probe = 0; probe < DICT_SLOTS; ++probe) {
        unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == DICT_EMPTY) {
            cur = atomicCAS(&table[h], DICT_EMPTY, key);
            if (cur == DICT_EMPTY) {
                if (atomicAdd(&state[0], 1) + 1 > limit) state[1] = 1;
                cur = key;
            }
        }
        if (cur == key) return;
        h = (h + 1) & (DICT_SLOTS - 1);
    }
    state[1] = 1;
}

__global__ __launch_bounds__(MK_BLOCK) void pat_maxlen(int64_t nrows, const int32_t *__restrict__ ip,
                                                       const int32_t *__restrict__ wg, int cmax, int *__restrict__ out) {
    int mx = 0;
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        if (!(wg[(r / MK_ROWS_PER_TILE) * cmax] & 1)) continue;        // rows of windowed tiles only
        const int len = ip[r + 1] - ip[r];
        mx = len > mx ? len : mx;
    }
    atomicMax(out, mx);
}

__global__ __launch_bounds__(MK_BLOCK) void pat_collect(int64_t nrows, const int32_t *__restrict__ ip,
                                                        const int32_t *__restrict__ ix,
                                                        const int32_t *__restrict__ wg, int cmax,
                                                        const uint32_t *__restrict__ pk, const uint16_t *__restrict__ sl,
                                                        int pmax, int limit, unsigned long long *table, int *state) {
    unsigned long long seen = DICT_EMPTY;
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        if (!(wg[(r / MK_ROWS_PER_TILE) * cmax] & 1)) continue;
        if (__hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        uint32_t w[32];
        const int len = pat_row_words(ip, pk, sl, r, (int)(r % MK_ROWS_PER_TILE), pmax, w);
        if (len < 0) {
            state[1] = 1;
            return;
        }
        int kd = 255;
        for (int q = 0; q < len; ++q)
            if ((int64_t)ix[ip[r] + q] == r) kd = q;
        const unsigned long long key = pat_hash(len, kd, w);
        if (key == seen) continue;
        set_insert(table, state, key, limit);
        seen = key;
    }
}

// mode 0: number the rows and fill the table (identical writes race benignly); mode 1: compare every row with it.
// padword: what the table holds behind a pattern's end -- the wide kernels read it (relative slot -256 = the lane's
// zero cell, mk_spmv_fmtw.h), the kernels of fmt 4 / 5 stop at plen
// plen[p] = entries of pattern p, plen[256 + p] = position of the DIAGONAL entry in it (255: none) -- the kernel hands
// x[r] to epilogues that ask for it from the LDS window instead of loading it again
__global__ __launch_bounds__(MK_BLOCK) void pat_assign(int64_t nrows, const int32_t *__restrict__ ip,
                                                       const int32_t *__restrict__ ix,
                                                       const int32_t *__restrict__ wg, int cmax,
                                                       const uint32_t *__restrict__ pk, const uint16_t *__restrict__ sl,
                                                       int pmax, int count, const double *__restrict__ sorted_keys,
                                                       uint8_t *__restrict__ pid, uint32_t *__restrict__ pat,
                                                       uint8_t *__restrict__ plen, int mode, uint32_t padword, int *state) {
    __shared__ unsigned long long k[256];
    k[threadIdx.x] = (threadIdx.x < count) ? (unsigned long long)__double_as_longlong(sorted_keys[threadIdx.x]) : ~0ULL;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        if (!(wg[(r / MK_ROWS_PER_TILE) * cmax] & 1)) {
            if (mode == 0) pid[r] = 0;
            continue;
        }
        uint32_t w[32];
        const int len = pat_row_words(ip, pk, sl, r, (int)(r % MK_ROWS_PER_TILE), pmax, w);
        int kd = 255;
        for (int q = 0; q < len; ++q)
            if ((int64_t)ix[ip[r] + q] == r) kd = q;
        if (mode == 0) {
            const unsigned long long key = pat_hash(len, kd, w);
            int lo = 0;
#pragma unroll
            for (int step = 128; step >= 1; step >>= 1)
                if (lo + step < 256 && k[lo + step] <= key) lo += step;
            pid[r] = (uint8_t)lo;
            plen[lo] = (uint8_t)len;
            plen[256 + lo] = (uint8_t)kd;
            for (int q = 0; q < pmax; ++q) pat[lo * pmax + q] = (q < len) ? w[q] : padword;
        } else {
            const int id = pid[r];
            bool same = (plen[id] == len) && (plen[256 + id] == kd);
            for (int q = 0; q < len && same; ++q) same = (pat[id * pmax + q] == w[q]);
            if (!same) state[1] = 1;
        }
    }
}

// fmt 8: the pattern table in the form the kernel takes through the scalar cache -- per entry {byte offset of the x
// value from the lane's own cell, value}; behind a pattern's end {-2048 = the lane's zero cell, +0.0}.  pinfo[p] =
// entries of pattern p | position of its diagonal entry << 8 (255: none)
struct PatEntryHost {
    int off, pad;
    double val;
};
__global__ __launch_bounds__(MK_BLOCK) void ptab_build(int npat, int pmax, const uint32_t *__restrict__ pat,
                                                       const uint8_t *__restrict__ plen, const double *__restrict__ dict,
                                                       PatEntryHost *__restrict__ ptab, int32_t *__restrict__ pinfo) {
    for (int e = blockIdx.x * MK_BLOCK + threadIdx.x; e < npat * pmax; e += gridDim.x * MK_BLOCK) {
        const int p = e / pmax, k = e - p * pmax;
        const uint32_t w = pat[e];
        PatEntryHost en;
        en.pad = 0;
        if (k < (int)plen[p]) {
            en.off = 8 * (int)(short)(w & 0xffffu);
            en.val = dict[(w >> 16) & 0xffu];
        } else {
            en.off = -8 * MK_BLOCK;
            en.val = 0.0;
        }
        ptab[e] = en;
    }
    for (int p = blockIdx.x * MK_BLOCK + threadIdx.x; p < 256; p += gridDim.x * MK_BLOCK)
        pinfo[p] = (p < npat) ? ((int)plen[p] | ((int)plen[256 + p] << 8)) : (255 << 8);
}
// fmt 7: the same for a matrix without dictionary -- per entry the byte offset alone
__global__ __launch_bounds__(MK_BLOCK) void poff_build(int npat, int pmax, const uint32_t *__restrict__ pat,
                                                       const uint8_t *__restrict__ plen, int32_t *__restrict__ poff,
                                                       int32_t *__restrict__ pinfo) {
    for (int e = blockIdx.x * MK_BLOCK + threadIdx.x; e < npat * pmax; e += gridDim.x * MK_BLOCK) {
        const int p = e / pmax, k = e - p * pmax;
        poff[p * (pmax < 16 ? 16 : pmax) + k] = (k < (int)plen[p]) ? 8 * (int)(short)(pat[e] & 0xffffu) : -8 * MK_BLOCK;
    }
    for (int p = blockIdx.x * MK_BLOCK + threadIdx.x; p < 256; p += gridDim.x * MK_BLOCK)
        pinfo[p] = (p < npat) ? ((int)plen[p] | ((int)plen[256 + p] << 8)) : (255 << 8);
}

// ------------------------------------------------------------------------------------------------ sliced ELL values
// fmt 5: the values of every windowed tile in the order the pattern kernel's lanes consume them (mk_spmv_fmt5.h).
// width[T] = longest row of tile T (0 for a tile without windows: it keeps the CSR gather path)
__global__ __launch_bounds__(MK_BLOCK) void sell_width(int64_t nrows, int64_t ntiles, const int32_t *__restrict__ ip,
                                                       const int32_t *__restrict__ wg, int cmax, int32_t *__restrict__ width) {
    __shared__ int mx;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) mx = 0;
        __syncthreads();
        const int64_t r = tile * MK_ROWS_PER_TILE + threadIdx.x;
        if ((wg[tile * cmax] & 1) && r < nrows) atomicMax(&mx, ip[r + 1] - ip[r]);
        __syncthreads();
        if (threadIdx.x == 0) width[tile] = mx;
    }
}

// sdesc[2 T] = start of tile T's block in units of 256 doubles, sdesc[2 T + 1] = its width
// (ds: ints per tile in sdesc -- 2 for fmt 5, 4 for the wide formats, whose third int is the start of the tile's slot block)
__global__ __launch_bounds__(MK_BLOCK) void sell_fill(int64_t nrows, int64_t ntiles, const int32_t *__restrict__ ip,
                                                      const double *__restrict__ dv, const int32_t *__restrict__ sdesc,
                                                      int ds, double *__restrict__ sval) {
    const int t = threadIdx.x;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int w = sdesc[ds * tile + 1];
        if (w == 0) continue;
        double *vb = sval + (int64_t)sdesc[ds * tile] * MK_ROWS_PER_TILE;
        const int64_t r = tile * MK_ROWS_PER_TILE + t;
        const int lo = (r < nrows) ? ip[r] : 0, len = (r < nrows) ? ip[r + 1] - lo : 0;
        for (int k = 0; k < w; ++k) {
            const double v = (k < len) ? dv[lo + k] : 0.0;
            const int64_t at = (k < (w & ~1)) ? (int64_t)(k >> 1) * 512 + 2 * t + (k & 1) : (int64_t)k * 256 + t;
            vb[at] = v;
        }
    }
}

// fmt 6: the LDS slots of a tile's nonzeros, entry k of row t at [(k >> 2) * 1024 + 4 t + (k & 3)]
__global__ __launch_bounds__(MK_BLOCK) void sell_fill_slots(int64_t nrows, int64_t ntiles, const int32_t *__restrict__ ip,
                                                            const uint16_t *__restrict__ sl, const int32_t *__restrict__ sdesc,
                                                            uint16_t *__restrict__ sslot) {
    const int t = threadIdx.x;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int w = sdesc[4 * tile + 1];
        if (w == 0) continue;
        uint16_t *sb = sslot + (int64_t)sdesc[4 * tile + 2] * 1024;
        const int64_t r = tile * MK_ROWS_PER_TILE + t;
        const int lo = (r < nrows) ? ip[r] : 0, len = (r < nrows) ? ip[r + 1] - lo : 0;
        const int w4 = (w + 3) & ~3;
        // (indices into the kernel's LDS counted from its 256 zeros: real slots + 256, padding = the lane's zero cell)
        for (int k = 0; k < w4; ++k) sb[(int64_t)(k >> 2) * 1024 + 4 * t + (k & 3)] = (k < len) ? (uint16_t)(sl[lo + k] + 256) : (uint16_t)t;
    }
}

// ------------------------------------------------------------------------------------------------ column blocks
constexpr int CB_MAX = 16;
struct CbPtrs {
    int32_t *indptr[CB_MAX];
    int32_t *indices[CB_MAX];
    double *data[CB_MAX];
};

// per row: how many of its entries fall into each column block -> cnt[b * (nrows + 1) + r + 1]
__global__ __launch_bounds__(MK_BLOCK) void cb_count(int64_t nrows, const int32_t *__restrict__ ip,
                                                     const int32_t *__restrict__ ix, int bw, int K, int32_t *cnt) {
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        int c[CB_MAX];
#pragma unroll
        for (int b = 0; b < CB_MAX; ++b) c[b] = 0;
        for (int32_t j = ip[r]; j < ip[r + 1]; ++j) {
            const int b = ix[j] / bw;
#pragma unroll
            for (int q = 0; q < CB_MAX; ++q) c[q] += (q == b) ? 1 : 0;
        }
#pragma unroll
        for (int b = 0; b < CB_MAX; ++b)
            if (b < K) cnt[(int64_t)b * (nrows + 1) + r + 1] = c[b];
    }
}

// per row: copy its entries (sorted by column, hence grouped by block) to the blocks' arrays
__global__ __launch_bounds__(MK_BLOCK) void cb_scatter(int64_t nrows, const int32_t *__restrict__ ip,
                                                       const int32_t *__restrict__ ix, const double *__restrict__ dv,
                                                       int bw, CbPtrs P) {
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        int cur = -1, dst = 0;
        for (int32_t j = ip[r]; j < ip[r + 1]; ++j) {
            const int col = ix[j], b = col / bw;
            if (b != cur) {
                cur = b;
                dst = P.indptr[b][r];
            }
            P.indices[b][dst] = col;
            P.data[b][dst] = dv[j];
            ++dst;
        }
    }
}

void plan_free(MkPlan &P) {
    for (mk_csr *B : P.cblocks) mk_csr_destroy(B);
    P.cblocks.clear();
    hipFree(P.d_cbsum);
    hipFree(P.d_carry);
    hipFree(P.d_slots);
    hipFree(P.d_wg);
    hipFree(P.d_wn);
    hipFree(P.d_pk);
    hipFree(P.d_dict);
    hipFree(P.d_pid);
    hipFree(P.d_pat);
    hipFree(P.d_plen);
    hipFree(P.d_sval);
    hipFree(P.d_sdesc);
    hipFree(P.d_sslot);
    hipFree(P.d_ptab);
    hipFree(P.d_pinfo);
    P = MkPlan();
}

int default_format() {
    static int f = [] {
        const char *e = getenv("MK_SPMV_FORMAT");
        int v = e ? atoi(e) : 11;
        return v < 0 ? 0 : (v > 11 ? 11 : v);
    }();
    return f;
}

// Column-block size in bytes of x (0: off, the default -- see cblocks_build)
int64_t colblock_bytes(const mk_csr *A) {
    static int64_t env = [] {
        const char *e = getenv("MK_COLBLOCK_KB");
        return e ? atoll(e) * 1024 : (int64_t)0;
    }();
    static const bool env_set = getenv("MK_COLBLOCK_KB") != nullptr;
    if (A->want_cb_kb < 0 && !env_set) {
        // AUTOMATIC (round 4): long rows gathered from an x several times the chip's L2 capacity -- the transposed operator
        // of a tall least-squares problem (lls/lsqr.py:264: 1e6 rows of ~20 entries over 4e6 columns, x = 32 MB) pulls a
        // 64-byte sector through the fabric for nearly every nonzero on the gather path (350 us per product), and its
        // tiles are too long for the resident-tile format.  In 4 MiB column blocks the same product takes 232 us (LSQR
        // 1 665 -> 2 050 passes per second, gpurun_out of tools/r04_cb.sh): with ~20 entries per row the 24 B per row and
        // block of carried sums are small beside the sectors saved.  (5 entries per row: slower, see cblocks_build.)
        const int64_t xbytes = 8 * A->x_len();
        const bool long_rows = A->nrows > 0 && A->nnz >= 12 * A->nrows;
        if (long_rows && xbytes >= ((int64_t)16 << 20) && A->ex.mode < 0) return -((int64_t)8 << 20);   // (negative: automatic)
        return 0;
    }
    const int64_t b = A->want_cb_kb >= 0 ? (int64_t)A->want_cb_kb * 1024 : env;
    return b < 0 ? 0 : b;
}
static void cblocks_drop(MkPlan &P) {
    for (mk_csr *B : P.cblocks) mk_csr_destroy(B);
    P.cblocks.clear();
    hipFree(P.d_cbsum);
    P.d_cbsum = nullptr;
}

int resident_plan(const mk_csr *A, MkPlan &P, bool forced);
constexpr int64_t RT_SLICE_BYTES = 3 << 19;              // a column phase of format 3: 1.5 MiB of x

// Column blocks for a plain-CSR matrix (scattered columns) whose x is more than two blocks long.  OFF by default:
// measured on BASELINE configs[2] (random n = 1e6, 5 nnz/row; x = 8 MB against 4 MiB of L2 per XCD) a product costs
// 53 us unblocked, 60 us in 4 blocks of 2 MB and 100 us in 8 blocks of 1 MB -- every extra launch re-reads the row
// pointers, carries the running sums through memory (24 B per row) and pays its own ramp-up, which together cost
// more than the 128-byte lines the blocked gathers no longer pull through the fabric.  Kept (mk_csr_set_colblocks,
// MK_COLBLOCK_KB) because the mechanism -- carried row sums, gate / epilogue split over launches -- is exact and
// the trade-off is different for wider rows.
int cblocks_build(const mk_csr *A, int64_t blk_override = 0) {
    MkPlan &P = A->plan;
    // automatic blocks (colblock_bytes < 0): 8 MiB of x each -- the fewest blocks whose tiles still fit the resident-tile kernel
    // on the transposed least-squares operator (4 blocks of ~5 entries per row: 176 us per product; 6 MiB 191, 4 MiB 208 us) --
    // and 4 MiB if a tile of an 8 MiB block outgrows the LDS budget
    int64_t blk = blk_override ? blk_override : colblock_bytes(A);
    const bool automatic = blk < 0;
    if (automatic) blk = -blk;
    if (blk <= 0 || A->ex.mode >= 0 || A->nnz == 0) return MK_OK;
    const int64_t xbytes = 8 * A->x_len();
    if (xbytes <= 2 * blk) return MK_OK;
    int K = (int)((xbytes + blk - 1) / blk);
    if (automatic) {
        //  the automatic choice was measured on x slices that fit an L2: beyond CB_MAX blocks of this size a clamped
        // K would make the slices larger than that again, and the second copy of the matrix (12 B per nonzero + K row-pointer
        // arrays) must fit comfortably beside everything else -- otherwise the matrix keeps its single launch
        if (K > CB_MAX) return MK_OK;
        size_t free_b = 0, total_b = 0;
        const int64_t extra = 12 * A->nnz + (int64_t)K * 4 * (A->nrows + 1) + 8 * A->nrows;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || extra > (int64_t)(free_b / 4)) return MK_OK;
    }
    if (K > CB_MAX) K = CB_MAX;
    int64_t bw = (A->x_len() + K - 1) / K;
    bw = (bw + 255) / 256 * 256;
    K = (int)((A->x_len() + bw - 1) / bw);
    if (K < 2) return MK_OK;
    hipStream_t st = mk_ctx().stream;
    const int64_t n1 = A->nrows + 1;
    int32_t *d_cnt = nullptr;
    MK_HIP(hipMalloc((void **)&d_cnt, sizeof(int32_t) * (size_t)(K * n1)));
    MK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (size_t)(K * n1), st));
    int grid = (int)((A->nrows + MK_BLOCK - 1) / MK_BLOCK);
    grid = grid < 1 ? 1 : (grid > 16384 ? 16384 : grid);
    hipLaunchKernelGGL(cb_count, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, A->d_indices, (int)bw, K, d_cnt);
    std::vector<int32_t> h((size_t)(K * n1));
    MK_HIP(hipMemcpyAsync(h.data(), d_cnt, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, st));
    MK_HIP(hipStreamSynchronize(st));
    hipFree(d_cnt);
    CbPtrs ptrs{};
    for (int b = 0; b < K; ++b) {
        int32_t *row = h.data() + (size_t)b * n1;
        for (int64_t r = 0; r < A->nrows; ++r) row[r + 1] += row[r];        // exclusive scan on the host (one-off)
        mk_csr *B = nullptr;
        int rc = mk_csr_alloc(A->nrows, A->ncols, row[A->nrows], &B);
        if (rc != MK_OK) return rc;
        B->plan.built = true;                                // a block is always a plain CSR stream
        P.cblocks.push_back(B);
        MK_HIP(hipMemcpyAsync(B->d_indptr, row, sizeof(int32_t) * (size_t)n1, hipMemcpyHostToDevice, st));
        ptrs.indptr[b] = B->d_indptr;
        ptrs.indices[b] = B->d_indices;
        ptrs.data[b] = B->d_data;
    }
    hipLaunchKernelGGL(cb_scatter, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, A->d_indices, A->d_data,
                       (int)bw, ptrs);
    MK_HIP(hipMalloc((void **)&P.d_cbsum, sizeof(double) * (size_t)(A->nrows > 0 ? A->nrows : 1)));
    MK_HIP(hipStreamSynchronize(st));
    MK_HIP(hipGetLastError());
    // Round 4: a block's rows are short (a long row cut into K pieces) and its columns span one slice of x that an L2 holds, so
    // its launch takes the resident-tile kernel of format 3 with ONE phase -- the tile's (column, value) stream by DMA into LDS,
    // lane t walks row t with a cursor, running sums in from / out to the carry vector -- instead of the gather path's two
    // staged passes per 2 048 nonzeros: the lls `A' u` product 231 -> measured in DESIGN.md 3.1-6.  All blocks or none (one grid).
    static const char *ecb = getenv("MK_CB_RESIDENT");
    if (!ecb || atoi(ecb) > 0) {
        bool all = true;
        for (size_t bi = 0; bi < P.cblocks.size(); ++bi) {
            mk_csr *B = P.cblocks[bi];
            plan_free(B->plan);
            B->plan.built = true;
            if (resident_plan(B, B->plan, true) != MK_OK || B->plan.fmt != 3) all = false;
            hipFree(B->plan.d_carry);                        // (a block never runs as a stepped pair product)
            B->plan.d_carry = nullptr;
            // phases over the block's OWN slice of x (bw columns from b * bw), 1.5 MiB each: a slice larger than an L2's share
            // is walked in lockstep like a whole format-3 matrix
            const int64_t c0 = (int64_t)bi * bw;
            int64_t kk = (8 * bw + RT_SLICE_BYTES - 1) / RT_SLICE_BYTES;
            static const char *ecp = getenv("MK_CB_PHASES");
            if (ecp && atoi(ecp) > 0) kk = atoi(ecp);
            kk = kk < 1 ? 1 : (kk > 64 ? 64 : kk);
            B->plan.rt_k = (int)kk;
            B->plan.rt_w = (int)((bw + kk - 1) / kk);
            B->plan.rt_c0 = (int)c0;
            B->plan.rt_reg = 0;
        }
        if (!all && automatic) {
            // a tile of some block is too long for LDS: smaller blocks once, if they still are at most CB_MAX; otherwise the
            // matrix keeps its single launch (blocks on the gather path were measured slower than that: DESIGN.md 3.1-4)
            cblocks_drop(P);
            if (blk > ((int64_t)4 << 20) && (xbytes + blk / 2 - 1) / (blk / 2) <= CB_MAX) return cblocks_build(A, -(blk / 2));
            return MK_OK;
        }
        if (!all)
            for (mk_csr *B : P.cblocks) {
                plan_free(B->plan);                          // (no leak of what resident_plan allocated)
                B->plan.built = true;
            }
    }
    return MK_OK;
}

// build the plan of an owning (non-alias) matrix; on any failure the matrix stays on the plain CSR path
// longest tile stream (nonzeros of 256 rows, counted from the 4-aligned start the kernels copy from)
__global__ __launch_bounds__(MK_BLOCK) void tile_extent_kernel(const int32_t *__restrict__ indptr, int64_t nrows,
                                                               int64_t ntiles, int *__restrict__ out) {
    int mx = 0, mrow = 0;
    for (int64_t t = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; t < ntiles; t += (int64_t)gridDim.x * MK_BLOCK) {
        const int64_t r0 = t * MK_ROWS_PER_TILE;
        const int64_t r1 = (r0 + MK_ROWS_PER_TILE < nrows) ? r0 + MK_ROWS_PER_TILE : nrows;
        const int len = indptr[r1] - (indptr[r0] & ~3);
        mx = len > mx ? len : mx;
    }
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        const int len = indptr[r + 1] - indptr[r];
        mrow = len > mrow ? len : mrow;
    }
    atomicMax(out, mx);
    atomicMax(out + 1, mrow);                                // longest row
}

// fmt 3: a matrix that stays on plain CSR, whose x is longer than an XCD's L2 (4 MiB; the gather path still wins up
// to 5 MiB, tools/fmt_compare.py) and whose tiles fit LDS eight
// to a CU: resident tiles, gathers ordered by column block (mk_device.h).  Slices of <= 1.5 MiB: measured best on
// 1e6 x 5 random (BiCGSTAB's fused product: K = 4 / 6 / 8 -> 38.8 / 37.9 / 38.7 us, 50.6 us with one phase, 53 us on
// the gather path; tools/ubench/spmv_cb.hip has the kernel variants that were tried).
constexpr int RT_CAP_MAX = 2560;                             // 30 KB of LDS per workgroup
int resident_plan(const mk_csr *A, MkPlan &P, bool forced) {
    const int64_t xbytes = 8 * A->x_len();
    if (!forced && (xbytes <= 5 * (1 << 20) || getenv("MK_NO_RESIDENT"))) return MK_OK;   // (crossover measured at 650 k rows x 5)
    if (A->ex.mode >= 0 && !forced) return MK_OK;            // (partitioned matrices: their x slices are short already)
    hipStream_t st = mk_ctx().stream;
    int *d_max = nullptr, h_mm[2] = {0, 0};
    int &h_max = h_mm[0];
    if (hipMalloc((void **)&d_max, 2 * sizeof(int)) != hipSuccess) return MK_OK;
    hipMemsetAsync(d_max, 0, 2 * sizeof(int), st);
    int grid = (int)((A->ntiles + MK_BLOCK - 1) / MK_BLOCK);
    grid = grid > 1024 ? 1024 : grid;
    hipLaunchKernelGGL(tile_extent_kernel, dim3(grid), dim3(MK_BLOCK), 0, st, A->d_indptr, A->nrows, A->ntiles, d_max);
    const bool ok = hipMemcpyAsync(h_mm, d_max, 2 * sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
                    hipStreamSynchronize(st) == hipSuccess;
    hipFree(d_max);
    if (!ok || h_max < 1) return MK_OK;
    const int cap = (h_max + 255) / 256 * 256;
    if (cap > RT_CAP_MAX) return MK_OK;                      // long rows: the chunked gather path
    // short rows and more tiles than resident workgroups: pairs of tiles, the second one in registers (mk_spmv_fmt3r.h).
    // MK_RT_REG=0 keeps one tile per step (A/B measurements).
    P.max_row = h_mm[1];
    static const char *ereg = getenv("MK_RT_REG");
    const int allow = ereg ? atoi(ereg) : 1;
    const int rt_reg = (allow >= 1 && h_mm[1] <= 5 && A->ntiles > MK_MAXP && 8 * A->x_len() < ((int64_t)1 << 31)) ? 1 : 0;
    static const char *estep = getenv("MK_RT_STEPPED");
    const bool stepped = rt_reg && A->ntiles > 2 * (int64_t)MK_MAXP && (!estep || atoi(estep) > 0);
    // column phases (speed only: a row's sum runs in column order whatever the phases).  Round 5 sweep (profiles/
    // r05_scattered_phase_sweep.txt): a product of ONE step per workgroup is fastest with slices of about 1 MiB (config 3: 8 - 9
    // phases, +1 % over the 6 of 1.5 MiB), a product of MANY steps with slices of 0.6 MiB (the 4e6-row `A v`: 12 - 16 phases,
    // LSQR +7 %): two neighbouring slices then share an L2 and workgroups a phase apart still hit it
    const int64_t slice = stepped ? (int64_t)640 << 10 : (int64_t)1 << 20;   // (column blocks set their own phases: cblocks_build)
    int64_t k = (xbytes + slice - 1) / slice;
    static const char *env = getenv("MK_RT_PHASES");
    if (env && atoi(env) > 0) k = atoi(env);
    k = k < 1 ? 1 : (k > 64 ? 64 : k);
    P.fmt = 3;
    P.rt_cap = cap;
    P.rt_k = (int)k;
    P.rt_w = (int)((A->x_len() + k - 1) / k);
    P.rt_reg = rt_reg;
    if (stepped) {
        if (hipMalloc((void **)&P.d_carry, sizeof(double) * MK_CARRY_SLOTS * MK_MAXP * MK_BLOCK) != hipSuccess) P.d_carry = nullptr;
    }
    return MK_OK;
}

// The values (and, `with_slots`, the LDS slots) of every windowed tile in tile-sliced ELL order: widths -> host scan ->
// block starts -> fill.  ds = ints per tile in the descriptor (2: fmt 5; 4: the wide formats).  Refuses too much
// padding (below).  On success the plan owns d_sval / d_sdesc (/ d_sslot).
bool sell_build(const mk_csr *A, MkPlan &P, int ds, bool with_slots) {
    hipStream_t st = mk_ctx().stream;
    const int cmax = P.wide ? MK_WCHUNKS_WIDE : MK_WCHUNKS_MAX;
    int32_t *d_sdesc = nullptr, *d_width = nullptr;
    double *d_sval = nullptr;
    uint16_t *d_sslot = nullptr;
    std::vector<int32_t> wd((size_t)A->ntiles), sd((size_t)ds * (size_t)A->ntiles, 0);
    int tg = (int)(A->ntiles < 16384 ? A->ntiles : 16384);
    bool ok = hipMalloc((void **)&d_sdesc, sizeof(int32_t) * sd.size()) == hipSuccess &&
              hipMalloc((void **)&d_width, sizeof(int32_t) * wd.size()) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(sell_width, dim3(tg), dim3(MK_BLOCK), 0, st, A->nrows, A->ntiles, A->d_indptr, P.d_wg, cmax, d_width);
        ok = hipMemcpyAsync(wd.data(), d_width, sizeof(int32_t) * wd.size(), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    hipFree(d_width);
    int64_t blocks = 0, sblocks = 0;
    if (ok) {
        for (int64_t t = 0; t < A->ntiles; ++t) {
            sd[ds * t] = (int32_t)blocks;
            sd[ds * t + 1] = wd[t];
            blocks += wd[t];
            if (ds == 4) {
                sd[ds * t + 2] = (int32_t)sblocks;
                if (with_slots) sblocks += (wd[t] + 3) / 4;
            }
            if (wd[t] > 32) ok = false;                       // (the row-walk kernels keep a row in <= 32 registers)
        }
        // (A->nnz: upper bound of the nonzeros the blocks hold)
        // padding the blocks may hold: 12.5 % beside row patterns (a stencil's rows are even; more means the tiles are
        // not what the format is for), 50 % for the slot format -- measured on banded matrices with ragged rows
        // (tools/ragged_time.py, 4e6 rows): longest / mean row 1.0, 1.2, 1.33, 1.5 -> 176, 198, 196, 193 us against
        // 272, 271, 254, 227 us on the gather path, which pays 12 bytes per nonzero and a gather for each
        const int64_t lim8 = with_slots ? 12 : 9;            // (eighths of the nonzeros)
        ok = ok && blocks > 0 && blocks < (int64_t)0x7fffffff && 8 * blocks * MK_ROWS_PER_TILE <= lim8 * A->nnz + 8 * 2048;
        if (with_slots) ok = ok && 8 * sblocks * 1024 <= 13 * A->nnz + 8 * 8192;
    }
    ok = ok && hipMalloc((void **)&d_sval, sizeof(double) * (size_t)(blocks * MK_ROWS_PER_TILE + 2)) == hipSuccess;
    if (ok && with_slots) ok = hipMalloc((void **)&d_sslot, sizeof(uint16_t) * (size_t)(sblocks * 1024 + 8)) == hipSuccess;
    if (ok) {
        ok = hipMemcpyAsync(d_sdesc, sd.data(), sizeof(int32_t) * sd.size(), hipMemcpyHostToDevice, st) == hipSuccess;
        hipLaunchKernelGGL(sell_fill, dim3(tg), dim3(MK_BLOCK), 0, st, A->nrows, A->ntiles, A->d_indptr, A->d_data, d_sdesc, ds, d_sval);
        if (with_slots)
            hipLaunchKernelGGL(sell_fill_slots, dim3(tg), dim3(MK_BLOCK), 0, st, A->nrows, A->ntiles, A->d_indptr, P.d_slots,
                               d_sdesc, d_sslot);
        ok = ok && hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
    }
    if (!ok) {
        hipFree(d_sdesc);
        hipFree(d_sval);
        hipFree(d_sslot);
        return false;
    }
    P.d_sdesc = d_sdesc;
    P.d_sval = d_sval;
    P.d_sslot = d_sslot;
    P.sell_entries = blocks * MK_ROWS_PER_TILE;
    P.slot_entries = sblocks * 1024;
    return true;
}

// fmt 2 -> fmt 4 when every row of every windowed tile follows one of a few patterns (see above).  Any failure leaves
// the matrix in fmt 2.  `raw`: fmt 1 -> fmt 5, the same for a matrix without a value dictionary -- patterns of the bare
// slots, and the values copied into tile-sliced ELL order (any failure, or more than 12.5 % of padding: stays fmt 1).
bool pattern_plan(const mk_csr *A, MkPlan &P, bool raw) {
    if (getenv("MK_NO_PATTERNS")) return false;
    const uint32_t *words = raw ? nullptr : P.d_pk;
    const int cmax = P.wide ? MK_WCHUNKS_WIDE : MK_WCHUNKS_MAX;
    const int cap_words = P.wide ? PAT_WORDS_WIDE : PAT_WORDS;
    hipStream_t st = mk_ctx().stream;
    int *d_state = nullptr;
    unsigned long long *d_table = nullptr;
    double *d_keys = nullptr;
    uint8_t *d_pid = nullptr, *d_plen = nullptr;
    uint32_t *d_pat = nullptr;
    const char *why = "";
    auto cleanup = [&]() {
        if (*why && getenv("MK_DEBUG_PLAN")) fprintf(stderr, "mikrylov: no row patterns (%s)\n", why);
        hipFree(d_state);
        hipFree(d_table);
        hipFree(d_keys);
        hipFree(d_pid);
        hipFree(d_plen);
        hipFree(d_pat);
        return false;
    };
    int h_state[2] = {0, 0};
    int grid = (int)((A->nrows + MK_BLOCK - 1) / MK_BLOCK);
    grid = grid > 4096 ? 4096 : (grid < 1 ? 1 : grid);
    if (hipMalloc((void **)&d_state, 2 * sizeof(int)) != hipSuccess) return cleanup();
    hipMemsetAsync(d_state, 0, 2 * sizeof(int), st);
    hipLaunchKernelGGL(pat_maxlen, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, P.d_wg, cmax, d_state);
    if (hipMemcpyAsync(h_state, d_state, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return cleanup();
    const int maxlen = h_state[0];
    const int pmax = maxlen <= 8 ? 8 : (maxlen <= 16 ? 16 : (maxlen <= 32 ? 32 : 0));
    why = "rows of more than 32 entries";
    if (pmax == 0 || maxlen < 1) return cleanup();
    why = "rows of more than 8 entries in the 16-chunk cover";
    if (raw && !P.wide && pmax != 8) return cleanup();       // (the fmt 5 kernel keeps a row's values in 8 registers)
    why = "allocation";
    const int limit = cap_words / pmax > 256 ? 256 : cap_words / pmax;
    std::vector<unsigned long long> empty(DICT_SLOTS, DICT_EMPTY);
    if (hipMalloc((void **)&d_table, sizeof(unsigned long long) * DICT_SLOTS) != hipSuccess ||
        hipMalloc((void **)&d_keys, sizeof(double) * 256) != hipSuccess ||
        hipMalloc((void **)&d_pid, (size_t)A->nrows + 16) != hipSuccess ||
        hipMalloc((void **)&d_plen, 512) != hipSuccess ||
        hipMalloc((void **)&d_pat, sizeof(uint32_t) * cap_words) != hipSuccess)
        return cleanup();
    hipMemcpyAsync(d_table, empty.data(), sizeof(unsigned long long) * DICT_SLOTS, hipMemcpyHostToDevice, st);
    hipMemsetAsync(d_state, 0, 2 * sizeof(int), st);
    hipMemsetAsync(d_pat, 0, sizeof(uint32_t) * cap_words, st);
    hipMemsetAsync(d_plen, 0, 512, st);
    hipLaunchKernelGGL(pat_collect, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, A->d_indices, P.d_wg, cmax, words, P.d_slots,
                       pmax, limit, d_table, d_state);
    why = "too many distinct patterns";
    if (hipMemcpyAsync(h_state, d_state, sizeof(h_state), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess || h_state[1] || h_state[0] < 1 || h_state[0] > limit)
        return cleanup();
    const int count = h_state[0];
    hipLaunchKernelGGL(dict_finalize, dim3(1), dim3(MK_BLOCK), 0, st, d_table, d_state, d_keys);
    for (int mode = 0; mode < 2; ++mode)
        hipLaunchKernelGGL(pat_assign, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, A->d_indices, P.d_wg, cmax,
                           words, P.d_slots, pmax, count, d_keys, d_pid, d_pat, d_plen, mode, P.wide ? 0xff00u : 0u, d_state);
    why = "verification against the table";
    if (hipMemcpyAsync(h_state, d_state, sizeof(h_state), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess || h_state[1])
        return cleanup();
    why = "padding of the value blocks";
    if (raw && !sell_build(A, P, P.wide ? 4 : 2, false)) return cleanup();
    if (P.wide) {                                            // fmt 8: the table as {offset, value} entries; fmt 7: offsets
        why = "pattern table";
        void *d_ptab = nullptr;
        int32_t *d_pinfo = nullptr;
        const size_t tbytes = sizeof(PatEntryHost) * (size_t)(count * (pmax < 16 ? 16 : pmax) + 64);
        bool ok = hipMalloc(&d_ptab, tbytes) == hipSuccess &&
                  hipMalloc((void **)&d_pinfo, sizeof(int32_t) * 256) == hipSuccess;
        if (ok) {
            hipMemsetAsync(d_ptab, 0, tbytes, st);
            if (raw)
                hipLaunchKernelGGL(poff_build, dim3(4), dim3(MK_BLOCK), 0, st, count, pmax, d_pat, d_plen,
                                   static_cast<int32_t *>(d_ptab), d_pinfo);
            else
                hipLaunchKernelGGL(ptab_build, dim3(4), dim3(MK_BLOCK), 0, st, count, pmax, d_pat, d_plen, P.d_dict,
                                   static_cast<PatEntryHost *>(d_ptab), d_pinfo);
            ok = hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
        }
        if (!ok) {
            hipFree(d_ptab);
            hipFree(d_pinfo);
            if (raw) {                                       // (the value blocks sell_build stored just above)
                hipFree(P.d_sval);
                hipFree(P.d_sdesc);
                P.d_sval = nullptr;
                P.d_sdesc = nullptr;
                P.sell_entries = 0;
            }
            return cleanup();
        }
        P.d_ptab = d_ptab;
        P.d_pinfo = d_pinfo;
    }
    why = "";
    P.d_pid = d_pid;
    P.d_pat = d_pat;
    P.d_plen = d_plen;
    d_pid = nullptr;
    d_pat = nullptr;
    d_plen = nullptr;
    P.npat = count;
    P.pmax = pmax;
    P.fmt = P.wide ? (raw ? 7 : 8) : (raw ? 5 : 4);
    if (getenv("MK_DEBUG_PLAN")) fprintf(stderr, "mikrylov: %lld rows, %d row patterns of <= %d entries (fmt %d)\n", (long long)A->nrows, count, pmax, P.fmt);
    // the per-nonzero streams are not read any more (tiles without windows gather from the CSR arrays)
    hipFree(P.d_pk);
    hipFree(P.d_slots);
    P.d_pk = nullptr;
    P.d_slots = nullptr;
    cleanup();
    return true;
}

// the value dictionary of a covered matrix: <= 256 distinct bit patterns -> P.d_dict / P.ndict and one packed word
// {slot | code << 16} per nonzero in P.d_pk.  Returns 1 = built, 0 = too many values (nothing kept), -1 = HIP failure.
int dictionary_build(const mk_csr *A, MkPlan &P) {
    hipStream_t st = mk_ctx().stream;
    unsigned long long *d_table = nullptr;
    int h_state[2] = {0, 0};
    auto drop = [&](int rc) {
        hipFree(d_table);
        hipFree(P.d_dict);
        hipFree(P.d_pk);
        P.d_dict = nullptr;
        P.d_pk = nullptr;
        return rc;
    };
    if (hipMalloc((void **)&d_table, sizeof(unsigned long long) * DICT_SLOTS + 2 * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&P.d_dict, sizeof(double) * 256) != hipSuccess)
        return drop(-1);
    int *d_state = reinterpret_cast<int *>(d_table + DICT_SLOTS);
    std::vector<unsigned long long> empty(DICT_SLOTS, DICT_EMPTY);
    if (hipMemcpyAsync(d_table, empty.data(), sizeof(unsigned long long) * DICT_SLOTS, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(d_state, 0, 2 * sizeof(int), st) != hipSuccess)
        return drop(-1);
    int grid = (int)((A->nnz + MK_BLOCK - 1) / MK_BLOCK);
    grid = grid > 4096 ? 4096 : grid;
    // (every thread's first lookups go to the same few table slots: 512 workgroups instead of 4096 take that hot spot
    // from 4.8-6 ms to a fraction of it, and the stream itself needs no more)
    hipLaunchKernelGGL(dict_collect, dim3(grid > 512 ? 512 : grid), dim3(MK_BLOCK), 0, st, A->nnz, A->d_data, d_table, d_state);
    if (hipMemcpyAsync(h_state, d_state, sizeof(h_state), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return drop(-1);
    if (h_state[1] || h_state[0] > 256 || h_state[0] < 1) return drop(0);
    // (the fmt 2 kernel copies whole 1 KiB pieces of the word stream: room for one piece behind the last nonzero)
    const size_t pkpad = (size_t)A->nnz + 256 + MK_CSR_PAD;
    if (hipMalloc((void **)&P.d_pk, sizeof(uint32_t) * pkpad) != hipSuccess) return drop(-1);
    hipMemsetAsync(P.d_pk, 0, sizeof(uint32_t) * pkpad, st);
    hipLaunchKernelGGL(dict_finalize, dim3(1), dim3(MK_BLOCK), 0, st, d_table, d_state, P.d_dict);
    hipLaunchKernelGGL(dict_encode, dim3(grid), dim3(MK_BLOCK), 0, st, A->nnz, A->d_data, P.d_dict, h_state[0], P.d_slots,
                       P.d_pk);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return drop(-1);
    hipFree(d_table);
    P.ndict = h_state[0];
    return 1;
}

// windows of every tile: the cover with 16 chunks / 2048 nonzeros per tile, or the wide one (32 / 8192)
bool cover_build(const mk_csr *A, MkPlan &P, bool wide) {
    hipStream_t st = mk_ctx().stream;
    const size_t pad = (size_t)A->nnz + MK_CSR_PAD;
    const int cmax = wide ? MK_WCHUNKS_WIDE : MK_WCHUNKS_MAX;
    int *d_stats = nullptr;
    int h_stats[4] = {0, 0, 0, 0};
    bool ok = hipMalloc((void **)&P.d_slots, sizeof(uint16_t) * pad) == hipSuccess &&
              hipMalloc((void **)&P.d_wg, sizeof(int32_t) * (size_t)cmax * (size_t)A->ntiles) == hipSuccess &&
              hipMalloc((void **)&P.d_wn, sizeof(uint32_t) * (size_t)(cmax / 4) * (size_t)A->ntiles) == hipSuccess &&
              hipMalloc((void **)&d_stats, 4 * sizeof(int)) == hipSuccess;
    ok = ok && hipMemsetAsync(P.d_slots, 0, sizeof(uint16_t) * pad, st) == hipSuccess &&
         hipMemsetAsync(d_stats, 0, 4 * sizeof(int), st) == hipSuccess;
    if (ok) {
        const int grid = (int)(A->ntiles < 4096 ? A->ntiles : 4096);
        if (wide)
            hipLaunchKernelGGL((cover_kernel<MK_WCHUNKS_WIDE, MK_WIDE_TILE>), dim3(grid), dim3(MK_BLOCK), 0, st, A->d_indptr,
                               A->d_indices, A->nrows, A->ntiles, A->x_len(), P.d_wg, P.d_wn, P.d_slots, d_stats);
        else
            hipLaunchKernelGGL((cover_kernel<MK_WCHUNKS_MAX, MK_SPMV_TILE>), dim3(grid), dim3(MK_BLOCK), 0, st, A->d_indptr,
                               A->d_indices, A->nrows, A->ntiles, A->x_len(), P.d_wg, P.d_wn, P.d_slots, d_stats);
        ok = hipMemcpyAsync(h_stats, d_stats, sizeof(h_stats), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
    }
    hipFree(d_stats);
    P.wide = wide;
    P.wchunks = h_stats[0];
    P.covered = h_stats[1];
    return ok;
}

// ------------------------------------------------------------------------------------------------ fmt 9
// z-marching bricks (mk_spmv_fmt9.h): is the matrix 7-point class -- every column offset in {0, +-1, +-L, +-P} -- with
// strides the brick geometry can tile (nrows % P == 0; whole aligned bricks when L % 128 == 0 and P % 4L == 0, partly empty
// ones -- the general-geometry kernels -- otherwise, pencil_plan) and <= 256 distinct values?  Then a row
// is described EXACTLY by a 63-bit key, 9 bits per offset in column order: 0 = no entry, 1 + the value's dictionary code
// otherwise.  Distinct keys (<= 256) are collected in the open-addressing set the dictionary uses, numbered in ascending
// order (deterministic), and every row gets the number of its key: no hashing, nothing to verify.
//
// One rank's SLAB of such a matrix (rows of whole planes, columns localised to [own | plane below | plane above] by
// mk_csr_localize mode 0) is of the class too: a received column of a first-plane row r is the -P neighbour iff it is
// nrows + r, one of a last-plane row the +P neighbour iff it is nrows + lo + (r - (nrows - P)).  Localising renumbers in
// place: a row keeps the storage order of its GLOBAL columns, which is the order the scalar loop adds in and the order of
// the slots -- the builder insists on it (slots strictly ascending along every row), so the bits are those of one device.
// stats: [0] smallest |offset| above 1, [1] largest |offset|, [2] longest row   (received columns, >= nrows, are not offsets)
__global__ __launch_bounds__(MK_BLOCK) void pen_scan(int64_t nrows, const int32_t *__restrict__ ip,
                                                     const int32_t *__restrict__ ix, int *__restrict__ stats) {
    int mn = 0x7fffffff, mx = 0, ml = 0;
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        const int lo = ip[r], hi = ip[r + 1];
        ml = (hi - lo > ml) ? hi - lo : ml;
        if (hi - lo > 7) continue;                           // (the caller gives up on ml > 7)
        for (int j = lo; j < hi; ++j) {
            if (ix[j] >= nrows) continue;
            const int64_t d = (int64_t)ix[j] - r;
            const int64_t a = d < 0 ? -d : d;
            if (a > 1) {
                mn = a < mn ? (int)a : mn;
                mx = a > mx ? (int)a : mx;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int omn = __shfl_down(mn, off), omx = __shfl_down(mx, off), oml = __shfl_down(ml, off);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        ml = oml > ml ? oml : ml;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&stats[0], mn);
        atomicMax(&stats[1], mx);
        atomicMax(&stats[2], ml);
    }
}

// slab geometry of a localised matrix: rows, widths of the lower / upper window of received columns (0 or P)
struct PenSlab {
    int64_t nrows, lo, hi;
};

// position 0..6 of column `col` in row r (-1: outside the class)
__device__ inline int pen_slot(int64_t col, int64_t r, int64_t L, int64_t P, const PenSlab &sl) {
    if (col >= sl.nrows) {
        const int64_t h = col - sl.nrows;
        if (sl.lo == P && r < P && h == r) return 0;
        if (sl.hi == P && r >= sl.nrows - P && h == sl.lo + (r - (sl.nrows - P))) return 6;
        return -1;
    }
    const int64_t d = col - r;
    if (d == -P) return 0;
    if (d == -L) return 1;
    if (d == -1) return 2;
    if (d == 0) return 3;
    if (d == 1) return 4;
    if (d == L) return 5;
    if (d == P) return 6;
    return -1;
}

// the key of row r (0 = the row does not fit the class: an offset outside the set, or a value outside the dictionary)
__device__ inline unsigned long long pen_row_key(const int32_t *__restrict__ ip, const int32_t *__restrict__ ix,
                                                 const double *__restrict__ data, const unsigned long long *dk, int count,
                                                 int64_t r, int64_t L, int64_t P, const PenSlab &sl) {
    const int lo = ip[r], hi = ip[r + 1];
    unsigned long long key = 0;
    if (hi - lo > 7 || hi == lo) return 0;
    int kprev = -1;
    for (int j = lo; j < hi; ++j) {
        const int k = pen_slot(ix[j], r, L, P, sl);
        if (k <= kprev) return 0;                            // (outside the class, or not stored in the order the march adds in)
        kprev = k;
        const unsigned long long v = (unsigned long long)__double_as_longlong(data[j]);
        int c = 0;                                           // largest index with dk[c] <= v
#pragma unroll
        for (int step = 128; step >= 1; step >>= 1)
            if (c + step < count && dk[c + step] <= v) c += step;
        if (dk[c] != v) return 0;
        key |= (unsigned long long)(c + 1) << (9 * k);
    }
    return key;
}

// pass 1 (sorted == null): every row's key into the set; pass 2: the number of the row's key among the sorted keys -> pid
__global__ __launch_bounds__(MK_BLOCK) void pen_rows(int64_t nrows, const int32_t *__restrict__ ip, const int32_t *__restrict__ ix,
                                                     const double *__restrict__ data, const double *__restrict__ dict, int count,
                                                     int64_t L, int64_t P, PenSlab sl, unsigned long long *table, int *state,
                                                     const double *__restrict__ sorted, int nkeys, uint8_t *__restrict__ pid) {
    __shared__ unsigned long long dk[256], sk[256];
    dk[threadIdx.x] = (threadIdx.x < count) ? (unsigned long long)__double_as_longlong(dict[threadIdx.x]) : ~0ULL;
    if (sorted) sk[threadIdx.x] = (threadIdx.x < nkeys) ? (unsigned long long)__double_as_longlong(sorted[threadIdx.x]) : ~0ULL;
    __syncthreads();
    unsigned long long seen = 0;
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        const unsigned long long key = pen_row_key(ip, ix, data, dk, count, r, L, P, sl);
        if (key == 0) {
            state[1] = 1;
            return;
        }
        if (!sorted) {
            if (key == seen) continue;
            if (__hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            set_insert(table, state, key, 256);
            seen = key;
        } else {
            int c = 0;
#pragma unroll
            for (int step = 128; step >= 1; step >>= 1)
                if (c + step < nkeys && sk[c + step] <= key) c += step;
            pid[r] = (uint8_t)c;
        }
    }
}

// 64 bytes per pattern: the seven values in column order (+0.0 where the row has no entry), the presence mask, 0
__global__ __launch_bounds__(MK_BLOCK) void pen_table(int nkeys, const double *__restrict__ sorted, const double *__restrict__ dict,
                                                      uint32_t *__restrict__ tab) {
    const int j = threadIdx.x;
    if (j >= nkeys) return;
    const unsigned long long key = (unsigned long long)__double_as_longlong(sorted[j]);
    unsigned mask = 0;
    for (int k = 0; k < 7; ++k) {
        const unsigned f = (unsigned)(key >> (9 * k)) & 511u;
        const unsigned long long v = f ? (unsigned long long)__double_as_longlong(dict[f - 1]) : 0ULL;
        tab[16 * j + 2 * k] = (uint32_t)v;
        tab[16 * j + 2 * k + 1] = (uint32_t)(v >> 32);
        mask |= f ? (1u << k) : 0u;
    }
    tab[16 * j + 14] = mask;
    tab[16 * j + 15] = 0;
}

// format 10: per row the 7-bit presence mask, and the values in position-major order sval[k * nrows + r] (+0.0 where the row
// has no entry at offset k); state[1] is raised by a row with an offset outside the class
__global__ __launch_bounds__(MK_BLOCK) void pen_stream_fill(int64_t nrows, const int32_t *__restrict__ ip, const int32_t *__restrict__ ix,
                                                            const double *__restrict__ data, int64_t L, int64_t P, PenSlab sl,
                                                            uint8_t *__restrict__ pid, double *__restrict__ sval, int *state) {
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        const int lo = ip[r], hi = ip[r + 1];
        double v[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        unsigned mask = 0;
        bool bad = hi - lo > 7;
        int kprev = -1;
        for (int j = lo; j < hi && !bad; ++j) {
            const int k = pen_slot(ix[j], r, L, P, sl);
            if (k <= kprev) {                                // (outside the class, or not stored in the order the march adds in)
                bad = true;
                break;
            }
            kprev = k;
            mask |= 1u << k;
            v[k] = data[j];
        }
        if (bad) {
            state[1] = 1;
            return;
        }
        pid[r] = (uint8_t)mask;
#pragma unroll
        for (int k = 0; k < 7; ++k) sval[(int64_t)k * nrows + r] = v[k];
    }
}

// format 11: is the format-10 matrix symmetric bit for bit?  A lower entry (slot k = 0, 1, 2 at distance off = P, L, 1) must be
// there exactly when row r - off has the mirrored upper entry (slot 6 - k), with the same bits; seen from every row this covers
// every pair.  A slab's first plane is exempt at slot 0 (its -P entries mirror the NEIGHBOUR's +P entries: kept beside the four
// arrays), its last plane's +P entries have no local mirror at all.  state[1] is raised by the first violation.
__global__ __launch_bounds__(MK_BLOCK) void pen_sym_check(int64_t nrows, int64_t L, int64_t P, int64_t lo_rows,
                                                          const uint8_t *__restrict__ pid, const double *__restrict__ sval,
                                                          int *state) {
    const int64_t off[3] = {P, L, 1};
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        const unsigned m = pid[r];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k == 0 && r < lo_rows) continue;             // (a slab's first plane: the mirror lives on the rank below)
            const int64_t q = r - off[k];
            const bool have = (m >> k) & 1u;
            const bool mirror = q >= 0 && ((pid[q] >> (6 - k)) & 1u);
            if (have != mirror) bad = true;
            else if (have && __double_as_longlong(sval[(int64_t)k * nrows + r]) != __double_as_longlong(sval[(int64_t)(6 - k) * nrows + q]))
                bad = true;
        }
        if (bad) {
            state[1] = 1;
            return;
        }
    }
}

// ... then only the diagonal and the upper values are kept: sym[(k - 3) * nrows + r] = sval[k * nrows + r], k = 3 .. 6, and
// behind them the -P values of the first plane (what a slab's first plane multiplies the plane below with; zeros otherwise)
__global__ __launch_bounds__(MK_BLOCK) void pen_sym_pack(int64_t nrows, int64_t P, const double *__restrict__ sval,
                                                         double *__restrict__ sym) {
    for (int64_t i = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; i < 4 * nrows + P; i += (int64_t)gridDim.x * MK_BLOCK)
        sym[i] = i < 4 * nrows ? sval[3 * nrows + i] : sval[i - 4 * nrows];
}

// Below this many rows the windowed pattern format (fmt 4) keeps the matrix: a brick march needs a few thousand
// (brick, chunk) items of >= 8 planes to fill the chip, and a cache-resident product is latency bound either way
int64_t pencil_min_rows() {
    static int64_t v = [] {
        const char *e = getenv("MK_PENCIL_MIN_ROWS");
        return e ? atoll(e) : (int64_t)1 << 21;
    }();
    return v;
}

// true: P holds format 9.  false: the matrix is not of the class (P untouched apart from freed scratch).
// bricks per line, brick rows per plane (the last brick of a line / the last group of four lines may be partly empty)
static inline int64_t pen_bx_of(int64_t L) { return (L + 127) / 128; }
static inline int64_t pen_by_of(int64_t L, int64_t PP) { return ((PP + L - 1) / L + 3) / 4; }
// whole aligned bricks (round 5's geometry; the vectors' 16-byte alignment is the C ABI's)?
static inline bool pen_aligned(int64_t L, int64_t PP) { return L % 128 == 0 && PP % (4 * L) == 0; }

void pencil_geometry(const mk_csr *A, MkPlan &P, int64_t L, int64_t PP) {
    P.pen_L = L;
    P.pen_P = PP;
    P.pen_nz = (int)(A->nrows / PP);
    P.pen_ny = (int)((PP + L - 1) / L);
    P.pen_bx = (int)pen_bx_of(L);
    P.pen_bpp = (int)(pen_bx_of(L) * pen_by_of(L, PP));
    static const char *env_gen = getenv("MK_PEN_GEN");       // (1: the general-geometry kernels on aligned geometries too, A/B runs)
    P.pen_gen = (!pen_aligned(L, PP) || (env_gen && atoi(env_gen) == 1)) ? 2 : 0;
    // XCD-contiguous deal: an eighth of the plane's bricks per XCD; on a general geometry ceil(bpp / 8) with empty item slots
    // behind the last brick, from 64 bricks per plane on (below that the empty slots would idle whole XCDs)
    P.pen_per = (P.pen_bpp % 8 == 0) ? P.pen_bpp / 8 : ((P.pen_gen && P.pen_bpp >= 64) ? (P.pen_bpp + 7) / 8 : 0);
    P.pen_xlo = A->loc_lo ? A->nrows : -1;                   // a slab: where the neighbours' planes sit in the input vector
    P.pen_xhi = A->loc_hi ? A->nrows + A->loc_lo : -1;
    // chunks: the kernel keeps two workgroups per CU resident (its register ring), so 512 (brick, chunk) items fill the
    // chip in one round; more chunks only add pipeline fills and re-read two planes per chunk start (512^3, tools/
    // r05_pencil_variants.sh: 4 chunks of 132 planes 441 us, 8 of 66 436, one of 516 421).  At least 6 planes per chunk, a
    // multiple of the ring depth so that only the matrix's last chunk has planes left over.
    static const char *env_zc = getenv("MK_PENCIL_ZC");
    int chunks = (512 + P.pen_bpp - 1) / P.pen_bpp;
    int zc = (P.pen_nz + chunks - 1) / chunks;
    zc = zc < MK_PEN_R ? MK_PEN_R : zc;
    if (env_zc && atoi(env_zc) > 0) zc = atoi(env_zc);
    zc = (zc + MK_PEN_R - 1) / MK_PEN_R * MK_PEN_R;
    P.pen_zc = zc;
    P.pen_chunks = (P.pen_nz + zc - 1) / zc;
}

// want: 9 = dictionary + patterns only; 10 = also the streamed-value twin (format 10) when the matrix has too many values or
// patterns for format 9
bool pencil_plan(const mk_csr *A, MkPlan &P, bool forced, int want) {
    // square, or one rank's slab with its columns localised to [own | plane below | plane above] (halo exchange)
    if (A->ncols != A->nrows + A->loc_lo + A->loc_hi || A->nnz > 7 * A->nrows || A->nrows < 1024 || A->alias || A->ex.mode == 1)
        return false;
    if (A->ex.mode == 0 && (A->ex.n_local != A->nrows || A->ex.n_halo != A->loc_lo + A->loc_hi)) return false;
    if (!forced && A->nrows < pencil_min_rows()) return false;
    hipStream_t st = mk_ctx().stream;
    int *d_stats = nullptr;
    unsigned long long *d_table = nullptr;
    double *d_dict = nullptr, *d_keys = nullptr;
    uint8_t *d_pid = nullptr;
    uint32_t *d_tab = nullptr;
    auto drop = [&]() {
        hipFree(d_stats);
        hipFree(d_table);
        hipFree(d_dict);
        hipFree(d_keys);
        hipFree(d_pid);
        hipFree(d_tab);
        return false;
    };
    int h_stats[3] = {0x7fffffff, 0, 0};
    if (hipMalloc((void **)&d_stats, sizeof(h_stats)) != hipSuccess) return drop();
    if (hipMemcpyAsync(d_stats, h_stats, sizeof(h_stats), hipMemcpyHostToDevice, st) != hipSuccess) return drop();
    int grid = (int)((A->nrows + MK_BLOCK - 1) / MK_BLOCK);
    grid = grid > 2048 ? 2048 : grid;
    hipLaunchKernelGGL(pen_scan, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, A->d_indices, d_stats);
    if (hipMemcpyAsync(h_stats, d_stats, sizeof(h_stats), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return drop();
    // strides: the smallest and the largest |offset| above 1.  ONE far stride only (a 5-point stencil: offsets {0, +-1, +-M}) is
    // marched line by line: planes of M rows, cut into lines of 128 rows that have no +-L entries (the kernels are index based)
    int64_t L = h_stats[0], PP = h_stats[1];
    if (h_stats[2] > 7 || L <= 1 || PP < L) return drop();
    const bool two_d = (L == PP);
    if (L == PP) {
        // (measured, tools/r06_march_sizes.py: CG on 2000^2 16.8 k passes per second on either format, on 4000^2 3.86 k on the
        //  windowed format 4 against 4.22 k marched: chosen automatically from four times the 3-D threshold on)
        if (!forced && A->nrows < 4 * pencil_min_rows()) return drop();
        L = 128;
    }
    if (L >= PP || A->nrows % PP != 0 || A->nrows / PP < 2 || pen_bx_of(L) * pen_by_of(L, PP) > (1 << 24)) return drop();
    // partly empty bricks: at least half of the lanes must have rows (L = 132, 9 lines: 39 %; L = 37: 29 %)
    if (!pen_aligned(L, PP) && 2 * PP < 512 * pen_bx_of(L) * pen_by_of(L, PP)) return drop();
    // ... and chosen AUTOMATICALLY only where the march wins (tools/r06_march_sizes.py, profiles/r06_march_sizes.txt: CG passes
    // per second, windowed -> march): bricks at least 90 % full from the usual 2^21 rows on (250^3 2 901 -> 3 931, 500^3 430 -> 584),
    // bricks 78 % full (L = 200, 300, 400) only from 2^24 rows on (200^3, 8 M rows: 7 828 -> 6 506; 300^3 1 982 -> 2 220; 400^3
    // 884 -> 939)
    if (!forced && !pen_aligned(L, PP) && 10 * PP < 9 * 512 * pen_bx_of(L) * pen_by_of(L, PP) && A->nrows < ((int64_t)1 << 24)) return drop();
    const size_t slack = (size_t)(4 * L + 256);              // rows past the end that a lane without rows may index (GEN)
    if ((A->loc_lo != 0 && A->loc_lo != PP) || (A->loc_hi != 0 && A->loc_hi != PP)) return drop();   // (whole planes only)
    const PenSlab sl{A->nrows, A->loc_lo, A->loc_hi};
    // the dictionary (values only: no per-nonzero words)
    int h_state[2] = {0, 0};
    if (hipMalloc((void **)&d_table, sizeof(unsigned long long) * DICT_SLOTS + 2 * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&d_dict, sizeof(double) * 256) != hipSuccess ||
        hipMalloc((void **)&d_keys, sizeof(double) * 256) != hipSuccess)
        return drop();
    int *d_state = reinterpret_cast<int *>(d_table + DICT_SLOTS);
    std::vector<unsigned long long> empty(DICT_SLOTS, DICT_EMPTY);
    auto reset_set = [&]() {
        return hipMemcpyAsync(d_table, empty.data(), sizeof(unsigned long long) * DICT_SLOTS, hipMemcpyHostToDevice, st) == hipSuccess &&
               hipMemsetAsync(d_state, 0, 2 * sizeof(int), st) == hipSuccess;
    };
    auto read_state = [&]() {
        return hipMemcpyAsync(h_state, d_state, sizeof(h_state), hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipStreamSynchronize(st) == hipSuccess;
    };
    if (!reset_set()) return drop();
    int gnz = (int)((A->nnz + MK_BLOCK - 1) / MK_BLOCK);
    hipLaunchKernelGGL(dict_collect, dim3(gnz > 512 ? 512 : gnz), dim3(MK_BLOCK), 0, st, A->nnz, A->d_data, d_table, d_state);
    auto stream_twin = [&]() -> bool {                      // format 10: mask byte + seven value arrays
        if (want < 10) return drop();
        double *d_sval = nullptr;
        if (hipMalloc((void **)&d_pid, (size_t)A->nrows + 64 + slack) != hipSuccess ||
            hipMalloc((void **)&d_sval, sizeof(double) * (7 * (size_t)A->nrows + slack) + 64) != hipSuccess) {
            hipFree(d_sval);
            return drop();
        }
        hipMemsetAsync(d_pid, 0, (size_t)A->nrows + 64 + slack, st);
        hipMemsetAsync(d_sval + 7 * (size_t)A->nrows, 0, sizeof(double) * slack + 64, st);
        hipMemsetAsync(d_state, 0, 2 * sizeof(int), st);
        hipLaunchKernelGGL(pen_stream_fill, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, A->d_indices, A->d_data, L, PP,
                           sl, d_pid, d_sval, d_state);
        if (!read_state() || h_state[1] || hipGetLastError() != hipSuccess) {
            hipFree(d_sval);
            return drop();
        }
        P.fmt = 10;
        P.sell_entries = 7 * A->nrows;
        if (want >= 11) {                                    // format 11: symmetric bit for bit? then 4 of the 7 arrays do
            double *d_sym = nullptr;
            hipMemsetAsync(d_state, 0, 2 * sizeof(int), st);
            hipLaunchKernelGGL(pen_sym_check, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, L, PP, A->loc_lo ? PP : (int64_t)0,
                               d_pid, d_sval, d_state);
            if (read_state() && !h_state[1] && hipGetLastError() == hipSuccess &&
                hipMalloc((void **)&d_sym, sizeof(double) * (4 * (size_t)A->nrows + (size_t)PP + slack) + 64) == hipSuccess) {
                hipMemsetAsync(d_sym + 4 * (size_t)A->nrows + (size_t)PP, 0, sizeof(double) * slack + 64, st);
                hipLaunchKernelGGL(pen_sym_pack, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, PP, d_sval, d_sym);
                if (hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess) {
                    hipFree(d_sval);
                    d_sval = d_sym;
                    P.fmt = 11;
                    P.sell_entries = 4 * A->nrows + PP;
                } else {
                    hipFree(d_sym);
                }
            }
        }
        hipFree(d_stats);
        hipFree(d_table);
        hipFree(d_dict);
        hipFree(d_keys);
        P.d_pid = d_pid;
        P.d_sval = d_sval;
        P.npat = 0;
        pencil_geometry(A, P, L, PP);
        P.pen_nol = two_d ? 1 : 0;
        return true;
    };
    if (!read_state()) return drop();
    if (h_state[1] || h_state[0] > 256 || h_state[0] < 1) return stream_twin();
    const int ndict = h_state[0];
    hipLaunchKernelGGL(dict_finalize, dim3(1), dim3(MK_BLOCK), 0, st, d_table, d_state, d_dict);
    // the rows' keys
    if (!reset_set()) return drop();
    hipLaunchKernelGGL(pen_rows, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, A->d_indices, A->d_data, d_dict, ndict,
                       L, PP, sl, d_table, d_state, (const double *)nullptr, 0, (uint8_t *)nullptr);
    if (!read_state()) return drop();
    if (h_state[1] || h_state[0] > 256 || h_state[0] < 1) return stream_twin();
    const int nkeys = h_state[0];
    hipLaunchKernelGGL(dict_finalize, dim3(1), dim3(MK_BLOCK), 0, st, d_table, d_state, d_keys);
    if (hipMalloc((void **)&d_pid, (size_t)A->nrows + 64 + slack) != hipSuccess ||
        hipMalloc((void **)&d_tab, 64 * 256) != hipSuccess)
        return drop();
    hipMemsetAsync(d_pid, 0, (size_t)A->nrows + 64 + slack, st);
    hipMemsetAsync(d_tab, 0, 64 * 256, st);
    hipLaunchKernelGGL(pen_rows, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, A->d_indptr, A->d_indices, A->d_data, d_dict, ndict,
                       L, PP, sl, d_table, d_state, (const double *)d_keys, nkeys, d_pid);
    hipLaunchKernelGGL(pen_table, dim3(1), dim3(MK_BLOCK), 0, st, nkeys, d_keys, d_dict, d_tab);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return drop();
    hipFree(d_stats);
    hipFree(d_table);
    hipFree(d_keys);
    P.fmt = 9;
    P.d_pid = d_pid;
    P.d_ptab = d_tab;
    P.d_dict = d_dict;
    P.ndict = ndict;
    P.npat = nkeys;
    pencil_geometry(A, P, L, PP);
    P.pen_nol = two_d ? 1 : 0;
    return true;
}

// Which format: `want` (mk_csr_set_format, MK_SPMV_FORMAT; default 8) is the highest one the builder may choose.
//   0 / 3     plain CSR (3: resident tiles when x is longer than an L2)
//   1 .. 5    the 16-chunk cover (tiles of <= 2048 nonzeros): 1 slots + values, 2 + dictionary, 4 + row patterns,
//             5 row patterns + values in tile-sliced ELL order (no dictionary)
//   6 .. 8    when that cover reaches less than half of the tiles (rows of more than 8 entries): the wide cover
//             (32 chunks, 8192 nonzeros) with 8 dictionary + row patterns, 7 row patterns + streamed values,
//             6 streamed slots + values.  want == 6 or 7 given explicitly also applies them to a matrix the narrow
//             cover would have served (format comparisons, tests).
int plan_build(const mk_csr *A) {
    MkPlan &P = A->plan;
    P.built = true;
    P.fmt = 0;
    int want = A->want_fmt >= 0 ? A->want_fmt : default_format();
    if (want == 11 && A->no_sym) want = 10;                  // (format 11 has kernels for plain products and CG only)
    if (A->nnz == 0 || A->ntiles == 0 || !mk_ctx().ready || A->host_fn || A->comp_kind) return MK_OK;
    auto plain = [&]() -> int {                              // plain CSR: x too long for an L2?
        if (cblocks_build(A) != MK_OK) {                     // (separate launches per column block: off by default)
            for (mk_csr *B : P.cblocks) mk_csr_destroy(B);
            P.cblocks.clear();
            hipFree(P.d_cbsum);
            P.d_cbsum = nullptr;
        }
        if (P.cblocks.empty() && want != 0) return resident_plan(A, P, want == 3);
        return MK_OK;
    };
    if (want == 0 || want == 3) return plain();
    if (want >= 9 && (A->march_pref != 0 || A->want_fmt >= 9) && pencil_plan(A, P, A->want_fmt >= 9, want))
        return MK_OK;                                        // fmt 9 / 10: 7-point-class matrices (mk_csr_march_pref)
    auto fail = [&](const char *what) {
        plan_free(P);
        P.built = true;
        return mk_fail(MK_ERR_HIP, "windowed format: %s failed (the matrix stays on the CSR path)", what);
    };
    const bool force_wide = (A->want_fmt == 6 || A->want_fmt == 7);
    if (!cover_build(A, P, false)) return fail("cover kernel");
    const bool narrow_ok = 2 * P.covered >= A->ntiles;
    if (narrow_ok && !force_wide) {
        P.fmt = 1;
        if (want < 2) return MK_OK;
        const int rc = dictionary_build(A, P);
        if (rc < 0) return fail("value dictionary");
        if (rc == 0) {                                       // too many distinct values: windows with raw values
            if (want >= 5) pattern_plan(A, P, true);         // ... in pattern order when the rows follow patterns (fmt 5)
            return MK_OK;
        }
        P.fmt = 2;
        if (want >= 4) pattern_plan(A, P, false);
        return MK_OK;
    }
    if (want < 6) {                                          // mostly scattered columns or long rows: the CSR path
        plan_free(P);
        P.built = true;
        return plain();
    }
    // ---- wide tiles
    plan_free(P);
    P.built = true;
    if (!cover_build(A, P, true)) return fail("wide cover kernel");
    if (2 * P.covered < A->ntiles) {
        plan_free(P);
        P.built = true;
        return plain();
    }
    if (want >= 8) {
        const int rc = dictionary_build(A, P);
        if (rc < 0) return fail("value dictionary");
        if (rc > 0) {
            if (pattern_plan(A, P, false)) return MK_OK;     // fmt 8
            hipFree(P.d_pk);                                 // (no wide kernel reads a per-nonzero word stream)
            hipFree(P.d_dict);
            P.d_pk = nullptr;
            P.d_dict = nullptr;
            P.ndict = 0;
        }
    }
    if (want >= 7 && pattern_plan(A, P, true)) return MK_OK; // fmt 7
    if (sell_build(A, P, 4, true)) {                         // fmt 6
        P.fmt = 6;
        hipFree(P.d_slots);                                  // (the kernel reads the slots from their ELL copy)
        P.d_slots = nullptr;
        return MK_OK;
    }
    plan_free(P);                                            // (more than 50 % padding: rows of very different lengths)
    P.built = true;
    return plain();
}

}  // namespace

const MkPlan *mk_csr_plan(const mk_csr *A) {
    const mk_csr *owner = A->base ? A->base : A;
    if (!owner->plan.built) plan_build(owner);
    return &owner->plan;
}

void mk_csr_plan_reset(const mk_csr *A) {
    plan_free(A->plan);
}

// The brick march is a per-MATRIX format, but whether it pays depends on the LOOP: CG's product epilogue (and a plain
// product's) loads nothing, so the march's software pipeline runs undisturbed (512^3: CG 442 -> 582 passes per second);
// the other loops' epilogues load their vectors inside the pipelined loop and every such load waits for everything issued
// before it (512^3: MINRES 369 -> 292, SYMMLQ 357 -> 270, BiCGSTAB 254 -> 229; tools/r05_solvers_on_bricks.py).  So the
// solver being created says what it is, and an automatically chosen format follows the last one -- unless other solvers
// are alive on the matrix (their partial-sum counts were sized for the format in use) or the caller fixed the format.
// a solver was created on (+1) / removed from (-1) A: counted on A's owner and on every matrix a composite borrows
void mk_csr_count_users(const mk_csr *A, int delta) {
    if (!A) return;
    const mk_csr *o = A->base ? A->base : A;
    o->solver_users += delta;
    if (o->comp_kind == 4 && o->grid) {
        for (const mk_csr *B : o->grid->blk) mk_csr_count_users(B, delta);
    } else if (o->comp_kind) {
        mk_csr_count_users(o->comp_a, delta);
        if (o->comp_kind <= 3) mk_csr_count_users(o->comp_b, delta);
    }
}

void mk_csr_march_pref(const mk_csr *A, int pref) {
    const mk_csr *o = A->base ? A->base : A;
    // (live solvers -- on the matrix itself or on a composite that borrows it, mk_csr_count_users -- sized their partial-sum
    //  counts for the format in use: the matrix keeps it while there is one)
    const bool pinned = o->plan.built && o->solver_users > 0;
    if (!o->comp_kind && !o->host_fn && !pinned) {
        // format 11 (the symmetric march) is CG's and plain products': any other loop gets the same matrix as format 10
        const bool no_sym = (pref == 0);
        if (no_sym != o->no_sym) {
            o->no_sym = no_sym;
            if (o->plan.built && ((no_sym && o->plan.fmt == 11) || (!no_sym && o->plan.fmt == 10 && o->want_fmt == 11))) {
                hipStreamSynchronize(mk_ctx().stream);
                plan_free(o->plan);
            }
        }
    }
    if (o->want_fmt >= 9 || o->comp_kind || o->host_fn) return;
    if (pinned) return;
    o->march_pref = pref;
    if (!o->plan.built) return;
    const bool march = mk_fmt_march(o->plan.fmt);
    if ((pref == 0 && march) || (pref == 1 && !march && o->nrows >= pencil_min_rows() && o->nnz <= 7 * o->nrows && o->ex.mode != 1)) {
        hipStreamSynchronize(mk_ctx().stream);
        plan_free(o->plan);
    }
}

extern "C" int mk_csr_set_format(mk_csr *A, int fmt) {
    MK_REQUIRE_INIT();
    MK_ARG(A != nullptr && fmt >= -1 && fmt <= 11);
    if (A->base) return mk_fail(MK_ERR_ARG, "mk_csr_set_format: set the format on the matrix a composed operator was built from");
    MK_HIP(hipStreamSynchronize(mk_ctx().stream));
    plan_free(A->plan);
    A->want_fmt = fmt;
    return MK_OK;
}

extern "C" int mk_csr_format_info(const mk_csr *A, int32_t *fmt, int64_t *tiles_windowed, int32_t *lds_chunks,
                                  int32_t *dict_size, int64_t *matrix_bytes_per_product) {
    MK_REQUIRE_INIT();
    MK_ARG(A != nullptr);
    const MkPlan *P = mk_csr_plan(A);
    if (fmt) *fmt = P->fmt;
    if (mk_fmt_march(P->fmt)) {                             // one byte per row + the pattern table (9) / seven (10) or four (11) values per row
        if (tiles_windowed) *tiles_windowed = 0;
        if (lds_chunks) *lds_chunks = 0;
        if (dict_size) *dict_size = P->ndict;
        if (matrix_bytes_per_product)
            *matrix_bytes_per_product = A->nrows + (P->fmt == 9 ? 64 * (int64_t)P->npat : (P->fmt == 10 ? 56 : 32) * A->nrows);
        return MK_OK;
    }
    const bool windowed = (P->fmt == 1 || P->fmt == 2 || P->fmt >= 4);
    if (tiles_windowed) *tiles_windowed = windowed ? P->covered : 0;
    if (lds_chunks) *lds_chunks = windowed ? P->wchunks : (P->fmt == 3 ? P->rt_k : 0);
    if (dict_size) *dict_size = P->ndict;
    if (matrix_bytes_per_product) {
        // bytes of matrix data one product streams from HBM (x and y not included): per nonzero 4 + 8 (CSR),
        // 2 + 8 (windows) or 4 (windows + dictionary: one packed word), plus row pointers and the window descriptors
        int64_t b = 4 * (A->nrows + 1);
        if (P->fmt == 0 || P->fmt == 3) b += 12 * A->nnz + (P->fmt == 3 ? 4 * A->nrows : 0);   // (fmt 3 reads every row pointer twice)
        else {
            // nonzeros of windowed tiles are not kept; the mixed case is bounded by the covered share
            const double share = A->ntiles ? (double)P->covered / (double)A->ntiles : 0.0;
            // (fmt 4: one byte per ROW and no row pointers for the windowed tiles)
            // (fmt 5: the same byte per row, the padded values of the windowed tiles and 8 bytes per tile for their blocks)
            // (fmt 6, 7, 8: the wide twins -- fmt 6 streams padded values and padded uint16 slots and reads no pattern byte)
            const double per = (P->fmt >= 4) ? 0.0 : ((P->fmt == 2) ? 4.0 : 10.0);
            b += (int64_t)(A->nnz * (share * per + (1.0 - share) * 12.0)) + (P->wide ? 160 : 80) * A->ntiles;
            if (P->fmt >= 4)
                b += (int64_t)(P->fmt == 6 ? 0.0 : share * (double)A->nrows) - (int64_t)(share * 4.0 * (double)(A->nrows + 1));
            if (P->fmt == 5) b += 8 * P->sell_entries + 8 * A->ntiles;
            if (P->fmt == 6 || P->fmt == 7) b += 8 * P->sell_entries + 2 * P->slot_entries + 16 * A->ntiles;
        }
        *matrix_bytes_per_product = b;
    }
    return MK_OK;
}

extern "C" int mk_csr_launch_info(const mk_csr *A, int32_t *grid, int32_t *tile_map) {
    MK_ARG(A != nullptr);
    if (grid) *grid = mk_grid_spmv_for(A);
    if (tile_map) *tile_map = mk_tile_map(A);                // (orders 3, 4: parameters through mk_csr_tile_order)
    return MK_OK;
}

extern "C" int mk_csr_set_tile_order(mk_csr *A, int32_t order, int32_t stripe, int32_t plane, int32_t nontemporal) {
    MK_ARG(A != nullptr && order >= -1 && order <= 4 && stripe >= 0 && plane >= 0 && nontemporal >= -1 && nontemporal <= 1);
    if (A->base) return mk_fail(MK_ERR_ARG, "mk_csr_set_tile_order: set it on the matrix a composed operator was built from");
    A->want_map = order;
    A->want_stripe = stripe;
    A->want_plane = plane;
    A->want_nt = nontemporal;
    return MK_OK;
}

extern "C" int mk_csr_tile_order(const mk_csr *A, int32_t *order, int32_t *stripe, int32_t *plane, int32_t *nontemporal) {
    MK_ARG(A != nullptr);
    const int m = mk_tile_map(A);
    if (order) *order = m;
    if (stripe) *stripe = (m == 3 || m == 4) ? mk_tile_stripe(A) : 0;
    if (plane) *plane = (m == 4) ? mk_tile_plane(A) : 0;
    if (nontemporal) *nontemporal = mk_stream_nt(A);
    return MK_OK;
}

extern "C" int mk_csr_colblocks(const mk_csr *A, int32_t *nblocks) {
    MK_REQUIRE_INIT();
    MK_ARG(A != nullptr && nblocks != nullptr);
    const MkPlan *P = mk_csr_plan(A);
    *nblocks = (int32_t)P->cblocks.size();
    return MK_OK;
}

extern "C" int mk_csr_set_colblocks(mk_csr *A, int32_t block_kb) {
    MK_REQUIRE_INIT();
    MK_ARG(A != nullptr && block_kb >= -1);
    if (A->base) return mk_fail(MK_ERR_ARG, "mk_csr_set_colblocks: set it on the matrix a composed operator was built from");
    MK_HIP(hipStreamSynchronize(mk_ctx().stream));
    plan_free(A->plan);
    A->want_cb_kb = block_kb;
    return MK_OK;
}

// dump rows of the general-geometry march (mk_spmv_fmt9.h, GEN): 512 doubles per workgroup of the largest grid; what lands
// there is never read.  One per context, allocated on first use, released by mk_shutdown.
double *mk_pen_dump() {
    MkContext &c = mk_ctx();
    if (!c.pen_dump && hipMalloc((void **)&c.pen_dump, sizeof(double) * 512 * (size_t)MK_MAXP) != hipSuccess) c.pen_dump = nullptr;
    return c.pen_dump;
}

extern "C" int mk_csr_march_info(const mk_csr *A, int64_t *info, int32_t cap) {
    MK_REQUIRE_INIT();
    MK_ARG(A != nullptr && info != nullptr && cap >= 0);
    const MkPlan *P = mk_csr_plan(A);
    const bool on = mk_fmt_march(P->fmt);
    const int64_t v[MK_MARCH_INFO_LEN] = {on ? P->fmt : 0, P->pen_L, P->pen_P, P->pen_nz, P->pen_ny, P->pen_bx,
                                          P->pen_bx ? P->pen_bpp / P->pen_bx : 0, P->pen_zc, P->pen_chunks, P->pen_gen, P->pen_per,
                                          P->npat};
    for (int k = 0; k < cap && k < MK_MARCH_INFO_LEN; ++k) info[k] = on ? v[k] : 0;
    return MK_OK;
}

extern "C" int mk_csr_pencil_info(const mk_csr *A, int64_t *stride_line, int64_t *stride_plane, int32_t *planes,
                                  int32_t *planes_per_chunk, int32_t *chunks, int32_t *patterns) {
    MK_REQUIRE_INIT();
    MK_ARG(A != nullptr);
    const MkPlan *P = mk_csr_plan(A);
    const bool on = mk_fmt_march(P->fmt);
    if (stride_line) *stride_line = on ? P->pen_L : 0;
    if (stride_plane) *stride_plane = on ? P->pen_P : 0;
    if (planes) *planes = on ? P->pen_nz : 0;
    if (planes_per_chunk) *planes_per_chunk = on ? P->pen_zc : 0;
    if (chunks) *chunks = on ? P->pen_chunks : 0;
    if (patterns) *patterns = on ? P->npat : 0;
    return MK_OK;
}
