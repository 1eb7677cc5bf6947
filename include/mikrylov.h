/* mikrylov.h -- C ABI of libmikrylov.so, the MI355X (gfx950) Krylov inner-loop engine.
 *
 * The reference (PythonOptimizers/pykrylov) is pure Python and has no FFI of its own;
 * its extension point is the duck-typed operator protocol `y = op * x`
 * (pykrylov/linop/linop.py:356-369) and the solver protocol
 * `Solver(op, **kw).solve(rhs, **kw)` (pykrylov/generic/generic.py:65-98).
 * Each entry point below names the reference code whose work it takes over.  The
 * Python package `pykrylov_amd` binds these with ctypes (INTEGRATION.md shows the
 * stub a reference maintainer would add).
 *
 * Conventions: every function returns 0 on success and a negative mk_status on
 * failure (text via mk_last_error()); no C++ exception crosses the boundary; all
 * "_dev" pointers are device (HBM) addresses obtained from mk_malloc; host arrays are
 * borrowed for the duration of the call only; one host thread per process drives one
 * device (one process per GPU).  Vectors are contiguous fp64, indices int32.
 */
#ifndef MIKRYLOV_H
#define MIKRYLOV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MK_VERSION 100

/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported (`nm -D`). */
#define MK_API __attribute__((visibility("default")))

typedef enum {
    MK_OK = 0,
    MK_ERR_HIP = -1,        /* a HIP runtime call failed (no GPU, OOM, launch failure) */
    MK_ERR_ARG = -2,        /* bad argument (null pointer, negative size, shape mismatch) */
    MK_ERR_STATE = -3,      /* call out of order (e.g. iterate before setup) */
    MK_ERR_COMM = -4,       /* RCCL failure or communicator missing */
    MK_ERR_UNSUPPORTED = -5
} mk_status;

/* ------------------------------------------------------------------ context ---- */
MK_API int mk_version(void);
/* Digest (16 hex digits) of the sources this binary was compiled from -- csrc/*.hip, csrc/*.h and this header, as
 * pykrylov_amd/build.py `source_sha()` computes it -- or "unknown" for a build made without build.py.  The Python package
 * refuses to load a library whose digest differs from its tree's (MIKRYLOV_ALLOW_STALE=1 overrides). */
MK_API const char *mk_build_info(void);
/* Bind the calling process to `device`, create the compute stream.  Idempotent. */
MK_API int mk_init(int device);
MK_API int mk_shutdown(void);
MK_API const char *mk_last_error(void);
/* name: at least 256 bytes.  Any out pointer may be NULL. */
MK_API int mk_device_info(char *name, int *compute_units, size_t *hbm_bytes);
MK_API int mk_sync(void);

/* ------------------------------------------------------------------ memory ----- */
MK_API int mk_malloc(void **dptr, size_t bytes);
MK_API int mk_free(void *dptr);
MK_API int mk_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
MK_API int mk_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
MK_API int mk_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes);
MK_API int mk_memset(void *dst_dev, int byte, size_t bytes);
/* Vector arena: ONE device allocation of `bytes` from which the solvers created afterwards carve their loop vectors
 * (2 MiB granules) instead of allocating each with hipMalloc; a solver that finds no room falls back to hipMalloc.
 * Meant to be called BEFORE a large matrix is built: how fast the fused update kernels stream depends on where the
 * vectors lie in HBM relative to each other, and a fresh device places them better than a device that
 * already holds 20 GB of matrix.  bytes = 0 releases the arena.  MK_ERR_STATE while vectors of it are in use. */
MK_API int mk_arena_reserve(size_t bytes);

/* Profiling aid: stream `bytes` of device memory with `width` (4, 8 or 16) bytes per lane, reading
 * (write = 0) or writing (write = 1).  Known byte counts to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE. */
MK_API int mk_calib_stream(void *dev, int64_t bytes, int width, int write);

/* Device vectors handed to this library (x, y, rhs, guess, preconditioner diagonals ...) must be 16-byte aligned and
 * readable up to an even number of entries (kernels move vectors in 16-byte pairs); every buffer from mk_malloc is.
 * Misaligned pointers are rejected with MK_ERR_ARG. */

/* ------------------------------------------------------------------ CSR -------- */
/* The device-resident operator behind `linop.LinearOperator`: replaces the user
 * `matvec` callable of pykrylov/linop/linop.py:114,:289 (Pysparse in
 * examples/demo_common.py:15-16).  Canonical CSR: sorted columns, no duplicates. */
typedef struct mk_csr mk_csr;

MK_API int mk_csr_create(int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *indptr_host,
                  const int32_t *indices_host, const double *data_host, mk_csr **out);
MK_API int mk_csr_destroy(mk_csr *A);
MK_API int mk_csr_shape(const mk_csr *A, int64_t *nrows, int64_t *ncols, int64_t *nnz);
/* Smallest and largest stored column index (2147483647 / -1 without entries), found on the device: the range check
 * a binding applies to a matrix built from caller arrays (the reference has none: its operator wraps a callable,
 * linop/linop.py:114; an out-of-range column would read outside x). */
MK_API int mk_csr_col_range(const mk_csr *A, int32_t *min_col, int32_t *max_col);
/* Canonical CSR from coordinate triples (host arrays, borrowed), built on the device: the on-disk side of the
 * path -- MatrixMarket coordinate files (examples/1138bus.mtx, jpwh_991.mtx; examples/demo_common.py:12-16 used
 * Pysparse for this) and the reference's CoordLinearOperator (linop.py:638-685).  Columns sorted per row,
 * duplicate entries summed in input order starting from 0.0 (the result of a stable host sort + np.add.at):
 * integer arrays and values are bit-identical to that host construction.  Symmetric storage must be mirrored by
 * the caller.  MK_ERR_UNSUPPORTED if one row holds more than 16384 entries (use mk_csr_create then). */
MK_API int mk_csr_from_coo(int64_t nrows, int64_t ncols, int64_t nentries, const int32_t *rows_host,
                    const int32_t *cols_host, const double *vals_host, mk_csr **out);
/* Copy the arrays back (any pointer may be NULL). */
MK_API int mk_csr_download(const mk_csr *A, int32_t *indptr_host, int32_t *indices_host, double *data_host);
/* The same for the rows [row_begin, row_end) only: indptr_host receives row_end - row_begin + 1 row pointers AS STORED
 * (offsets into the whole matrix: subtract the first to index the slices), indices_host / data_host the entries
 * indptr[row_begin] .. indptr[row_end).  Call once with NULL arrays for the row pointers to learn the sizes.  This is
 * how a 512^3 matrix (11.8 GB of arrays) is checked against the oracle slab by slab (tests/test_gpu_full_size.py). */
MK_API int mk_csr_download_rows(const mk_csr *A, int64_t row_begin, int64_t row_end, int32_t *indptr_host,
                         int32_t *indices_host, double *data_host);
/* B = A^T as a new canonical CSR built on the device (operator `.T`, linop.py:148-171;
 * feeds `A.T * u` of pykrylov/lls/lsqr.py:200,264). */
MK_API int mk_csr_transpose(const mk_csr *A, mk_csr **out);
/* Synthetic matrices generated directly in HBM.  Rows
 * [row_begin,row_end) of the global matrix, global column ids.
 * poisson2d: 5-point, m x m grid (the matrix of pykrylov/gallery/gallery.py:10-29).
 * poisson3d: 7-point, nx x ny x nz grid, x fastest. */
MK_API int mk_csr_poisson2d(int64_t m, int64_t row_begin, int64_t row_end, mk_csr **out);
MK_API int mk_csr_poisson3d(int64_t nx, int64_t ny, int64_t nz, int64_t row_begin, int64_t row_end, mk_csr **out);
/* poisson3d_varcoef: -div(k grad u) on the same grid with the same sparsity (integer arrays identical to poisson3d),
 * k a positive cell field hashed from `seed`; entries are minus the harmonic means of neighbouring cells, the diagonal
 * their sum plus k itself per missing neighbour (Dirichlet).  SPD, practically all stored values distinct: the workload
 * on which no constant-coefficient compression applies (bench.py `poisson3d-512-varcoef`; the matrix class a user's
 * `matvec` brings to linop/linop.py:271-298). */
MK_API int mk_csr_poisson3d_varcoef(int64_t nx, int64_t ny, int64_t nz, uint64_t seed, int64_t row_begin, int64_t row_end,
                             mk_csr **out);
/* stencil27: the 27-point box stencil on the same grid (HPCG's sparsity; rows of 8 ... 27 entries, columns ascending).
 * seed == 0: -1.0 off the diagonal, 26.0 on it; seed != 0: harmonic means of the hashed cell field as above, the
 * diagonal their left-to-right sum over the 26 directions (k itself per missing neighbour).  SPD.  No reference
 * counterpart beyond "a user's matvec" (linop/linop.py:271-298): the test matrix of the wide storage formats. */
MK_API int mk_csr_stencil27(int64_t nx, int64_t ny, int64_t nz, uint64_t seed, int64_t row_begin, int64_t row_end, mk_csr **out);

/* Operator algebra that stays on the device (linop.py:307-330 `alpha * op`, :375-398 `op + other`, :403-426
 * `op - other`, :400-401 `-op`, with `other` a DiagonalOperator (:473-516), an IdentityOperator (:455-470) or a scalar
 * multiple of one).  The reference evaluates such operators as `alpha * (op * x)`, `(op * x) + (other * x)`, ... one
 * NumPy expression per node; mk_csr_compose makes an operator that shares A's arrays (A must outlive it) and applies
 * the same expressions, in the same order, to every row sum before anything else sees it:
 *     t = (A x)_r ; for each step:   term = x_r ; if diag: term = diag[r] * term ; if has_scale: term = scale * term
 *        MK_ROW_SCALE  t = scale * t      MK_ROW_ADD  t = t + term      MK_ROW_SUB  t = t - term      MK_ROW_RSUB  t = term - t
 * Steps of A itself (if it is a composed operator) run first.  Square matrices only when a step uses x_r.
 * Every product of the library (mk_spmv and all solver kernels) honours the steps. */
enum { MK_ROW_SCALE = 1, MK_ROW_ADD = 2, MK_ROW_SUB = 3, MK_ROW_RSUB = 4, MK_ROWPROG_MAX = 4 };
typedef struct mk_rowop {
    int32_t code;         /* MK_ROW_* */
    int32_t has_scale;    /* term is multiplied by `scale` (always 1 for MK_ROW_SCALE) */
    double scale;
    const double *diag;   /* device array, nrows entries, borrowed; NULL: identity */
} mk_rowop;
MK_API int mk_csr_compose(const mk_csr *A, int32_t nops, const mk_rowop *ops, mk_csr **out);

/* Sum, difference and product of two device matrices as ONE device operator (linop.py:375-398 `op + other`, :403-426
 * `op - other`, :332-354 `op * other`): the reference evaluates them as `(A*x) + (B*x)`, `(A*x) - (B*x)` and `A*(B*x)`
 * -- two complete products and one element-wise operation -- and so does the device: the first product's row sums go
 * to a temporary, the second product's kernel combines them with its own row sums (or multiplies the temporary)
 * and feeds the result to the fused epilogue of whatever solver kernel asked for the product.  Same bits as the
 * reference's closures.  A and B are borrowed (destroying one while the result is alive is deferred until the result is destroyed); they may carry a row program
 * (mk_csr_compose) but must not be composites, matrix-free or partitioned.  sign = +1 / -1. */
MK_API int mk_csr_create_sum(const mk_csr *A, const mk_csr *B, int sign, mk_csr **out);
MK_API int mk_csr_create_product(const mk_csr *A, const mk_csr *B, mk_csr **out);

/* Restriction of a device matrix to the rows `rows` and columns `cols` (host index arrays, copied), as one device
 * operator: reference ReducedLinearOperator / SymmetricallyReducedLinearOperator (linop/linop.py:560-623), evaluated
 * the same way -- `z = 0; z[cols] = x; y = (A z)[rows]` -- with scatter, product and gather on the device.  `cols` must
 * not repeat an index.  A is borrowed (see mk_csr_create_sum). */
MK_API int mk_csr_create_reduced(const mk_csr *A, int64_t nrows, const int32_t *rows, int64_t ncols, const int32_t *cols,
                          mk_csr **out);

/* A grid of device matrices as ONE operator (reference linop/blkop.py:8-152 BlockLinearOperator, :154-257
 * BlockDiagonalLinearOperator): `blocks` lists nbr x nbc handles row by row (NULL = zero block), block (i, j) being
 * heights[i] x widths[j].  A product evaluates every block product completely and adds the results to the block row's
 * accumulator one block at a time, starting from +0.0 -- the reference's `y_i += B_ij * x_j` (blkop.py:86-96) -- so
 * results carry the same roundings, and the operator is accepted wherever a device matrix is (solver product sites
 * with their fused epilogues included).  The blocks are borrowed: destroying one while the block operator is alive is
 * deferred until the block operator is destroyed.  Device matrices with or without a row program; not composites. */
MK_API int mk_csr_create_block(int32_t nbr, int32_t nbc, const mk_csr *const *blocks, const int64_t *heights,
                        const int64_t *widths, mk_csr **out);
/* Matrix-free operator: the products of the returned handle are computed by a HOST callback -- the reference's own
 * operator protocol, `LinearOperator(nargin, nargout, matvec=callable)` (linop/linop.py:114,271-298), e.g. the gallery
 * operators its CG test runs on (cg/tests/test_diagdom.py:38-40).  Everything else of a solver loop (dots, updates,
 * scalar recurrences, stopping tests) still runs on the device: at each product site the loop's input vector is
 * materialised on the device, copied to the host, `fn(user, transpose, x_host, y_host)` is called (x_host: ncols
 * entries, or nrows when transpose != 0; return 0 on success), and the result feeds the same fused epilogue kernel
 * a CSR product would have fed.  The callback is invoked exactly when the reference would have invoked `op * v`
 * (never after the loop condition failed).  Single GPU only. */
typedef int (*mk_matvec_fn)(void *user, int transpose, const double *x_host, double *y_host);
MK_API int mk_csr_create_callback(int64_t nrows, int64_t ncols, mk_matvec_fn fn, void *user, int transpose, mk_csr **out);

/* Storage format the products of A stream from HBM, chosen per matrix and built on the device at the first product
 * (an acceleration structure beside the CSR arrays; results are bit-identical in every format):
 *   0  plain CSR: 4-byte columns + 8-byte values, x gathered through L1/L2;
 *   1  windowed tiles: per 256-row tile the referenced columns are covered by contiguous windows of x that the kernel
 *      stages in LDS with coalesced loads; per nonzero a uint16 LDS slot replaces the column (2 + 8 bytes);
 *   2  format 1 + value dictionary: matrices with <= 256 distinct values store ONE 32-bit word per nonzero
 *      {LDS slot : 16 | dictionary index : 8} and no values (4 bytes); slots, words and windows go to LDS by direct copies;
 *   4  format 2 + row patterns: when the rows of the windowed tiles follow <= 256 distinct sequences of
 *      {LDS slot - lane, dictionary index} words (stencils), ONE BYTE per row names its sequence and nothing is read
 *      per nonzero;
 *   5  format 1's windows + row patterns for matrices WITHOUT a dictionary (variable-coefficient stencils): one byte
 *      per row names its sequence of LDS slots, the values are streamed in tile-sliced ELL order (8 bytes per nonzero);
 *   6, 7, 8  the wide twins for rows of up to 32 entries (tiles of up to 8192 nonzeros in 32 window chunks), chosen when
 *      the cover of formats 1 .. 5 (16 chunks, 2048 nonzeros) reaches less than half of the tiles: 8 = dictionary + row
 *      patterns (one byte per row), 7 = row patterns + streamed values, 6 = uint16 slots + values both streamed in
 *      tile-sliced ELL order (10 bytes per nonzero; needs neither patterns nor a dictionary).  6 or 7 asked for
 *      explicitly are also applied to matrices formats 1 .. 5 would have served;
 *   3  plain CSR for matrices without a window cover whose x is longer than an L2: the tile's stream is held in LDS
 *      and the gathers of all workgroups walk x slice by slice (same arrays as format 0);
 *   9  z-marching bricks for 7-point-class matrices: every column offset in {0, +-1, +-L, +-P}, nrows % P == 0 (the 7-point
 *      stencil of ANY nx x ny x nz grid, L = nx, P = nx ny, any boundary treatment; any band matrix of that shape; a 5-point
 *      matrix with one far stride M as L = 128, P = M) and <= 256 distinct values: ONE BYTE per row names its pattern
 *      {7 values, presence mask}; a workgroup owns a brick of 4 lines x 128 rows and marches through the planes with the
 *      planes z-1, z, z+1 of its own rows in registers, so that every x entry is loaded once per product (format 4
 *      requests each five times).  Round 6: lines that are no multiple of 128 rows, planes that are no multiple of four
 *      lines and odd strides are served by general-geometry kernels whose partly empty bricks discard the rows that do not
 *      exist (mk_csr_march_info); at least half of the bricks' lanes must have rows.  General-geometry kernels exist for
 *      plain products and CG; any other loop on such a matrix runs the CSR gather kernel on the same arrays (same row sums
 *      bit for bit; its fused dots then follow tile order 0 over the march's grid, not the brick order).  Chosen
 *      AUTOMATICALLY from 2^21 rows on (MK_PENCIL_MIN_ROWS) where the bricks are at least 90 % full, from 2^24 rows on
 *      where they are 50 .. 90 % full, for 5-point matrices from 2^23 rows on -- where it was measured to win
 *      (profiles/r06_march_sizes.txt) --, and only while the last solver created on the matrix is CG or none; asked for
 *      explicitly it is applied to any matrix of the class.  Matrices outside the class degrade to 8 and below.
 *  10  format 9's march for matrices of the class WITHOUT a value dictionary (variable-coefficient stencils): the byte per
 *      row is its 7-bit presence mask and the values are streamed from seven position-major arrays (56 bytes per row,
 *      +0.0 where a row has no entry) -- what format 5 streams, with every x entry loaded once.  Chosen automatically
 *      (same size rule) when format 9 finds more than 256 values or patterns; asked for explicitly on any matrix of the class.
 *      Formats 9 .. 11 also serve ONE RANK'S SLAB of such a matrix -- whole planes, columns localised by mk_csr_localize
 *      mode 0, halo exchange -- taking the neighbours' planes from the received entries.
 *  11  format 10 for matrices that are SYMMETRIC bit for bit (what CG runs on): only the diagonal and the three upper values
 *      of a row are stored and streamed (32 bytes per row); the lower ones are the neighbouring rows' upper values, which
 *      the march has in registers (plane below) or passes through an LDS image of the plane's values (row before, line
 *      below).  The builder verifies the symmetry entry by entry; a matrix that is not symmetric stays format 10.
 * fmt = -1 restores the default (environment MK_SPMV_FORMAT, else 11 = the most compact format the matrix qualifies for;
 * format 3 is chosen automatically for scattered matrices with more than 5 MiB of x).  A request is an upper bound and
 * degrades silently (11 -> 10, 10 -> 9 -> 8, 8 -> 7 -> 6 -> 0, 5 -> 1, 4 -> 2 -> 1 -> 0, 3 -> 0): tiles with scattered columns always take the
 * gather path of format 0.
 * mk_csr_format_info reports what is in use: the format, the number of windowed tiles, the LDS chunks (128
 * doubles each) a workgroup reserves (format 3: the number of column phases), the dictionary size and the bytes of
 * matrix data (everything except x and y) one product streams. */
MK_API int mk_csr_set_format(mk_csr *A, int fmt);
MK_API int mk_csr_format_info(const mk_csr *A, int32_t *fmt, int64_t *tiles_windowed, int32_t *lds_chunks,
                       int32_t *dict_size, int64_t *matrix_bytes_per_product);
/* Column blocks: a plain-CSR matrix (format 0) can additionally be stored as K <= 16 column blocks of block_kb KiB of x
 * each (a second copy of the CSR data); its products then run block after block with the running row sums carried from one
 * launch to the next -- the same left-to-right sum, same bits -- every block a resident tile of format 3 with column phases
 * over its own slice of x, which fits an XCD's L2.
 * AUTOMATIC since round 4 (block_kb = -1, the state of a new matrix): on for long-row operators -- >= 12 entries per row on
 * average gathered from an x of >= 16 MB, e.g. the transpose of a tall least-squares matrix -- with 8 MiB blocks (4 MiB if a
 * block's tile outgrows the LDS budget), PROVIDED that x fits 16 blocks, the second copy fits a quarter of the free device
 * memory and every block's tiles are LDS resident; otherwise the matrix keeps its single launch.  The environment
 * variable MK_COLBLOCK_KB overrides the automatic choice (0 = never).  On 5-entry rows the single launch of format 3 is
 * faster, which is why short-row matrices are never blocked automatically.
 * block_kb > 0 forces blocks of that size for A, 0 turns them off.  mk_csr_colblocks reports K (0: not blocked). */
MK_API int mk_csr_set_colblocks(mk_csr *A, int32_t block_kb);
/* Tile order of the SpMV launches (speed only; it also fixes which rows a workgroup's partial sums of a fused dot
 * cover): 0 round robin, 1 each XCD sweeps its own contiguous eighth, 2 every step of the grid is cut into eight
 * XCD-contiguous blocks, 3 stripes of `stripe` tiles dealt round-robin to the XCDs, 4 as 3 but an XCD walks its strip
 * through all planes (`plane` tiles apart) before it takes its next strip; -1 = library default.  `nontemporal`: matrix
 * data that is read once per product, and the product vector of the CG loop, go past the caches (1 / 0 / -1 = default:
 * on when a vector is larger than the 256 MiB Infinity Cache).  mk_csr_tile_order reports what a launch would use now. */
MK_API int mk_csr_set_tile_order(mk_csr *A, int32_t order, int32_t stripe, int32_t plane, int32_t nontemporal);
MK_API int mk_csr_tile_order(const mk_csr *A, int32_t *order, int32_t *stripe, int32_t *plane, int32_t *nontemporal);
MK_API int mk_csr_colblocks(const mk_csr *A, int32_t *nblocks);
/* Launch geometry of A's product kernels (what fixes the summation order of the dots fused into them): the grid,
 * and the tile order (0 round robin; 1 each XCD sweeps a contiguous eighth; 2 XCD-contiguous blocks per step). */
MK_API int mk_csr_launch_info(const mk_csr *A, int32_t *grid, int32_t *tile_map);
/* Geometry of formats 9 / 10 (all zero when A is in another format): the line and plane strides L and P found in the matrix,
 * the number of planes, the planes a workgroup marches through per (brick, chunk) item, the chunks per brick and the
 * number of distinct row patterns.  Workgroup b of a launch of `grid` workgroups takes the items b, b + grid, ...;
 * item i = brick (i % bricks_per_plane) of chunk (i / bricks_per_plane) -- dealt XCD-contiguously when bricks_per_plane and
 * the grid are multiples of 8: brick (i % 8) bpp / 8 + (i / 8) % (bpp / 8) of chunk (i / 8) / (bpp / 8) --,
 * bricks_per_plane = bpp = (L / 128) (P / 4L), brick j
 * starting at row (j / (L / 128)) 4L + (j % (L / 128)) 128 of a plane -- what oracle/gpu_order.py restates for the
 * fused dots (test infrastructure). */
MK_API int mk_csr_pencil_info(const mk_csr *A, int64_t *stride_line, int64_t *stride_plane, int32_t *planes,
                              int32_t *planes_per_chunk, int32_t *chunks, int32_t *patterns);
/* The same and the rest of the brick geometry as an array (round 6: any grid side; all zero when A is in another format):
 *   info[0] storage format (9 / 10 / 11)      [1] L, line stride            [2] P, plane stride         [3] planes
 *   [4] lines per plane = ceil(P / L)          [5] bricks per line = ceil(L / 128)                       [6] brick rows = ceil(lines / 4)
 *   [7] planes per chunk                       [8] chunks                    [9] 2 = general geometry (partly empty bricks, pairs
 *   at any 8-byte boundary: a lane whose row does not exist -- in-line position >= L or in-plane index >= P -- discards its
 *   row sum), 0 = whole aligned bricks          [10] per: bricks per XCD of the XCD-contiguous deal, 0 = round robin
 *   [11] row patterns (format 9).
 * Item i of a launch whose grid is a multiple of 8 with per > 0: brick (i % 8) per + (i / 8) % per of chunk (i / 8) / per
 * (8 per item slots per chunk; a slot whose brick number is >= bricks per plane is empty); otherwise brick i % bpp of
 * chunk i / bpp.  Brick j starts at row (j / bx) 4L + (j % bx) 128 of a plane.  A 5-point matrix (one far stride M) is
 * reported as L = 128, P = M: it is marched line by line.  At most `cap` entries are written (MK_MARCH_INFO_LEN exist). */
#define MK_MARCH_INFO_LEN 12
MK_API int mk_csr_march_info(const mk_csr *A, int64_t *info, int32_t cap);

/* y = A x   (K1; `self.op * p`, pykrylov/cg/cg.py:115 and every other solver).
 * x_dev must be 16-byte aligned and readable ONE ENTRY PAST ITS END (the kernels read pairs; a pair may start at the last
 * entry when a plane stride of the brick-march formats is odd): every buffer from mk_malloc has 16 bytes of slack.
 * Per row the products are added left to right with one rounding per multiply and per
 * add, so the result is bit-identical to a scalar CSR loop. */
MK_API int mk_spmv(const mk_csr *A, const double *x_dev, double *y_dev);

/* ------------------------------------------------------------------ BLAS-1 ----- */
/* K2/K3/K4 of SURVEY.md section 2.2: np.dot / np.linalg.norm / in-place updates of the
 * solver loops (e.g. pykrylov/cg/cg.py:117,130-131,146,150-151).  Deterministic
 * fixed-tree reductions; results returned to the host. */
MK_API int mk_dot(int64_t n, const double *x_dev, const double *y_dev, double *result_host);
MK_API int mk_nrm2(int64_t n, const double *x_dev, double *result_host);
MK_API int mk_axpy(int64_t n, double alpha, const double *x_dev, double *y_dev);               /* y += alpha*x   */
MK_API int mk_axpby(int64_t n, double alpha, const double *x_dev, double beta, double *y_dev); /* y = alpha*x + beta*y */
MK_API int mk_scal(int64_t n, double alpha, double *x_dev);                                    /* x *= alpha     */

/* ------------------------------------------------------------------ comm ------- */
/* Row-partitioned multi-GPU execution (one process per GPU, RCCL over xGMI).
 * unique_id: 128 bytes from mk_comm_unique_id on rank 0, broadcast by the caller. */
MK_API int mk_comm_unique_id(void *id128);
MK_API int mk_comm_init(int nranks, int rank, const void *id128);
/* Host-staged transport: the same collectives carried by caller-supplied functions on HOST buffers (for instance
 * torch.distributed with the gloo backend).  Every collective stages through pinned memory and synchronises the
 * stream: meant for exercising the multi-rank logic where RCCL cannot be used (several ranks on one GPU, CI), not
 * for speed.  Callbacks return 0 on success.
 *   allreduce(buf, count)                       in-place sum over all ranks
 *   exchange(send, send_count, send_off, recv, recv_count, recv_off)
 *                                               rank r gets send[send_off[r] .. +send_count[r]) and delivers
 *                                               recv_count[r] entries to recv + recv_off[r] (arrays of nranks)
 *   allgather(send, count, recv)                recv = concatenation over ranks of `count` entries each */
typedef int (*mk_host_allreduce_fn)(double *buf, int64_t count);
typedef int (*mk_host_exchange_fn)(const double *send, const int64_t *send_count, const int64_t *send_off,
                                   double *recv, const int64_t *recv_count, const int64_t *recv_off);
typedef int (*mk_host_allgather_fn)(const double *send, int64_t count, double *recv);
MK_API int mk_comm_init_host(int nranks, int rank, mk_host_allreduce_fn allreduce, mk_host_exchange_fn exchange,
                      mk_host_allgather_fn allgather);
MK_API int mk_comm_destroy(void);
MK_API int mk_comm_info(int *nranks, int *rank);
/* Which transport carries the collectives: *kind = 0 none, 1 RCCL, 2 host-staged callbacks; *rccl_ranks = what
 * ncclCommCount reports for the communicator (0 without RCCL) -- bench.py prints it so that a host-staged fallback can
 * never be mistaken for an RCCL measurement; *halo_comm_split = 1 if the halo messages have their own communicator. */
MK_API int mk_comm_transport(int *kind, int *rccl_ranks, int *halo_comm_split);
/* Attach an exchange plan to a local matrix whose columns are already remapped to
 * [local rows | halo]: before each SpMV the `send_count[r]` entries `send_idx`
 * (local indices, grouped by destination rank) go to rank r and `recv_count[r]` entries
 * arrive into the halo region in rank order.  mode 0 = halo send/recv, 1 = allgather
 * (then the matrix keeps global column ids and every rank owns `n_local` rows, the
 * last rank possibly fewer). */
MK_API int mk_csr_set_exchange(mk_csr *A, int mode, int64_t n_local, int64_t n_halo, const int64_t *send_count,
                        const int64_t *recv_count, const int32_t *send_idx_host);
/* Rewrite the global column ids of a row block [col_begin, col_end) of a square matrix into the
 * local numbering, in place, on the device.
 * mode 0 (halo): columns outside the owned range must lie in two contiguous windows next to it
 *   (banded / stencil matrices); their widths are returned in halo_lo / halo_hi and the new
 *   numbering is [owned | lower window | upper window], ncols becomes n_local + lo + hi.
 * mode 1 (allgather): col' = n_local + col, ncols becomes n_local + gathered_len. */
MK_API int mk_csr_localize(mk_csr *A, int mode, int64_t col_begin, int64_t col_end, int64_t gathered_len,
                    int64_t *halo_lo, int64_t *halo_hi);
/* x_ext_dev has n_local + n_halo entries: the rank's own slice first, received entries after. */
MK_API int mk_exchange(const mk_csr *A, double *x_ext_dev);
/* Sum `count` (<= 2048) host doubles over all ranks, in place; a no-op without a communicator.  For the few
 * reductions the host side of a partitioned run needs (e.g. tools.check_symmetric, utils.py:63-85). */
MK_API int mk_comm_allreduce_host(double *vals_host, int64_t count);
/* Timing aids (collective: every rank calls them with the same arguments in the same order).  Average duration
 * in microseconds of `reps` back-to-back in-stream exchanges of A's plan (halo: pack + grouped send/recv; all-gather)
 * resp. all-reduces of `count` (<= 2048) doubles, bracketed by one HIP event pair on the library stream. */
MK_API int mk_comm_time_exchange(const mk_csr *A, double *x_ext_dev, int64_t reps, double *avg_us);
MK_API int mk_comm_time_allreduce(int64_t count, int64_t reps, double *avg_us);
/* Duration of the last overlapped halo message group on the second stream (0: none yet). */
MK_API int mk_csr_comm_last_us(const mk_csr *A, double *us);
/* Halo mode overlaps the messages with the product: tiles (256 rows) whose rows reference only owned columns are
 * multiplied while the neighbours' entries travel on a second stream, the remaining tiles afterwards.  Reports the
 * split (0, 0: no overlap plan -- single rank, all-gather mode, or every tile touches the halo). */
MK_API int mk_csr_overlap_info(const mk_csr *A, int64_t *n_interior_tiles, int64_t *n_boundary_tiles);

/* ------------------------------------------------------------------ solvers ---- */
typedef enum {
    MK_CG = 1,        /* pykrylov/cg/cg.py:46-165            */
    MK_BICGSTAB = 2,  /* pykrylov/bicgstab/bicgstab.py:43-151 */
    MK_CGS = 3,       /* pykrylov/cgs/cgs.py:40-123           */
    MK_TFQMR = 4,     /* pykrylov/tfqmr/tfqmr.py:39-159       */
    MK_MINRES = 5,    /* pykrylov/minres/minres.py:115-410    */
    MK_SYMMLQ = 6,    /* pykrylov/symmlq/symmlq.py:65-400     */
    MK_LSQR = 7,      /* pykrylov/lls/lsqr.py:86-453          */
    MK_LSMR = 8,      /* pykrylov/lls/lsmr.py:64-492          */
    MK_CRAIG = 9,     /* pykrylov/lls/craig.py:104-520        */
    MK_CRAIGMR = 10   /* pykrylov/lls/craigmr.py:51-241       */
} mk_solver_kind;

typedef struct {
    int32_t struct_size;      /* = sizeof(mk_params) */
    int32_t kind;             /* mk_solver_kind */
    double abstol;            /* generic.py:74 */
    double reltol;            /* generic.py:75 */
    int64_t matvec_max;       /* cg.py:82 (default 2n is applied by the caller) */
    int32_t check_curvature;  /* cg.py:67 */
    int32_t has_shift;        /* symmlq.py:92-93: shift None vs value */
    double shift;             /* minres.py:122, symmlq.py:92 */
    double rtol;              /* minres.py:126, symmlq.py:90 */
    double etol;              /* minres.py:127 */
    int64_t itnlim;           /* minres.py:125 */
    int32_t window;           /* minres.py:130 */
    int32_t spmv_event_stride; /* >0: bracket the SpMV kernel of every k-th pass with HIP events */
    double damp;              /* lsqr.py:86, lsmr.py:64 */
    double atol;              /* lsqr.py:86 */
    double btol;              /* lsqr.py:86 */
    double conlim;            /* lsqr.py:87 */
} mk_params;

typedef struct {
    int32_t struct_size;
    int32_t halted;           /* the loop condition of the reference became false */
    int64_t nMatvec;
    int64_t itn;
    int64_t hist_len;         /* entries available through mk_solver_history */
    int32_t converged;
    int32_t definite;         /* cg.py:162 */
    int32_t istop;            /* minres.py:87-98, symmlq.py:99-109 */
    int32_t reserved;
    double residNorm;
    double residNorm0;
    double threshold;
    double Anorm, Acond, Arnorm, ynorm, xnorm;
    double aux[8];
} mk_result;

typedef struct mk_solver mk_solver;

MK_API int mk_solver_create(const mk_csr *A, const mk_params *params, mk_solver **out);
MK_API int mk_solver_destroy(mk_solver *s);
/* The least-squares solvers (MK_LSQR ... MK_CRAIGMR) need `A.T * u` (lls/lsqr.py:200,264): hand them the
 * transposed matrix (mk_csr_transpose) before mk_solver_setup.  rhs then has nrows(A) entries and x
 * ncols(A) (CRAIG-MR: nrows(A), craigmr.py:112). */
MK_API int mk_solver_set_transpose(mk_solver *s, const mk_csr *At);
/* Several GPUs: mark `A` as THIS rank's block of consecutive rows of a taller m x n operator (its columns are global
 * and unlocalized; no mk_csr_set_exchange).  The least-squares solvers then keep every m-space vector (rhs, u, r;
 * x of CRAIG-MR) sliced like the rows and every n-space vector (v, w, x) whole on every rank: `A * v` needs no
 * exchange, `A.T * u` (lsqr.py:264) is the local transposed block's product summed over the ranks (one all-reduce
 * of n doubles per iteration) and only the m-space inner products are all-reduced.  `At` is the transpose of the
 * local block.  Needs a communicator (mk_comm_init / mk_comm_init_host); without one the flag changes nothing. */
MK_API int mk_csr_set_row_block(mk_csr *A, int on);
/* Diagonal (Jacobi-type) preconditioner: `diag` is a device array with nrows(A) entries holding the diagonal of
 * the operator the reference applies as `precon * r` (cg.py:91-92,137-138; bicgstab.py:96-99,120-123;
 * cgs.py:79-82,88-91; tfqmr.py:77-80; minres.py:162-163,249; symmlq.py:134,228), i.e. what a
 * linop.DiagonalOperator(diag) (linop.py:473-516) multiplies by.  Borrowed: it must stay alive until the solver
 * is destroyed.  NULL removes it.  Call before mk_solver_setup.  MK_ERR_UNSUPPORTED for the lls kinds. */
MK_API int mk_solver_set_precon_diag(mk_solver *s, const double *diag);
/* General preconditioner: any operator the reference would apply as `precon * r` (generic/generic.py:76), evaluated
 * by a HOST callback `fn(user, r_host, y_host)` (n entries each; return 0 on success).  The loop stays on the device: at
 * each preconditioner site the vector is copied to the host, the callback runs, and the inner product that involves
 * its result is formed on the device afterwards.  Not invoked once the loop has halted; in BiCGSTAB / CGS / TFQMR the
 * application that precedes a product happens before that product's loop test, so the callback may run once more
 * than in the reference (its last result is unused).  The six square solvers; MK_ERR_UNSUPPORTED for the lls kinds
 * and on partitioned operators.  Replaces a diagonal set earlier.  Call before mk_solver_setup. */
typedef int (*mk_precon_fn)(void *user, const double *r_host, double *y_host);
MK_API int mk_solver_set_precon_callback(mk_solver *s, mk_precon_fn fn, void *user);
/* ... or a DEVICE operator: `precon * r` (generic/generic.py:76) evaluated as a product with a device matrix or
 * composite -- a sparse approximate inverse such as the inverted diagonal blocks of block-Jacobi
 * (pykrylov_amd.tools.block_jacobi) -- at the same sites as the callback, without leaving HBM.  M is borrowed, square,
 * of the solver's (local) size, without an exchange plan (on several GPUs: a rank-local preconditioner).  NULL
 * removes it.  Call before mk_solver_setup. */
MK_API int mk_solver_set_precon_csr(mk_solver *s, const mk_csr *M);
/* The least-squares kinds take two preconditioners, applied by the reference as `u = M(Mu)` in the m-space and
 * `v = N(Nv)` in the n-space of the Golub-Kahan process (lls/lsqr.py:189-190,201-202,253-254,265-266 and the same
 * lines of lsmr.py, craig.py, craigmr.py): device arrays with the diagonals of M (nrows(A) entries) and N
 * (ncols(A) entries), either may be NULL; borrowed until the solver is destroyed.  Call before mk_solver_setup. */
MK_API int mk_solver_set_lls_precon(mk_solver *s, const double *diag_m, const double *diag_n);
/* M and / or N as HOST callbacks `fn(user, in_host, out_host)` -- any callable the reference would apply as
 * `u = M(Mu)` (nrows(A) entries) or `v = N(Nv)` (ncols(A) entries); a NULL function leaves that side to
 * mk_solver_set_lls_precon.  The loop stays on the device: the vector is copied to the host right after the kernel that
 * formed it, the callback's result replaces u / v and <u, Mu> / <v, Nv> are re-formed on the device.  Not invoked once
 * the loop has halted, nor for N when beta = 0 (lsqr.py:258).  Single GPU.  Call before mk_solver_setup. */
MK_API int mk_solver_set_lls_precon_callback(mk_solver *s, mk_precon_fn fn_m, void *user_m, mk_precon_fn fn_n, void *user_n);
/* Everything before the `while` loop of the reference's solve().  rhs_dev has n_local
 * entries; guess_dev may be NULL (x0 = 0).  Neither is modified. */
MK_API int mk_solver_setup(mk_solver *s, const double *rhs_dev, const double *guess_dev);
/* Run at most max_iters further passes of the loop body entirely on the device (no
 * host round trip per iteration); stops early when the reference's loop condition
 * fails.  iters_done may be NULL. */
MK_API int mk_solver_iterate(mk_solver *s, int64_t max_iters, int64_t *iters_done);
/* Everything after the loop (e.g. SYMMLQ's CG-point transfer) + result scalars. */
MK_API int mk_solver_finish(mk_solver *s, mk_result *res);
MK_API int mk_solver_x(const mk_solver *s, const double **x_dev);
MK_API int mk_solver_history(const mk_solver *s, double *hist_host, int64_t cap);
/* 1 when the solver runs FUSED passes: CG (cg.py:113-158) on a single-device matrix in storage format 9 applies a pass's
 * `x += alpha p ; p = beta p - r` (cg.py:130,150-151) inside the NEXT pass's product kernel, which loads the rows of p, r
 * and x once for both (one sweep over p less per iteration; every bit of x, p, r and of the history unchanged).  The
 * product kernel then moves 48 bytes per row besides the matrix data (p, r, x in; p, x, A p out) instead of 16.
 * mk_solver_x / mk_solver_vector always hand out the up-to-date iterate (formed into scratch vectors between passes).
 * Environment MK_CG_FUSE=0 turns it off.  Valid after mk_solver_setup. */
MK_API int mk_solver_fused(const mk_solver *s, int32_t *fused);
/* Second per-iteration channel, same length as the history (MINRES: truncated direct-error
 * estimate / energy norm, `dir_errors_window` of minres.py:307-308; NaN while itn <= window). */
MK_API int mk_solver_history2(const mk_solver *s, double *hist_host, int64_t cap);
/* Other device vectors of the loop by solver-specific index (CG: 0 = r, 1 = p, the
 * direction stored as `infiniteDescent`, cg.py:122).  len may be NULL. */
MK_API int mk_solver_vector(const mk_solver *s, int index, const double **v_dev, int64_t *len);
/* Device time of the last mk_solver_iterate call (HIP events on the solver's stream)
 * and the accumulated time/launch count of its SpMV kernel. */
MK_API int mk_solver_timing(const mk_solver *s, double *iterate_ms, double *spmv_ms, int64_t *spmv_launches);
/* Average duration of the solver's fused SpMV kernel: `launches` back-to-back launches of exactly the
 * kernel a loop pass uses, bracketed by ONE pair of HIP events on the solver's stream (a pair around
 * each single launch would add ~3-6 us of marker overhead to a ~20 us kernel).  Destroys the product
 * vector of the current pass: call it after the timed iterations. */
MK_API int mk_solver_time_spmv(mk_solver *s, int64_t launches, double *avg_us);
/* The same for a chosen product of the pass: which = 0 the first (every solver; = mk_solver_time_spmv), 1 the second --
 * BiCGSTAB's `A z` with its three fused dots (bicgstab.py:125), CGS's `A z` with `r -= alpha A z` (cgs.py:96-100),
 * TFQMR's second `A z` (tfqmr.py:145-147), the least-squares solvers' `A.T * u` with the fused v update (lsqr.py:264).
 * Launched without the loop gate.  MK_ERR_ARG if the solver has no such product. */
MK_API int mk_solver_time_product(mk_solver *s, int which, int64_t launches, double *avg_us);
/* One-shot convenience: setup + iterate(until halted) + finish. */
MK_API int mk_solver_solve(mk_solver *s, const double *rhs_dev, const double *guess_dev, mk_result *res);

#ifdef __cplusplus
}
#endif
#endif /* MIKRYLOV_H */
