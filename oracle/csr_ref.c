/* CPU oracle: CSR sparse matrix-vector product.   TEST INFRASTRUCTURE ONLY.
 *
 * The reference has no CSR product of its own: a LinearOperator wraps a user
 * callable (reference pykrylov/linop/linop.py:114, :289) and the examples get
 * theirs from Pysparse (examples/demo_common.py:15-16, not vendored).  The
 * golden fixtures were produced with SciPy 1.15.3's csr_matvec (third party,
 * scipy/sparse/sparsetools/csr.h, not under /root/reference), whose published
 * algorithm is restated here: per row, start from the current y[i] (zero),
 * add data[j] * x[indices[j]] left to right, one rounding per multiply and one
 * per add (no FMA contraction: build with -ffp-contract=off).
 *
 * Pinned by tests/test_oracle_golden.py against products captured from SciPy.
 */
#include <stdint.h>

void ref_csr_matvec(int64_t nrows, const int32_t *indptr, const int32_t *indices,
                    const double *data, const double *x, double *y)
{
    /* Rows are independent: with OpenMP (OMP_NUM_THREADS > 1; bench.py's all-core baseline) they are split across
     * threads, each row still summed left to right -- the result does not change by a bit. */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nrows; ++i) {
        double sum = 0.0;
        for (int32_t j = indptr[i]; j < indptr[i + 1]; ++j)
            sum += data[j] * x[indices[j]];
        y[i] = sum;
    }
}

/* y = A^T x computed the way the oracle operator does it: through an explicit
 * CSR copy of A^T (so the summation runs over ascending original row index). */
void ref_csr_transpose(int64_t nrows, int64_t ncols, const int32_t *indptr, const int32_t *indices,
                       const double *data, int32_t *t_indptr, int32_t *t_indices, double *t_data)
{
    for (int64_t c = 0; c <= ncols; ++c) t_indptr[c] = 0;
    for (int64_t j = 0; j < indptr[nrows]; ++j) t_indptr[indices[j] + 1]++;
    for (int64_t c = 0; c < ncols; ++c) t_indptr[c + 1] += t_indptr[c];
    for (int64_t i = 0; i < nrows; ++i)
        for (int32_t j = indptr[i]; j < indptr[i + 1]; ++j) {
            int32_t c = indices[j];
            int32_t dst = t_indptr[c]++;
            t_indices[dst] = (int32_t)i;
            t_data[dst] = data[j];
        }
    for (int64_t c = ncols; c > 0; --c) t_indptr[c] = t_indptr[c - 1];
    t_indptr[0] = 0;
}

/* Generator twin of oracle/csr_ref.py poisson3d_varcoef (and of the device's gen_poisson3d_varcoef): rows
 * [row_begin, row_end) of -div(k grad u) on an mx x my x mz grid, x fastest, global column ids, written straight into
 * caller-provided CSR arrays.  Exists so that bench.py's cpu_baseline can be MEASURED at the full 512^3 size (the
 * NumPy twin sorts COO triples and needs several times the matrix in scratch memory).  Pinned bit for bit against the
 * NumPy twin by tests/test_oracle_golden.py.  Same operation order: k(c) = 0.5 + (splitmix64(c) >> 11) * 2^-53, face
 * term ((2 ka) kb) / (ka + kb) or ka where the neighbour is missing, diagonal = left-to-right sum over
 * (-z, -y, -x, +x, +y, +z), off-diagonals = -term, columns ascending. */
static inline double ref_cell_field(uint64_t c, uint64_t seed)
{
    uint64_t z = (c + 1u) * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return 0.5 + (double)(z >> 11) * 0x1.0p-53;
}

/* pass 1: row pointers (indptr has row_end - row_begin + 1 entries, rebased to 0); returns the number of entries */
int64_t ref_poisson3d_indptr(int64_t mx, int64_t my, int64_t mz, int64_t row_begin, int64_t row_end, int32_t *indptr)
{
    int64_t at = 0;
    indptr[0] = 0;
    for (int64_t i = row_begin; i < row_end; ++i) {
        const int64_t gx = i % mx, gy = (i / mx) % my, gz = i / (mx * my);
        at += 1 + (gz > 0) + (gy > 0) + (gx > 0) + (gx < mx - 1) + (gy < my - 1) + (gz < mz - 1);
        indptr[i - row_begin + 1] = (int32_t)at;
    }
    return at;
}

void ref_poisson3d_varcoef_fill(int64_t mx, int64_t my, int64_t mz, uint64_t seed, int64_t row_begin, int64_t row_end,
                                const int32_t *indptr, int32_t *indices, double *data)
{
    const int64_t off[6] = {-mx * my, -mx, -1, 1, mx, mx * my};
#pragma omp parallel for schedule(static)
    for (int64_t i = row_begin; i < row_end; ++i) {
        const int64_t gx = i % mx, gy = (i / mx) % my, gz = i / (mx * my);
        const int ok[6] = {gz > 0, gy > 0, gx > 0, gx < mx - 1, gy < my - 1, gz < mz - 1};
        const double kc = ref_cell_field((uint64_t)i, seed);
        double term[6];
        for (int d = 0; d < 6; ++d) {
            if (ok[d]) {
                const double kb = ref_cell_field((uint64_t)(i + off[d]), seed);
                term[d] = ((2.0 * kc) * kb) / (kc + kb);
            } else {
                term[d] = kc;
            }
        }
        const double diag = ((((term[0] + term[1]) + term[2]) + term[3]) + term[4]) + term[5];
        int64_t p = indptr[i - row_begin];
        for (int d = 0; d < 3; ++d)
            if (ok[d]) { indices[p] = (int32_t)(i + off[d]); data[p] = -term[d]; ++p; }
        indices[p] = (int32_t)i; data[p] = diag; ++p;
        for (int d = 3; d < 6; ++d)
            if (ok[d]) { indices[p] = (int32_t)(i + off[d]); data[p] = -term[d]; ++p; }
    }
}

/* The constant-coefficient 7-point Laplacian (diagonal 6, off-diagonals -1; BASELINE configs[4]) written the same way:
 * twin of csr_ref.py poisson3d, pinned bit for bit by tests/test_oracle_golden.py.  Row pointers: ref_poisson3d_indptr. */
void ref_poisson3d_const_fill(int64_t mx, int64_t my, int64_t mz, int64_t row_begin, int64_t row_end,
                              const int32_t *indptr, int32_t *indices, double *data)
{
    const int64_t off[6] = {-mx * my, -mx, -1, 1, mx, mx * my};
#pragma omp parallel for schedule(static)
    for (int64_t i = row_begin; i < row_end; ++i) {
        const int64_t gx = i % mx, gy = (i / mx) % my, gz = i / (mx * my);
        const int ok[6] = {gz > 0, gy > 0, gx > 0, gx < mx - 1, gy < my - 1, gz < mz - 1};
        int64_t p = indptr[i - row_begin];
        for (int d = 0; d < 3; ++d)
            if (ok[d]) { indices[p] = (int32_t)(i + off[d]); data[p] = -1.0; ++p; }
        indices[p] = (int32_t)i; data[p] = 6.0; ++p;
        for (int d = 3; d < 6; ++d)
            if (ok[d]) { indices[p] = (int32_t)(i + off[d]); data[p] = -1.0; ++p; }
    }
}
