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
