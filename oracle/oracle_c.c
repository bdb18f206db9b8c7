/* CPU restatement in plain C of the two heavy loops of the path, used ONLY as
 * the checker / CPU baseline (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  Not part of the product; the product never links this.
 *
 *  - oracle_spmm_csr_f32:  Y = M @ X — the restated gspmm('mul','sum') of
 *    model.py:102 / :430 / :442 on the coalesced COO of utils.py:32-38, in CSR
 *    row order (PARITY UNPINNED at the DGL boundary, see oracle/oracle.py).
 *  - oracle_propagate_mean_f32: model.py:101-105 (K products, mean of K+1 layers).
 *  - oracle_score_topk_f32: model.py:120-122 + trainer.py:149-163
 *    (dense dot, -inf masks, top-k; ties -> lower id).
 * OpenMP over rows / users; `threads` <= 0 means all cores.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                         const float *x, float *y, int64_t n_rows, int32_t d, int threads)
{
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rows; ++r) {
        float *yr = y + r * d;
        for (int j = 0; j < d; ++j) yr[j] = 0.f;
        for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
            const float w = val ? val[p] : 1.f;
            const float *xr = x + (int64_t)col[p] * d;
            for (int j = 0; j < d; ++j) yr[j] += w * xr[j];
        }
    }
}

/* out = mean(X0, A X0, ..., A^K X0); work: 2 * n_rows * d floats */
void oracle_propagate_mean_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                               const float *x0, float *out, float *work, int64_t n_rows, int32_t d,
                               int32_t n_layers, int threads)
{
    const int64_t n = n_rows * d;
    float *a = work, *b = work + n;
    memcpy(out, x0, (size_t)n * sizeof(float));
    const float *cur = x0;
    for (int l = 0; l < n_layers; ++l) {
        float *nxt = (l & 1) ? b : a;
        oracle_spmm_csr_f32(rowptr, col, val, cur, nxt, n_rows, d, threads);
#pragma omp parallel for
        for (int64_t i = 0; i < n; ++i) out[i] += nxt[i];
        cur = nxt;
    }
    const float s = 1.f / (float)(n_layers + 1);
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) out[i] *= s;
}

void oracle_score_topk_f32(const float *user_rows, const int64_t *user_ids, int64_t batch,
                           const float *item_rows, int64_t n_items, int32_t d,
                           const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                           int32_t k, int64_t *out_idx, float *out_val, int threads)
{
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        float *s = (float *)malloc((size_t)n_items * sizeof(float));
#pragma omp for schedule(dynamic, 8)
        for (int64_t b = 0; b < batch; ++b) {
            const int64_t u = user_ids ? user_ids[b] : b;
            const float *ur = user_rows + u * d;
            for (int64_t i = 0; i < n_items; ++i) {
                const float *ir = item_rows + i * d;
                float acc = 0.f;
                for (int j = 0; j < d; ++j) acc += ur[j] * ir[j];
                s[i] = (banned && banned[i]) ? -INFINITY : acc;
            }
            if (excl_rowptr)
                for (int64_t p = excl_rowptr[u]; p < excl_rowptr[u + 1]; ++p) s[excl_col[p]] = -INFINITY;
            /* k rounds of selection keep the code obviously correct; k is ~20 */
            for (int r = 0; r < k; ++r) {
                int64_t best = -1;
                for (int64_t i = 0; i < n_items; ++i) {
                    if (isnan(s[i])) continue;
                    if (best < 0 || s[i] > s[best]) best = i;
                }
                out_idx[b * k + r] = best;
                out_val[b * k + r] = best >= 0 ? s[best] : -INFINITY;
                if (best >= 0) s[best] = NAN;          /* taken */
            }
        }
        free(s);
    }
}
