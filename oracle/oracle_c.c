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

/* rows [lo, hi) of thread t of T when the rows are cut into T chunks of (nearly) equal nonzeros + rows */
static void balanced_chunk(const int64_t *rowptr, int64_t n_rows, int t, int T, int64_t *lo, int64_t *hi)
{
    const int64_t total = rowptr[n_rows] + n_rows;
    int64_t b[2];
    for (int k = 0; k < 2; ++k) {
        const int64_t target = (int64_t)((double)total * (t + k) / T);
        int64_t l = 0, h = n_rows;                       /* first row r with rowptr[r] + r >= target */
        while (l < h) {
            const int64_t m = (l + h) / 2;
            if (rowptr[m] + m < target) l = m + 1; else h = m;
        }
        b[k] = l;
    }
    *lo = b[0];
    *hi = (t == T - 1) ? n_rows : b[1];
}

static inline void spmm_rows(const int64_t *rowptr, const int32_t *col, const float *val, const float *restrict x,
                             float *restrict y, int64_t lo, int64_t hi, int32_t d)
{
    float acc[256];
    for (int64_t r = lo; r < hi; ++r) {
        for (int j = 0; j < d; ++j) acc[j] = 0.f;
        for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
            const float w = val ? val[p] : 1.f;
            const float *restrict xr = x + (int64_t)col[p] * d;
#pragma omp simd
            for (int j = 0; j < d; ++j) acc[j] += w * xr[j];
        }
        float *restrict yr = y + r * d;
        for (int j = 0; j < d; ++j) yr[j] = acc[j];
    }
}

/* Y = M @ X.  Each row is summed in storage order into a local accumulator (same arithmetic as the plain loop); the
 * rows are cut into one contiguous chunk per thread with equal nonzeros + rows (a power-law graph deals very different
 * work to equal row counts), so a thread always writes — and first touches — the same part of every output buffer. */
void oracle_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                         const float *x, float *y, int64_t n_rows, int32_t d, int threads)
{
    if (d > 256) return;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
    {
        int64_t lo, hi;
        balanced_chunk(rowptr, n_rows, omp_get_thread_num(), omp_get_num_threads(), &lo, &hi);
        spmm_rows(rowptr, col, val, x, y, lo, hi, d);
    }
#else
    spmm_rows(rowptr, col, val, x, y, 0, n_rows, d);
#endif
}

/* out = mean(X0, A X0, ..., A^K X0); work: 2 * n_rows * d floats */
void oracle_propagate_mean_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                               const float *x0, float *out, float *work, int64_t n_rows, int32_t d,
                               int32_t n_layers, int threads)
{
    const int64_t n = n_rows * d;
    float *a = work, *b = work + n;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = x0[i];
    const float *cur = x0;
    for (int l = 0; l < n_layers; ++l) {
        float *nxt = (l & 1) ? b : a;
        oracle_spmm_csr_f32(rowptr, col, val, cur, nxt, n_rows, d, threads);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) out[i] += nxt[i];
        cur = nxt;
    }
    const float s = 1.f / (float)(n_layers + 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] *= s;
}

/* model.py:120-122 + trainer.py:149-163 for `batch` users: dense dots (sequential float sum over d, as a plain loop),
 * -inf masks, the k best by (score desc, id asc).  Users are scored 8 at a time against each item row (the item table
 * is streamed once per 8 users instead of once per user); the k best are kept in a sorted list per user. */
void oracle_score_topk_f32(const float *user_rows, const int64_t *user_ids, int64_t batch,
                           const float *item_rows, int64_t n_items, int32_t d,
                           const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                           int32_t k, int64_t *out_idx, float *out_val, int threads)
{
    enum { UB = 8 };
    if (d > 256) return;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        float *s = (float *)malloc((size_t)n_items * UB * sizeof(float));
#pragma omp for schedule(dynamic, 1)
        for (int64_t b0 = 0; b0 < batch; b0 += UB) {
            const int nb = (int)((batch - b0) < UB ? (batch - b0) : UB);
            float ut[256 * UB];                                   /* the 8 user rows, transposed: [d][8] */
            for (int q = 0; q < UB; ++q) {
                const int64_t bq = b0 + (q < nb ? q : 0);
                const float *u_row = user_rows + (user_ids ? user_ids[bq] : bq) * d;
                for (int j = 0; j < d; ++j) ut[j * UB + q] = u_row[j];
            }
            for (int64_t i = 0; i < n_items; ++i) {
                const float *restrict ir = item_rows + i * d;
                float acc[UB] = {0};
                for (int j = 0; j < d; ++j) {
                    const float v = ir[j];
#pragma omp simd
                    for (int q = 0; q < UB; ++q) acc[q] += ut[j * UB + q] * v;
                }
                const int ban = banned && banned[i];
                for (int q = 0; q < UB; ++q) s[(size_t)q * n_items + i] = ban ? -INFINITY : acc[q];
            }
            for (int q = 0; q < nb; ++q) {
                const int64_t b = b0 + q;
                const int64_t u = user_ids ? user_ids[b] : b;
                float *sq = s + (size_t)q * n_items;
                if (excl_rowptr)
                    for (int64_t p = excl_rowptr[u]; p < excl_rowptr[u + 1]; ++p) sq[excl_col[p]] = -INFINITY;
                int64_t *bi = out_idx + b * k;
                float *bv = out_val + b * k;
                int have = 0;
                for (int64_t i = 0; i < n_items; ++i) {
                    const float v = sq[i];
                    if (have == k && !(v > bv[k - 1])) continue;         /* ties keep the lower id (seen first) */
                    int pos = have < k ? have : k - 1;
                    while (pos > 0 && v > bv[pos - 1]) { bv[pos] = bv[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                    bv[pos] = v; bi[pos] = i;
                    if (have < k) ++have;
                }
                for (int r = have; r < k; ++r) { bi[r] = -1; bv[r] = -INFINITY; }
            }
        }
        free(s);
    }
}
