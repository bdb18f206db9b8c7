/* A plain C caller of libigcn_hip.so — no Python, no torch: the drop-in boundary used the way a C / cgo / JNI binding would.
 * Builds a small CSR matrix, calls igcn_spmm_csr_f32_args with the EIGHT required fields of a zeroed struct (what the reference
 * call site has, /root/reference/model.py:99-102: graph, X, edge values), copies the result back and compares it with a loop on
 * the host.  Also drives the 34-argument positional form, a few refusals, and the scoring half of the boundary
 * (igcn_score_topk_f32 with exclusion lists against a host brute force).  Exit status 0 = all good; prints what failed.
 * Built (gcc -std=c99) and run by tests/test_spmm_gpu.py::test_a_plain_c_program_drives_the_library; the HIP runtime is linked for
 * hipMalloc / hipMemcpy only. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "igcn_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)

int main(void)
{
    const int64_t n_rows = 1000, n_cols = 700;
    const int32_t d = 64;
    if (igcn_abi_version() != IGCN_ABI_VERSION) { printf("ABI %d, header %d\n", igcn_abi_version(), IGCN_ABI_VERSION); return 1; }
    /* a deterministic ragged matrix: row r has (r * 7) % 41 nonzeros, columns ascending */
    int64_t *rowptr = (int64_t *)malloc((n_rows + 1) * sizeof(int64_t));
    rowptr[0] = 0;
    for (int64_t r = 0; r < n_rows; ++r) rowptr[r + 1] = rowptr[r] + (r * 7) % 41;
    const int64_t nnz = rowptr[n_rows];
    int32_t *col = (int32_t *)malloc(nnz * sizeof(int32_t));
    float *val = (float *)malloc(nnz * sizeof(float));
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t p = rowptr[r], j = 0; p < rowptr[r + 1]; ++p, ++j) {
            col[p] = (int32_t)((r * 13 + j * 17) % n_cols);
            val[p] = 0.01f * (float)((r + 3 * j) % 19 - 9);
        }
    /* (columns need not be ascending for the product itself) */
    float *x = (float *)malloc(n_cols * d * sizeof(float)), *y = (float *)malloc(n_rows * d * sizeof(float));
    for (int64_t i = 0; i < n_cols * d; ++i) x[i] = 0.001f * (float)((i * 37) % 201 - 100);

    int64_t *d_rowptr; int32_t *d_col; float *d_val, *d_x, *d_y;
    CHECK_HIP(hipMalloc((void **)&d_rowptr, (n_rows + 1) * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&d_col, nnz * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_val, nnz * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_x, n_cols * d * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_y, n_rows * d * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_rowptr, rowptr, (n_rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_col, col, nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_val, val, nnz * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_x, x, n_cols * d * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(d_y, 0xFF, n_rows * d * sizeof(float)));

    igcn_spmm_args a;
    memset(&a, 0, sizeof a);
    a.struct_size = sizeof a;
    a.rowptr = d_rowptr; a.col = d_col; a.val = d_val;
    a.n_rows = n_rows; a.n_cols = n_cols;
    a.x = d_x; a.y = d_y; a.d = d;
    int rc = igcn_spmm_csr_f32_args(&a, NULL);
    if (rc != IGCN_OK) { printf("igcn_spmm_csr_f32_args: %d (%s)\n", rc, igcn_error_string(rc)); return 1; }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(y, d_y, n_rows * d * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (int64_t r = 0; r < n_rows; ++r)
        for (int32_t j = 0; j < d; ++j) {
            double acc = 0.0;
            for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) acc += (double)val[p] * (double)x[(int64_t)col[p] * d + j];
            const double diff = fabs(acc - (double)y[r * d + j]);
            if (diff > worst) worst = diff;
            if (fabs(acc) > scale) scale = fabs(acc);
        }
    printf("struct call: max |diff| %.3g of %.3g\n", worst, scale);
    if (!(worst <= 1e-4 * scale)) { printf("struct call: result off\n"); return 1; }

    /* the positional form: the same bits */
    float *y2 = (float *)malloc(n_rows * d * sizeof(float)), *d_y2;
    CHECK_HIP(hipMalloc((void **)&d_y2, n_rows * d * sizeof(float)));
    const float *no_adds[1] = {NULL};
    rc = igcn_spmm_csr_f32(d_rowptr, d_col, d_val, d_x, d, d_y2, d, n_rows, n_cols, d, 1.0f, no_adds, 0, 1.0f, NULL, NULL, NULL, 0, NULL, 0,
                           NULL, 256, NULL, 0, 1.0f, NULL, 0, nnz, NULL, NULL, NULL, NULL, NULL, NULL);
    if (rc != IGCN_OK) { printf("igcn_spmm_csr_f32: %d\n", rc); return 1; }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(y2, d_y2, n_rows * d * sizeof(float), hipMemcpyDeviceToHost));
    if (memcmp(y, y2, n_rows * d * sizeof(float)) != 0) { printf("positional and struct form differ\n"); return 1; }

    /* refusals come back as codes, nothing is launched */
    a.y = d_x;
    if (igcn_spmm_csr_f32_args(&a, NULL) != IGCN_E_RANGE) { printf("in-place call not refused\n"); return 1; }
    a.y = d_y; a.struct_size = 8;
    if (igcn_spmm_csr_f32_args(&a, NULL) != IGCN_E_SHAPE) { printf("short struct not refused\n"); return 1; }
    if (igcn_spmm_csr_f32_args(NULL, NULL) != IGCN_E_NULL) { printf("NULL struct not refused\n"); return 1; }
    if (igcn_set_tuning("no_such_knob", 1) != IGCN_E_RANGE) { printf("unknown knob not refused\n"); return 1; }

    /* ---- the scoring half of the boundary: igcn_score_topk_f32 (replaces torch.mm + the exclusion loop + torch.topk of
     * /root/reference/trainer.py:149-163) on integer-valued tables, where fp32 dot products are exact: ids and values must equal a
     * host brute force, ties going to the lower item id ---- */
    const int64_t n_users = 300, n_items = 5000;
    const int32_t k = 10;
    float *u = (float *)malloc(n_users * d * sizeof(float)), *it = (float *)malloc(n_items * d * sizeof(float));
    for (int64_t i = 0; i < n_users * d; ++i) u[i] = (float)((i * 7 + i / d) % 9 - 4);
    for (int64_t i = 0; i < n_items * d; ++i) it[i] = (float)((i * 11 + 3 * (i / d)) % 7 - 3);
    /* every user excludes item (3 u) % n_items and item 17 */
    int64_t *ex_rowptr = (int64_t *)malloc((n_users + 1) * sizeof(int64_t));
    int32_t *ex_col = (int32_t *)malloc(2 * n_users * sizeof(int32_t));
    for (int64_t b = 0; b <= n_users; ++b) ex_rowptr[b] = 2 * b;
    for (int64_t b = 0; b < n_users; ++b) {
        int32_t e0 = (int32_t)((3 * b) % n_items), e1 = 17;
        if (e0 == e1) e0 = 18;
        ex_col[2 * b] = e0 < e1 ? e0 : e1;                 /* ascending inside a row */
        ex_col[2 * b + 1] = e0 < e1 ? e1 : e0;
    }
    float *d_u, *d_it, *d_val_out; int64_t *d_idx, *d_ex_rowptr; int32_t *d_ex_col; void *d_ws;
    const int64_t ws_bytes = igcn_score_topk_workspace_bytes(n_users, n_items, d, k);
    if (ws_bytes < 0) { printf("workspace query refused\n"); return 1; }
    CHECK_HIP(hipMalloc((void **)&d_u, n_users * d * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_it, n_items * d * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_idx, n_users * k * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&d_val_out, n_users * k * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_ex_rowptr, (n_users + 1) * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&d_ex_col, 2 * n_users * sizeof(int32_t)));
    CHECK_HIP(hipMalloc(&d_ws, (size_t)(ws_bytes > 0 ? ws_bytes : 8)));
    CHECK_HIP(hipMemcpy(d_u, u, n_users * d * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_it, it, n_items * d * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_ex_rowptr, ex_rowptr, (n_users + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_ex_col, ex_col, 2 * n_users * sizeof(int32_t), hipMemcpyHostToDevice));
    rc = igcn_score_topk_f32(d_u, d, NULL, n_users, d_it, d, n_items, d, d_ex_rowptr, d_ex_col, NULL, k, d_idx, d_val_out, d_ws, NULL);
    if (rc != IGCN_OK) { printf("igcn_score_topk_f32: %d (%s)\n", rc, igcn_error_string(rc)); return 1; }
    CHECK_HIP(hipDeviceSynchronize());
    int64_t *idx = (int64_t *)malloc(n_users * k * sizeof(int64_t));
    float *best = (float *)malloc(n_users * k * sizeof(float)), *score = (float *)malloc(n_items * sizeof(float));
    CHECK_HIP(hipMemcpy(idx, d_idx, n_users * k * sizeof(int64_t), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(best, d_val_out, n_users * k * sizeof(float), hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < n_users; ++b) {
        for (int64_t i = 0; i < n_items; ++i) {
            float acc = 0.f;
            for (int32_t j = 0; j < d; ++j) acc += u[b * d + j] * it[i * d + j];      /* small integers: exact in any order */
            score[i] = acc;
        }
        score[ex_col[2 * b]] = -INFINITY;
        score[ex_col[2 * b + 1]] = -INFINITY;
        for (int32_t r = 0; r < k; ++r) {                   /* selection by (score, lower id) */
            int64_t arg = 0;
            for (int64_t i = 1; i < n_items; ++i) if (score[i] > score[arg]) arg = i;
            if (idx[b * k + r] != arg || best[b * k + r] != score[arg]) {
                printf("top-k: user %ld rank %d: got item %ld (%.1f), want %ld (%.1f)\n", (long)b, (int)r, (long)idx[b * k + r],
                       best[b * k + r], (long)arg, score[arg]);
                return 1;
            }
            score[arg] = -INFINITY;
        }
    }
    printf("top-k call: %ld users x %ld items, k = %d: ids and values equal the host's\n", (long)n_users, (long)n_items, (int)k);
    if (igcn_score_topk_workspace_bytes(n_users, n_items, 66, k) != -1) { printf("d = 66 not refused\n"); return 1; }
    printf("ok\n");
    return 0;
}
