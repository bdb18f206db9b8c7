/*
 * igcn_hip.h — C ABI of libigcn_hip.so: the MI355X (gfx950) kernels behind the
 * INMO / LightGCN propagation + scoring path.
 *
 * The reference (WuYunfan/igcn_cf) has no FFI: its hot path reaches the device
 * through three library calls.  Each entry point below names the reference
 * call site it replaces (file:line in the reference tree).
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer unless the name ends in _host;
 *   - the caller owns all memory; the library allocates and frees nothing;
 *   - all launches go to the caller's `stream` (a hipStream_t passed as void*,
 *     NULL = the default stream); no call synchronises;
 *   - every function returns 0 on success, a negative IGCN_E_* on a bad
 *     argument, or a positive hipError_t from the launch.
 */
#ifndef IGCN_HIP_H
#define IGCN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGCN_ABI_VERSION 10

#define IGCN_OK            0
#define IGCN_E_NULL       -1   /* a required pointer is NULL               */
#define IGCN_E_SHAPE      -2   /* a size / leading dimension is invalid    */
#define IGCN_E_ALIGN      -3   /* a pointer / stride misses its alignment  */
#define IGCN_E_RANGE      -4   /* a scalar argument is out of range        */
#define IGCN_E_NO_DEVICE  -5   /* no HIP device is available               */
#define IGCN_E_CAPTURE    -6   /* the stream is capturing and the call cannot be captured: igcn_csr_transpose (rocPRIM's sort),
                                  or a kernel that carries scratch (none does in the library as built; csrc/common.h) */

#define IGCN_MAX_ADDS      8   /* epilogue addends of igcn_spmm_csr_f32     */
#define IGCN_SPMM_MASKED_ROWS_ZERO  1   /* flags of igcn_spmm_csr_f32: rows with row_mask[r] == 0 are set to zero (else left untouched) */
#define IGCN_SPMM_CLOSING_SEGMENTS  2   /* ... the plan marks a closing segment per cut row and row_order deals it late (ABI v8, below) */
#define IGCN_MAX_TOPK    256   /* k of igcn_score_topk_f32                  */
#define IGCN_FAST_FALLBACK_MAX 256   /* flagged users igcn_score_topk_fast_f32 finishes by itself (ABI v7) */

int         igcn_abi_version(void);
const char *igcn_error_string(int code);

/* Developer / test knobs of the launch heuristics (no reference counterpart).  name: "spmm_blocks_per_cu",
 * "spmm_multirow", "topk_slots", "topk_waves_per_cu", "topk_cap", "topk_stagger", "topk_fast_order", "topk_fast_exit",
 * "topk_fast_wide", "topk_fast_extra" (candidates kept beyond k), "topk_fast_give_up" (0: no wave hands users over),
 * "topk_fast_narrow" (0: small batches keep 64-user wave-groups), "topk_fast_share" (0: the pieces of a cut sweep
 * keep their thresholds to themselves), "topk_fast_fallback" (0: igcn_score_topk_fast_f32 leaves every flagged user to the
 * caller), "topk_fast_early_checks" (0: exit checks every 24 tiles only, give-up from tile 48), "topk_fast_pieces" (0: the narrow bounded sweep in at most 58 pieces per group; default up to 232), "topk_fast_filter" (0: every flagged user of igcn_score_topk_fast_f32 takes the bounded fp32 sweep; default: users with a tight
 * bound take a streaming filter over all items first), "topk_fast_warm" (tiles of the
 * candidate sweep's warm-up pass, 0: none; default 128, taken where the item rows at its end are still half as long as the first), "topk_fast_mode" (candidate sweep of
 * igcn_score_topk_fast_f32: 3 = one fp16 plane each side, the default; 2 = two fp16 user planes; 1 = two bf16 planes each side,
 * d = 64 only), "topk_fast_poison" (TEST ONLY, 1: igcn_score_topk_fast_f32 does not clear its order build's counting bins — the
 * build must notice and take the id order); value < 0 restores the library
 * default.  Results never depend on them (tests/test_spmm_gpu.py::test_launch_shape_does_not_change_results).
 * Process-wide: not to be changed while another thread launches (a launch reads its knobs once, at entry; the SpMM takes
 * its three per call instead — igcn_spmm_args.tune_*).  Returns IGCN_E_RANGE for an unknown name. */
int igcn_set_tuning(const char *name, int32_t value);

/* One piece of a long CSR row (a "row segment"): nonzeros [start, start+len)
 * of row `row`, whose partial sum goes to partial[slot]; slot == the segment's index in the array.
 * long_index (ABI v8; was a reserved word): bits 0..30 = the row's entry in the igcn_long_row array; bit 31 = this is
 * the row's CLOSING segment (igcn_spmm_plan_fill_host marks the row's last one).  With IGCN_SPMM_CLOSING_SEGMENTS the
 * launch adds a cut row up itself: every segment counts itself in on the row's arrival counter, the closing segment —
 * which row_order must deal AFTER the row's other segments, the later the better (graph.py puts it a quarter of the way
 * into the rows of the same phase: CLOSING_AT) — waits until the others have arrived, adds the partial sums in slot order and applies the
 * epilogue — when igcn_set_tuning("spmm_fold", 1) asks for it (measured: a gain of 1-4 % with a few thousand segments, a loss
 * of 5 % with the XCD plan's 35 000 on the headline graph; the default is the second small kernel either way; same bits).
 * Built once per graph by igcn_spmm_plan_fill_host. */
typedef struct igcn_row_segment {
    int64_t start;
    int32_t len;
    int32_t slot;
    int32_t row;
    int32_t long_index;
} igcn_row_segment;

/* One long row: its partial sums are partial[first_slot .. first_slot+n_slots), slots in the order of the row's nonzeros. */
typedef struct igcn_long_row {
    int32_t row;
    int32_t first_slot;
    int32_t n_slots;
    int32_t reserved;
} igcn_long_row;

/* Host-side schedule for rows longer than `long_threshold` nonzeros (power-law
 * item rows): each is cut into segments of at most `segment_len` nonzeros.
 * Replaces the per-call graph object the reference rebuilds in
 * model.py:99-100 / :428-429 / :439-440 (dgl.graph((column,row))).
 * _count returns the sizes; _fill writes the two arrays (host memory). */
int igcn_spmm_plan_count_host(const int64_t *rowptr_host, int64_t n_rows,
                              int32_t long_threshold, int32_t segment_len,
                              int64_t *n_long_rows, int64_t *n_segments);
int igcn_spmm_plan_fill_host(const int64_t *rowptr_host, int64_t n_rows,
                             int32_t long_threshold, int32_t segment_len,
                             igcn_long_row *long_rows_host, int64_t n_long_rows,
                             igcn_row_segment *segments_host, int64_t n_segments);

/* Y = epilogue( M @ X ) for a CSR matrix M (n_rows x n_cols), X [n_cols, d]
 * fp32 row-major with leading dimension ldx, Y [n_rows, d] with ldy.
 *
 *   acc[r]  = sum_{p in rowptr[r]..rowptr[r+1]}  w_p * X[col[p]]
 *   w_p     = (val ? val[p] : 1) * (col_scale ? col_scale[col[p]] : 1) * drop_p
 *   drop_p  = keep_prob >= 1 ? 1
 *           : (hash(seed, edge_id ? edge_id[p] : p) keeps) ? 1/keep_prob : 0
 *   Y[r]    = (out_scale * acc[r] + add_scale * sum_i adds[i][r])
 *             * (row_scale ? row_scale[r] : 1)
 *   rows with row_mask[r] == 0 (row_mask != NULL) are not computed: Y[r] is left
 *   untouched or, with IGCN_SPMM_MASKED_ROWS_ZERO in `flags`, set to zero.
 *
 * Replaces dgl.ops.gspmm(g,'mul','sum',X,w) at model.py:102, :430, :442, the
 * layer mean of model.py:104-105 / :444-445 (adds + scales: the callers evaluate
 * the mean as a factored polynomial, one addend table per launch — ops.mean_plan),
 * NGCF.dropout_sp_mat as used at model.py:435 (drop_p, no structure rebuild)
 * and the row-constant values of IGCN.update_feat_mat, model.py:374-377
 * (val == NULL, row_scale = row_sum^exponent; the transposed view used by the
 * backward pass takes the same vector as col_scale).
 *
 * rowptr int64 [n_rows+1]; col int32 [nnz] (may be NULL when nnz == 0); val fp32 [nnz] or NULL;
 * adds_host: HOST array of n_adds device pointers, each [n_rows, d] with ldy;
 * long rows (may be NULL / 0 when the matrix has none): long_rows / segments
 * as produced by igcn_spmm_plan_fill_host, copied to the device, and `partial`
 * a device workspace of n_segments * d floats — with IGCN_SPMM_CLOSING_SEGMENTS followed by n_long_rows * 128 bytes
 * of arrival counters (one int32 per cut row, 128 bytes apart: agent-scope atomics on one line are serialised), ZERO when
 * the buffer is first handed to the library and zero again when a launch has finished;
 * edge_id int32 [nnz] or NULL (used when M is a transposed view, so that both
 * views drop the same edges);
 * row_mask uint8 [n_rows] or NULL: the rows whose output the caller needs (a
 * training step needs the propagated rows of its batch only, so the last layer
 * and the layer before it — and by symmetry the first backward hop — shrink to
 * the batch rows / their neighbourhood; see igcn_mark_rows);
 * nnz = rowptr[n_rows] as the caller knows it (it sizes the launch: the heavier the rows, the fewer
 * of them a wave is given; <= 0 = unknown, a mean degree of 21 is assumed);
 * row_order int32 [n_rows + n_segments] or NULL: the order in which rows (entries < n_rows) and row segments
 * (n_rows + segment index) are dealt to the waves — a permutation of 0..n_rows+n_segments-1, e.g. per phase of a
 * bipartite matrix the segments of its long rows first, then its rows by descending length, so that heavy work
 * starts first and the rows a wave works on together carry equal work; it changes which wave computes what, not
 * the result;
 * col_mask: NULL, or one BIT per column (uint32 words, bit c & 31 of word c >> 5; igcn_pack_mask_bits): rows of X the
 * caller knows to be all zero (bit clear) are not read —
 * the first backward hops of a training step, whose operand is non-zero on the batch rows / their neighbourhood
 * only.  Edges whose weight comes out zero (masked here, or dropped out) issue no gather at all;
 * order_bits (ABI v9): NULL, or row_mask once more as one BIT per entry of the DEALING order (bit vv of word vv >> 5 set when the
 * row of entry vv — row_order[vv], or the row of the segment it names; vv itself without a row_order — has row_mask != 0),
 * 2 * ceil((n_rows + n_segments) / 64) + 2 words, written by igcn_pack_mask_bits_ordered.  Needs row_mask.  A launch that
 * need not zero its masked rows (no IGCN_SPMM_MASKED_ROWS_ZERO), drops nothing out and folds nothing then reads the bits of 64
 * visits at a time and visits only the wanted entries: a masked-out entry otherwise costs its wave a chain of dependent loads
 * (row_order -> row_mask / rowptr): 8 of the 38 us of the last forward launch of a training step (~6 000 of 206 151 rows
 * on the Amazon-book-like graph).  Ignored where it does not apply; same result with and without;
 * seed_dev: NULL, or the dropout seed in device memory (overrides `seed`): a launch captured in a HIP graph reads it
 * at every replay, so the caller changes the dropped edges by writing 8 bytes, not by re-capturing;
 * xcd_off (ABI v5): NULL, or int64 [9] in device memory: row_order then holds EIGHT lists back to back, list x =
 * row_order[xcd_off[x] .. xcd_off[x + 1]), xcd_off[0] = 0, and workgroup b walks list b % 8 only (workgroups b and b + 8
 * share an XCD and its 4 MiB L2 under the hardware's round-robin placement).  Every row that is not cut into segments
 * and every row segment appears in exactly one list (rows that are cut may be left out).  The caller groups into one
 * list the rows / segments that gather from the same slice of X, so that the slice stays in that XCD's L2 — there is
 * no counterpart in the reference (DGL's CPU gspmm walks rows in order); results do not depend on it. */
int igcn_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                      const float *x, int64_t ldx, float *y, int64_t ldy,
                      int64_t n_rows, int64_t n_cols, int32_t d,
                      float out_scale, const float *const *adds_host, int32_t n_adds,
                      float add_scale, const float *row_scale, const float *col_scale,
                      const igcn_long_row *long_rows, int64_t n_long_rows,
                      const igcn_row_segment *segments, int64_t n_segments,
                      float *partial, int32_t long_threshold,
                      const int32_t *edge_id, uint64_t seed, float keep_prob,
                      const uint8_t *row_mask, int32_t flags /* IGCN_SPMM_MASKED_ROWS_ZERO | IGCN_SPMM_CLOSING_SEGMENTS */,
                      int64_t nnz, const int32_t *row_order, const uint32_t *col_mask,
                      const uint64_t *seed_dev, const int64_t *xcd_off, const uint32_t *order_bits, void *stream);

/* The same launch through ONE struct (ABI v10) — the form a binding should use: the positional call above has 34 arguments.
 * Zero-initialise the struct, set struct_size = sizeof(igcn_spmm_args) and fill what the call site has.  The reference call
 * site (model.py:99-102: a graph, X and the edge values) has exactly the eight fields of the first block; every field
 * below it may stay zero: zero / NULL means "not used" or "the library default" (ldx / ldy 0 = d: dense rows;
 * out_scale / add_scale / keep_prob 0 = 1; nnz 0 = unknown).  The library reads the first struct_size bytes only, so a
 * caller compiled against this struct keeps working when later versions append fields (IGCN_E_SHAPE when struct_size
 * does not even cover the first block).  Field meanings: as the arguments of igcn_spmm_csr_f32 above; `adds` are the
 * device pointers themselves (no host array to keep alive).
 * tune_*: the result-neutral launch knobs of THIS call — value + 1 (0 = not given: the process-wide igcn_set_tuning
 * value, else the library default): "spmm_blocks_per_cu", "spmm_multirow", "spmm_fold".  A launch reads its knobs once,
 * at entry, into a value of its own: callers on different streams or threads that want different launch shapes pass them
 * here and never touch igcn_set_tuning, which stays what it was — a process-wide developer switch, not to be flipped while
 * another thread launches. */
typedef struct igcn_spmm_args {
    uint32_t struct_size;                 /* sizeof(igcn_spmm_args) as the caller compiled it */
    uint32_t flags;                       /* IGCN_SPMM_MASKED_ROWS_ZERO | IGCN_SPMM_CLOSING_SEGMENTS */
    /* ---- required: what model.py:99-102 has ---- */
    const int64_t *rowptr;                /* [n_rows + 1]                     */
    const int32_t *col;                   /* [nnz]                            */
    const float   *val;                   /* [nnz] or NULL (all ones)         */
    int64_t        n_rows, n_cols;
    const float   *x;                     /* [n_cols, d]                      */
    float         *y;                     /* [n_rows, d]                      */
    int32_t        d;
    /* ---- optional: zero = off / default ---- */
    int32_t        n_adds;
    int64_t        ldx, ldy;              /* 0 = d                            */
    int64_t        nnz;                   /* 0 = unknown                      */
    float          out_scale, add_scale;  /* 0 = 1                            */
    float          keep_prob;             /* 0 = 1 (no dropout)               */
    int32_t        long_threshold;
    const float   *adds[IGCN_MAX_ADDS];
    const float   *row_scale, *col_scale;
    const igcn_long_row    *long_rows;  int64_t n_long_rows;
    const igcn_row_segment *segments;   int64_t n_segments;
    float         *partial;
    const int32_t *edge_id;
    uint64_t       seed;
    const uint64_t *seed_dev;
    const uint8_t *row_mask;
    const uint32_t *col_mask, *order_bits;
    const int32_t *row_order;
    const int64_t *xcd_off;
    int32_t        tune_blocks_per_cu, tune_multirow, tune_fold;     /* value + 1; 0 = not given */
    int32_t        reserved;
} igcn_spmm_args;
int igcn_spmm_csr_f32_args(const igcn_spmm_args *args, void *stream);

/* Row masks for igcn_spmm_csr_f32.  mask1[ids[i] + offsets...] = 1 for every listed row;
 * when rowptr/col are given, mask2[r] = 1 for every listed row r and every column
 * of row r (its neighbourhood).  ids int64 [n]; masks uint8 [n_rows], zeroed by the
 * caller; mask2 may be NULL. */
int igcn_mark_rows(const int64_t *ids, int64_t n, const int64_t *rowptr, const int32_t *col,
                   uint8_t *mask1, uint8_t *mask2, int64_t n_rows, void *stream);

/* n_masks uint8 masks of n entries each (mask m starts at masks + m * stride) -> one bit per entry, (n + 31) / 32
 * uint32 words per mask (mask m at bits + m * words): the form igcn_spmm_csr_f32's col_mask takes — 8x smaller, so
 * the lookups of a launch stay in the CU's L1 instead of going to L2. */
int igcn_pack_mask_bits(const uint8_t *masks, int64_t n, int64_t stride, int32_t n_masks, uint32_t *bits, void *stream);

/* igcn_pack_mask_bits and, in the same launch, masks[0] in the dealing order of a matrix with n rows: order_bits as
 * igcn_spmm_csr_f32 takes it (uint32, room for 2 * ceil((n + n_segments) / 64) + 2 words; the two padding words behind the
 * last entry are zeroed here).  row_order int32 [n_order] / segments: the matrix's (n_order = n + n_segments for a plain
 * dealing order, fewer with an XCD plan, whose lists hold no cut row; row_order NULL = rows, then segments, in index order —
 * what the launch visits then; n_order is ignored). */
int igcn_pack_mask_bits_ordered(const uint8_t *masks, int64_t n, int64_t stride, int32_t n_masks, uint32_t *bits,
                                const int32_t *row_order, int64_t n_order, const igcn_row_segment *segments,
                                int64_t n_segments, uint32_t *order_bits, void *stream);

/* Device-side index utilities for the graph-swap path (model.py:402-421 is_updating,
 * run/dropui/igcn_dropui.py:26-35): the reference rebuilds its sparse structures on the host
 * with scipy (utils.py:32-38).
 *  igcn_csr_from_sorted_coo: rowptr [n_rows+1] of a row-major sorted COO (sorted_row int64 [nnz]).
 *  igcn_csr_transpose: CSR of M^T (t_rowptr int64 [n_cols+1], t_col int32 [nnz] = source rows in
 *  ascending order) and edge_id int32 [nnz] = position of each transposed entry in M, so that
 *  igcn_spmm_csr_f32 drops the same edges in both views.  workspace: 256-byte aligned,
 *  igcn_csr_transpose_workspace_bytes(nnz) bytes; nnz < 2^31.  The one call that refuses a CAPTURING stream
 *  (IGCN_E_CAPTURE): its sort is rocPRIM's, whose kernels carry a private segment (csrc/common.h, capture_guard). */
int igcn_csr_from_sorted_coo(const int64_t *sorted_row, int64_t nnz, int64_t n_rows, int64_t *rowptr, void *stream);
int64_t igcn_csr_transpose_workspace_bytes(int64_t nnz);
int igcn_csr_transpose(const int64_t *rowptr, const int32_t *col, int64_t n_rows, int64_t n_cols, int64_t nnz,
                       int64_t *t_rowptr, int32_t *t_col, int32_t *edge_id, void *workspace, void *stream);

/* out[e] = row_sum[row(e)] ^ exponent for every stored entry of a CSR matrix:
 * IGCN.update_feat_mat, model.py:374-377, as explicit values (the propagation
 * path itself uses row_scale instead and never materialises them). */
int igcn_csr_row_pow_f32(const int64_t *rowptr, const float *row_sum, float exponent,
                         float *val_out, float *row_scale_out, int64_t n_rows, void *stream);

/* Fused BPR triplet scoring, forward:  trainer.py:238-243 (+ :306-311 with w),
 * model.py:110-116 / :295-299 / :62-67 (gathers and squared norms).
 *   pos_b = sum_j U[u_b,j] P[p_b,j] w_j ,  neg_b likewise with N[n_b]
 *   loss_out[0] = (1/B) sum_b softplus(neg_b - pos_b)
 *   loss_out[1] = (1/B) sum_b (|L_u[u_b]|^2 + |L_p[p_b]|^2 + |L_n[n_b]|^2)
 * U/P/N are the tables the scores gather from (the caller applies row offsets
 * such as "+ n_users" to the base pointers), leading dimension ld;
 * l2_* are the tables the L2 term gathers from (may equal the score tables;
 * all three NULL skips the term and loss_out[1] = 0); w is NULL or [d].
 * work: device scratch of 3*B floats; work[0..B) holds sigmoid(neg_b - pos_b)
 * afterwards and must be passed unchanged to igcn_bpr_bwd_f32.
 * The two sums are reduced in a fixed order (bitwise reproducible). */
int igcn_bpr_fwd_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                     const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                     const int64_t *users, const int64_t *pos, const int64_t *neg,
                     int64_t batch, int32_t d, const float *w,
                     float *loss_out, float *work, void *stream);

/* The forward pass split in two for the embedding-column-sharded multi-GPU path, where a
 * rank holds only a slice of every row: igcn_bpr_dots_f32 writes the PARTIAL dots of the
 * slice, dots = [pos_b | neg_b | l2_b] (3*B floats); the caller sums them over the ranks
 * (all-reduce); igcn_bpr_finish_f32 turns the complete dots into loss_out[0..1] and the
 * `work` buffer igcn_bpr_bwd_f32 expects.  Arguments as igcn_bpr_fwd_f32. */
int igcn_bpr_dots_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                      const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                      const int64_t *users, const int64_t *pos, const int64_t *neg,
                      int64_t batch, int32_t d, const float *w, float *dots, void *stream);
int igcn_bpr_finish_f32(const float *dots, int64_t batch, float *loss_out, float *work, void *stream);

/* Backward of igcn_bpr_fwd_f32: accumulates (float atomic add, 256-byte row
 * segments) row-sparse gradients into dense gradient tables laid out like the
 * forward tables (same leading dimensions and base offsets).
 *   g_out: DEVICE pointer to 2 floats, d total / d loss_out[0..1].
 * gw_out [d] (NULL if w is NULL) accumulates d/dw.  Gradient tables for the L2
 * term may be NULL when the l2 tables are NULL. */
int igcn_bpr_bwd_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                     const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                     const int64_t *users, const int64_t *pos, const int64_t *neg,
                     int64_t batch, int32_t d, const float *w, const float *work,
                     const float *g_out,
                     float *gu_tab, float *gp_tab, float *gn_tab,
                     float *gl2_u_tab, float *gl2_p_tab, float *gl2_n_tab,
                     float *gw_out, void *stream);

/* Fused score + mask + top-k:  torch.mm at model.py:122 / :71, the -inf masking
 * of trainer.py:149-161 and torch.topk at trainer.py:163, without ever
 * materialising the [B, n_items] score matrix.
 *   score[b,i] = <user_rows[user_ids[b]], item_rows[i]>   (exact fp32 MFMA chain)
 *   masked (treated as -inf): items in excl_col[excl_rowptr[u]..excl_rowptr[u+1])
 *   with u = user_ids[b] (sorted ascending per user), and items with
 *   banned[i] != 0.
 *   out_idx[b, 0..k) = the k best item ids, best first; ties -> lower id first;
 *   out_val[b, 0..k) = their scores (-inf for masked fill-ins).
 * user_ids int64 [B] or NULL (then row b of user_rows is user b);
 * excl_rowptr int64 / excl_col int32 may be NULL; banned uint8 [n_items] or NULL.
 * d <= 256, d % 4 == 0 (built for d <= 128: 64 users per wave up to 64, 32 at 128; 129..256 runs the same kernel with a
 * whole item row and user row in registers — ~340 of them, so ONE wave per SIMD: correct, slower per flop); k <= IGCN_MAX_TOPK and k <= n_items (k <= 24 runs 8 waves per CU; the heaps of a
 * larger k take more of the CU's LDS and fewer waves are resident: 4 up to 56, 2 up to 120, 1 above).
 * workspace (8-byte aligned): igcn_score_topk_workspace_bytes(B, n_items, d, k) bytes (partial lists of the
 * item-range splits that fill the chip when B is small + the banned items packed one bit each). */
int64_t igcn_score_topk_workspace_bytes(int64_t batch, int64_t n_items, int32_t d, int32_t k);
int igcn_score_topk_f32(const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                        const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                        const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                        int32_t k, int64_t *out_idx, float *out_val,
                        void *workspace, void *stream);

/* igcn_score_topk_f32 for users whose k-th best score the caller can bound from below (ABI v5): lower_bound float
 * [batch] (device) — only items whose score reaches lower_bound[b] are looked at, which removes the list warm-up of a
 * sweep (k ln(n / k) heap updates per user, half of them in the first 2 % of the items).  The bound must be valid (at
 * least k unmasked items score >= it), else the list comes back short (id -1).  Same lists as igcn_score_topk_f32
 * otherwise.  Used by the two-stage evaluation for the users it hands back (their re-scored candidates give the bound).
 * No reference counterpart: trainer.py:163 is a dense torch.topk. */
int igcn_score_topk_bounded_f32(const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                                const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                                const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                                int32_t k, const float *lower_bound, int64_t *out_idx, float *out_val,
                                void *workspace, void *stream);

/* The same evaluation in two stages (d = 64 or 128, k <= 60).  Stage 1 sweeps all items on the 16-bit matrix cores — items
 * and users as ONE fp16 plane each, both tables rescaled by a power of two, fp32 accumulate: scores off by at most
 * 2^-10 |u| max|i| (igcn_set_tuning("topk_fast_mode", 2): users as two fp16 planes, 2^-11, twice the MFMAs — at d = 128
 * two 32-user groups per wave at ONE wave per SIMD, or with "topk_fast_wide" 0 one group at two waves; "topk_fast_mode" 1,
 * d = 64 only: two bf16 planes each side, three products, 2^-14) — and keeps the k + 6 best candidates of every user
 * (k + 4 in modes 1 and 2; masks applied as in igcn_score_topk_f32).
 * Stage 2 re-computes the candidates' scores in fp32 in the order the fp32 sweep adds the products, orders them
 * (score, then lower id) and writes the best k: out_idx / out_val as igcn_score_topk_f32 writes them.  A user for
 * whom an item dropped by stage 1 could still reach the k-th exact score (its bound does not stay below it: near-ties
 * at the k-th place) is reported instead of trusted: flagged[0] = how many, flagged[1..] = their positions in the
 * batch (int32 [batch + 1]), flagged_lower_bound[0..] (float [batch] or NULL) = for each of them, in the same order,
 * the k-th exact score among its candidates — a valid lower bound for igcn_score_topk_bounded_f32.
 * ABI v7: when flagged_lower_bound is given, the call FINISHES the first IGCN_FAST_FALLBACK_MAX flagged users itself: the
 * bounded fp32 sweep is launched behind stage 2, planned for that many users and run for as many as flagged[0] holds when
 * it starts (the batch of that sweep is the device-side list flagged[1..]) — no host read between the stages.  The caller
 * reads flagged[0] afterwards and re-does only positions flagged[1 + IGCN_FAST_FALLBACK_MAX ..] with
 * igcn_score_topk_bounded_f32 (bounds flagged_lower_bound[IGCN_FAST_FALLBACK_MAX ..]); with flagged_lower_bound NULL
 * every flagged user is the caller's (igcn_score_topk_fast_finished_max tells how many the call finishes).  No host synchronisation
 * inside, no runtime memset (round 5: the call's state is zeroed by a kernel of the library's own — a captured hipMemsetAsync becomes a
 * memset NODE, which ROCm 7.2 does not order against the kernel nodes behind it; round 4's replay fault, found under rocgdb) and no
 * kernel with a private segment: the call can be captured into a HIP graph by the standard recipe — warm up on a side stream,
 * capture, replay on any stream.  (Late round 4: of those first users, the ones whose bound is
 * the k-th exact score of a complete candidate list first take a streaming filter — every (user, item) pair scored once with the
 * fp32 sweep's arithmetic, the pairs that reach the bound kept and ranked — and the bounded sweep runs for the rest: users whose wave
 * gave up on them, users without a bound, users whose ties overflow the filter's 256 entries.  Same lists either way.)
 * ABI v5: stage 1 meets the items by DESCENDING squared norm (likely winners first: the running thresholds are near
 * their final values early and most later items fail the cheap selection test; -16 % on the Amazon-like evaluation),
 * not by id: a permutation, its inverse and the exclusion lists in sweep positions are built per call in the
 * workspace (a counting sort of the norms' upper 16 bits and per-row sorts of the library's own), which is why the exclusion CSR's size is passed: excl_rows = rows of excl_rowptr
 * (every user id of the batch < excl_rows), excl_nnz = its entries; both 0 when excl_rowptr is NULL.  The lists
 * returned do not depend on the order.  igcn_set_tuning("topk_fast_order", 0) sweeps in id order.
 * In that order a wave leaves the sweep once no row still to come can reach any of its users (|score| <= |u| |i|), and a
 * wave that outlasts three quarters of the others hands the users it could not finish to the flagged list as well
 * (ABI v6; which users take that way depends on timing, the lists do not).  Round 4: a wave checks whether it may leave every 6
 * tiles up to tile 48 and every 24 after that, and may hand its users over from tile 12 on ("topk_fast_early_checks": the
 * cadence, 0 = every 24 tiles / from tile 48) — on trained tables most users are out of reach after 8 tiles.
 * Where nobody can leave that early (item rows at tile 128 still half as long as the first: untrained or normalised tables), a
 * whole sweep at d = 64, k + extra <= 32 begins with a warm-up pass over its first 128 tiles that only keeps the best score of each
 * accumulator slot; the (k + extra)-th largest of a user's 32 slot maxima bounds its (k + extra)-th best score from below, and the
 * sweep proper starts from that threshold instead of from an empty list ("topk_fast_warm": the tiles, 0 = none).
 * workspace: igcn_score_topk_fast_workspace_bytes(...) bytes, 256-byte aligned. */
int64_t igcn_score_topk_fast_workspace_bytes(int64_t batch, int64_t n_items, int32_t d, int32_t k,
                                             int64_t excl_rows, int64_t excl_nnz);
/* How many flagged users a call of that batch size finishes itself under the current knobs (ABI v8): 0 without flagged_lower_bound
 * or with "topk_fast_fallback" 0, else min(batch, IGCN_FAST_FALLBACK_MAX).  The caller re-does flagged[1 + that ..]. */
int64_t igcn_score_topk_fast_finished_max(int64_t batch, int32_t with_lower_bound);
int igcn_score_topk_fast_f32(const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                             const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                             const int64_t *excl_rowptr, const int32_t *excl_col, int64_t excl_rows, int64_t excl_nnz,
                             const uint8_t *banned, int32_t k, int64_t *out_idx, float *out_val,
                             int32_t *flagged, float *flagged_lower_bound, void *workspace, void *stream);

/* hit[u, j] = 1 if rec[u, j] is in eval_col[eval_rowptr[u]..eval_rowptr[u+1])
 * (sorted ascending), else 0: the membership loop of trainer.py:111-115.
 * eval_col may be NULL when every list is empty (eval_rowptr all equal): no hit anywhere. */
int igcn_hit_matrix(const int64_t *rec, int64_t n_users, int32_t k,
                    const int64_t *eval_rowptr, const int32_t *eval_col,
                    float *hit, void *stream);

/* calculate_metrics (trainer.py:109-138) without the hit matrix: for every cut-off topks_host[t] (host array, n_topks <=
 * IGCN_MAX_METRIC_CUTS, each <= k_rec) the SUMS over the users with a non-empty list of  hits / k,  hits / |list|,
 * DCG / IDCG  (per user in float32 as the reference forms them, summed in float64 in a fixed order):
 *   out[3 t], out[3 t + 1], out[3 t + 2]  (device, double [3 * IGCN_MAX_METRIC_CUTS + 1]),  out[3 * IGCN_MAX_METRIC_CUTS] = that number
 * of users; Precision / Recall / NDCG @ k_t are the sums divided by it (0 / 0 = nan, the reference's empty mean).
 * rec int64 [n_users, k_rec] = the recommended ids; eval lists as for igcn_hit_matrix (eval_col NULL: all empty).
 * workspace: igcn_eval_metrics_workspace_bytes(n_users) bytes, 8-byte aligned. */
#define IGCN_MAX_METRIC_CUTS 8
int64_t igcn_eval_metrics_workspace_bytes(int64_t n_users);
int igcn_eval_metrics(const int64_t *rec, int64_t n_users, int32_t k_rec,
                      const int64_t *eval_rowptr, const int32_t *eval_col,
                      const int32_t *topks_host, int32_t n_topks, double *out, void *workspace, void *stream);

/* Device-side BPR negative sampler (dataset.py:119-131): for each of `batch`
 * draws, a uniform user with a non-empty train list, a uniform positive from
 * it, and a uniform negative item rejected while it is in the list.
 * train_rowptr int64 / train_col int32 sorted per user; nonempty_users int32
 * [n_nonempty]; out [batch, 3] int64 (user, pos, neg). */
int igcn_bpr_sample(const int64_t *train_rowptr, const int32_t *train_col,
                    const int32_t *nonempty_users, int64_t n_nonempty, int64_t n_items,
                    int64_t batch, uint64_t seed, int64_t *out, void *stream);

/* The same draws as igcn_bpr_sample(seed), written as the NODE ids a graph model gathers (model.py:110-115):
 * out [3, batch] = users | item_offset + positives | item_offset + negatives (item_offset = n_users). */
int igcn_bpr_sample_nodes(const int64_t *train_rowptr, const int32_t *train_col,
                          const int32_t *nonempty_users, int64_t n_nonempty, int64_t n_items,
                          int64_t batch, uint64_t seed, int64_t item_offset, int64_t *out, void *stream);

/* igcn_bpr_fwd_f32 / igcn_bpr_bwd_f32 for a caller that wants the training loss itself (trainer.py:242:
 * loss = bpr + l2_reg * mean l2_norm_sq) as ONE differentiable scalar — no elementwise launches between the kernels:
 * loss_out3[0..1] as igcn_bpr_fwd_f32, loss_out3[2] = loss_out3[0] + l2_weight * loss_out3[1];
 * the backward takes g_loss = d total / d loss_out3[2] (DEVICE pointer to one float) and the same l2_weight. */
int igcn_bpr_loss_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                      const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                      const int64_t *users, const int64_t *pos, const int64_t *neg,
                      int64_t batch, int32_t d, const float *w, float l2_weight,
                      float *loss_out3, float *work, void *stream);
int igcn_bpr_loss_bwd_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                          const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                          const int64_t *users, const int64_t *pos, const int64_t *neg,
                          int64_t batch, int32_t d, const float *w, const float *work, const float *g_loss,
                          float l2_weight,
                          float *gu_tab, float *gp_tab, float *gn_tab,
                          float *gl2_u_tab, float *gl2_p_tab, float *gl2_n_tab,
                          float *gw_out, void *stream);

/* igcn_bpr_loss_bwd_f32 for a loss that enters the total with a weight: d total / d (bpr_weight * bpr + l2_weight * l2)
 * given g_loss = d total / d that sum (DEVICE pointer to one float).  The INMO step's auxiliary loss
 * (trainer.py:304-312: aux_reg * mean softplus over raw template rows weighted by w) adds its row gradients straight
 * into the dense gradient of the template table with it — no zero-filled table and no dense add of its own. */
int igcn_bpr_loss_bwd_scaled_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                                 const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                                 const int64_t *users, const int64_t *pos, const int64_t *neg,
                                 int64_t batch, int32_t d, const float *w, const float *work, const float *g_loss,
                                 float bpr_weight, float l2_weight,
                                 float *gu_tab, float *gp_tab, float *gn_tab,
                                 float *gl2_u_tab, float *gl2_p_tab, float *gl2_n_tab,
                                 float *gw_out, void *stream);

/* The row-sparse tail of a training step's backward pass (trainer.py:244-246: loss.backward()), one launch:
 *  dst != NULL: dst[ids[i]] += scale_host * (scale_dev ? *scale_dev : 1) * src[ids[i]] for i < n, float atomics (an id may
 *  repeat) — the gradient of the L2 term on the raw embedding rows (model.py:110-113) added to the dense gradient the
 *  propagation backward produced, instead of a second dense table and a dense add;
 *  zero_tab != NULL: zero_tab[ids[i]] = 0 — puts the persistent batch-gradient table back to zero after a step. */
int igcn_rows_finish_f32(float *dst, int64_t ldd, const float *src, int64_t lds, float *zero_tab, int64_t ldz,
                         const int64_t *ids, int64_t n, int32_t d, const float *scale_dev, float scale_host, void *stream);

/* Row-sharded multi-GPU training (igcn_cf_amd/dist.py; not in the reference, which is single-device): of the n node ids
 * of a batch (users < n_users <= items) this rank owns the users in [ulo, uhi) and the items (id - n_users) in
 * [ilo, ihi).  gather: out[i, 0:d] = the owned row of tab_u / tab_i (local row = id - ulo / item - ilo), zeros for
 * ids of other ranks — every rank fills its rows and ONE all-reduce completes the [n, d] block.  scatter_add: the
 * gradient rows of the ids this rank owns are added (float atomics; ids may repeat) into its local tables. */
int igcn_owned_rows_gather_f32(const int64_t *ids, int64_t n, int64_t n_users, int64_t ulo, int64_t uhi,
                               int64_t ilo, int64_t ihi, const float *tab_u, int64_t ld_u,
                               const float *tab_i, int64_t ld_i, int32_t d, float *out, int64_t ld_out, void *stream);
int igcn_owned_rows_scatter_add_f32(const int64_t *ids, int64_t n, int64_t n_users, int64_t ulo, int64_t uhi,
                                    int64_t ilo, int64_t ihi, const float *g, int64_t ld_g, int32_t d,
                                    float *gu, int64_t ld_gu, float *gi, int64_t ld_gi, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IGCN_HIP_H */
