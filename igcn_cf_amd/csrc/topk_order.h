// Internal interface between score_topk.hip and topk_order.hip (the sweep order of the two-stage evaluation).
#pragma once
#include "common.h"

namespace igcn {

// byte offsets into the order workspace (each 256-aligned); norm2 is filled by topk_row_stats_kernel
struct TopkOrderLayout {
    // bins: the counting sort's 65 536 counters — FIRST in the workspace and zero on entry: the caller clears them with the memset that
    // clears its own state right in front of this workspace (igcn_score_topk_fast_f32).  huge: [count][row ids] of the rows the wave
    // kernel leaves to the workgroup kernel.
    int64_t bins, totals, status, norm2, keys, perm, inv, needed, huge, excl_pos, total;
};
constexpr int64_t kOrderBinsBytes = (int64_t)65536 * 4;

int topk_order_layout(int64_t n_items, int64_t excl_rows, int64_t excl_nnz, TopkOrderLayout *L);

// perm (position -> item id, descending |row|^2 on its upper 16 bits, equal keys in no particular order), its inverse, and — when an exclusion CSR is given
// — that CSR's entries as sweep positions, ascending inside every row, for the rows the call's users own (user_ids[0..batch), or
// rows 0..batch-1 when user_ids is NULL).  Everything lives in `ws`, whose first kOrderBinsBytes must be zero on entry — if they are
// not, the build notices (the counts no longer add up to n_items), takes the identity order and leaves 1 in ws[status]: the call's lists
// stay right, its sweep is the slower id-order one.
int topk_order_build(const TopkOrderLayout &L, char *ws, int64_t n_items, const int64_t *excl_rowptr, const int32_t *excl_col,
                     int64_t excl_rows, int64_t excl_nnz, const int64_t *user_ids, int64_t batch, hipStream_t st,
                     const int32_t **perm_out, const int32_t **excl_pos_out);

}  // namespace igcn
