// Sweep order of the two-stage evaluation (igcn_score_topk_fast_f32): the candidate sweep meets the items by DESCENDING
// squared norm instead of by id.
//
// Why: a user's running k-th-best threshold only rises, and every item that beats it costs a staging + heap update
// (k ln(n / k) of them per user for exchangeable scores).  Large-norm rows are the likely winners (a 1.2 x longer row
// needs 2.9 instead of 3.5 sigma of alignment to reach a top-20 of 96 k): met first, they put the thresholds near
// their final values early and most of the later items are rejected by the 12-instruction selection alone.  Measured
// on MI355X (scripts/dev_topk_item_order.py, Amazon-like shapes, no masks): two-stage sweep 4.41 -> 3.69 ms at random
// init, 3.85 -> 3.23 ms on a table with log-normal row scales; ascending norms — what an unlucky id order can be —
// 6.3 ms.  The reference has no counterpart (torch.topk over a dense score block, trainer.py:163); the lists returned
// do not depend on the order (they come from the exact re-scoring stage, ranked by score, then lower item id).
//
// Built per call, in HBM (the embeddings change between evaluations):
//   perm[pos] = item id at sweep position pos   (stable descending radix sort of the norms: ties keep ascending id)
//   inv[item] = pos
//   excl_pos  = the exclusion CSR's entries as sweep positions, ascending inside every row (the sweep walks each
//               user's list with a cursor): inv[col], then one segmented radix sort (rocPRIM) over the rows — a global
//               sort of (row << 32 | position) keys took 6 passes x 29 us on the Amazon-like lists and ate the gain.
//               (Tried and dropped, round 3: hand-written per-row sorts — rank counting across a (half-)wave for short
//               rows, bitonic networks in LDS for longer ones: 185-370 us against rocPRIM's 115 us + a host read; one
//               wave per 23-entry row is three dependent memory round trips and little else.)
// The norms are sorted on their upper 16 bits (exponent + 8 significant bits: two radix passes): any order is valid, a finer one buys
// nothing, and the sort is two passes shorter.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include "topk_order.h"

namespace igcn {

static inline int64_t al256(int64_t n) { return (n + 255) / 256 * 256; }

__global__ __launch_bounds__(kBlock) void order_keys_kernel(const float *__restrict__ norm2, int64_t n, uint32_t *__restrict__ keys,
                                                            int32_t *__restrict__ iota)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    keys[i] = __float_as_uint(norm2[i]);             // non-negative floats: the bit patterns order like the values
    iota[i] = (int32_t)i;
}

__global__ __launch_bounds__(kBlock) void invert_perm_kernel(const int32_t *__restrict__ perm, int64_t n, int32_t *__restrict__ inv)
{
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p < n) inv[perm[p]] = (int32_t)p;
}

__global__ __launch_bounds__(kBlock) void excl_positions_kernel(const int32_t *__restrict__ col, int64_t nnz, const int32_t *__restrict__ inv,
                                                                uint32_t *__restrict__ pos)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e < nnz) pos[e] = (uint32_t)inv[col[e]];
}

static int bits_for(int64_t n) { int b = 1; while (b < 31 && ((int64_t)1 << b) < n) ++b; return b; }
constexpr int kNormBeginBit = 15;                    // float bits [15, 31): exponent + 8 significant bits (the sign is 0)

int topk_order_layout(int64_t n_items, int64_t excl_rows, int64_t excl_nnz, TopkOrderLayout *L)
{
    if (n_items < 1 || n_items >= ((int64_t)1 << 31) || excl_rows < 0 || excl_nnz < 0 || excl_nnz >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    size_t t1 = 0, t2 = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, t1, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const int32_t *)nullptr,
                                                  (int32_t *)nullptr, (size_t)n_items, kNormBeginBit, 31, (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    if (excl_nnz > 0) {
        e = rocprim::segmented_radix_sort_keys(nullptr, t2, (const uint32_t *)nullptr, (uint32_t *)nullptr, (unsigned int)excl_nnz,
                                               (unsigned int)excl_rows, (const int64_t *)nullptr, (const int64_t *)nullptr, 0,
                                               bits_for(n_items), (hipStream_t)0);
        if (e != hipSuccess) return (int)e;
    }
    L->tmp_bytes = t1 > t2 ? t1 : t2;
    int64_t off = 0;
    L->norm2 = off; off += al256(n_items * 4);
    L->keys = off; off += al256(n_items * 4);
    L->keys_sorted = off; off += al256(n_items * 4);
    L->iota = off; off += al256(n_items * 4);
    L->perm = off; off += al256(n_items * 4);
    L->inv = off; off += al256(n_items * 4);
    L->ekeys = off; off += al256(excl_nnz * 4);
    L->ekeys_sorted = 0;
    L->excl_pos = off; off += al256(excl_nnz * 4);
    L->tmp = off; off += al256((int64_t)L->tmp_bytes);
    L->total = off;
    return IGCN_OK;
}

int topk_order_build(const TopkOrderLayout &L, char *ws, int64_t n_items, const int64_t *excl_rowptr, const int32_t *excl_col,
                     int64_t excl_rows, int64_t excl_nnz, hipStream_t st, const int32_t **perm_out, const int32_t **excl_pos_out)
{
    const float *norm2 = reinterpret_cast<const float *>(ws + L.norm2);
    uint32_t *keys = reinterpret_cast<uint32_t *>(ws + L.keys), *keys_sorted = reinterpret_cast<uint32_t *>(ws + L.keys_sorted);
    int32_t *iota = reinterpret_cast<int32_t *>(ws + L.iota), *perm = reinterpret_cast<int32_t *>(ws + L.perm);
    int32_t *inv = reinterpret_cast<int32_t *>(ws + L.inv);
    const unsigned ib = (unsigned)((n_items + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(order_keys_kernel, dim3(ib), dim3(kBlock), 0, st, norm2, n_items, keys, iota);
    size_t tmp_bytes = L.tmp_bytes;
    hipError_t e = rocprim::radix_sort_pairs_desc(ws + L.tmp, tmp_bytes, (const uint32_t *)keys, keys_sorted, (const int32_t *)iota, perm,
                                                  (size_t)n_items, kNormBeginBit, 31, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(invert_perm_kernel, dim3(ib), dim3(kBlock), 0, st, (const int32_t *)perm, n_items, inv);
    *perm_out = perm;
    *excl_pos_out = nullptr;
    if (excl_nnz > 0) {
        if (excl_rows >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
        uint32_t *raw = reinterpret_cast<uint32_t *>(ws + L.ekeys), *pos = reinterpret_cast<uint32_t *>(ws + L.excl_pos);
        const unsigned eb = (unsigned)((excl_nnz + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(excl_positions_kernel, dim3(eb), dim3(kBlock), 0, st, excl_col, excl_nnz, (const int32_t *)inv, raw);
        tmp_bytes = L.tmp_bytes;
        e = rocprim::segmented_radix_sort_keys(ws + L.tmp, tmp_bytes, (const uint32_t *)raw, pos, (unsigned int)excl_nnz,
                                               (unsigned int)excl_rows, excl_rowptr, excl_rowptr + 1, 0, bits_for(n_items), st);
        if (e != hipSuccess) return (int)e;
        *excl_pos_out = reinterpret_cast<const int32_t *>(pos);
    }
    return launch_status();
}

}  // namespace igcn
