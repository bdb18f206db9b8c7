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
//               user's list with a cursor): inv[col], sorted row by row by three kernels of this file (round 4) —
//               rows of <= 32 entries by a half-wave each (rank counting: an entry's place is the number of smaller ones,
//               32 shuffles), rows of 33 .. 256 by a wave each (the same with four entries per lane), longer rows by a
//               workgroup each (a bitonic network in its all-ascending form, in LDS up to 8 192 entries, in place in HBM
//               beyond).  Which class a row falls into is decided on the device: the first kernel appends the rows it does
//               not take to two lists, the other two run over fixed grids and read the lists' lengths there — NO host
//               read.  Rounds 2-3 used rocprim::segmented_radix_sort_keys, which partitions its segments by size and copies
//               the partition sizes to the HOST before it can launch its sort kernels: a stream synchronisation inside an
//               entry point whose contract says there is none (it could not be captured into a HIP graph), 115 us of kernels
//               and 15-140 us of an idle GPU behind the read (profiles/r03ae_*, r04d_*).  A global sort of
//               (row << 32 | position) keys took 6 passes x 29 us on the Amazon-like lists.
// The norms are sorted on their upper 16 bits (exponent + 8 significant bits: two radix passes): any order is valid, a finer one buys
// nothing, and the sort is two passes shorter.
#include <rocprim/device/device_radix_sort.hpp>
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

__global__ __launch_bounds__(kBlock) void invert_perm_kernel(const int32_t *__restrict__ perm, int64_t n, int32_t *__restrict__ inv,
                                                             unsigned int *__restrict__ list_counts)
{
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p < n) inv[perm[p]] = (int32_t)p;
    if (p == 0 && list_counts) { list_counts[0] = 0u; list_counts[1] = 0u; }      // lengths of the two row lists of the kernels below
}

// ---- the exclusion lists as sweep positions, ascending inside every row ------------------------------------------------
constexpr int kExclShort = 32, kExclMid = 256, kExclLds = 8192;
constexpr uint32_t kPosNone = 0xFFFFFFFFu;

// Rows of <= kExclShort entries: one HALF-WAVE per row.  Longer rows go to mid_list (<= kExclMid) / long_list.
__global__ __launch_bounds__(kBlock) void excl_sort_short_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                 int64_t n_rows, const int32_t *__restrict__ inv, uint32_t *__restrict__ pos,
                                                                 int32_t *__restrict__ mid_list, int32_t *__restrict__ long_list,
                                                                 unsigned int *__restrict__ list_counts)
{
    const int l32 = threadIdx.x & 31;
    const int64_t hw = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 5;
    const int64_t n_hw = ((int64_t)gridDim.x * kBlock) >> 5;
    for (int64_t r = hw; r < n_rows; r += n_hw) {
        const int64_t s = rowptr[r];
        const int len = (int)(rowptr[r + 1] - s);
        if (len <= 0) continue;
        if (len > kExclShort) {
            if (l32 == 0) {
                const int which = len > kExclMid ? 1 : 0;
                const unsigned int slot = atomicAdd(list_counts + which, 1u);
                (which ? long_list : mid_list)[slot] = (int32_t)r;
            }
            continue;
        }
        uint32_t key = kPosNone;
        if (l32 < len) key = (uint32_t)inv[col[s + l32]];
        int rank = 0;
        for (int c = 0; c < len; ++c) {                          // (len is uniform inside the half-wave; the shuffle stays inside it)
            const uint32_t other = (uint32_t)__shfl((int)key, c, 32);
            rank += (other < key || (other == key && c < l32)) ? 1 : 0;
        }
        if (l32 < len) pos[s + rank] = key;
    }
}

// Rows of kExclShort + 1 .. kExclMid entries: one WAVE per row of mid_list, up to four entries per lane.
__global__ __launch_bounds__(kBlock) void excl_sort_mid_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                               const int32_t *__restrict__ inv, uint32_t *__restrict__ pos,
                                                               const int32_t *__restrict__ mid_list, const unsigned int *__restrict__ list_counts)
{
    constexpr int Q = kExclMid / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t w0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_w = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t n = list_counts[0];
    for (int64_t i = w0; i < n; i += n_w) {
        const int64_t r = mid_list[i];
        const int64_t s = rowptr[r];
        const int len = (int)(rowptr[r + 1] - s);
        uint32_t key[Q];
        int rank[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int e = q * kWave + lane;
            key[q] = e < len ? (uint32_t)inv[col[s + e]] : kPosNone;
            rank[q] = 0;
        }
#pragma unroll
        for (int q2 = 0; q2 < Q; ++q2) {
            const int lim = len - q2 * kWave < kWave ? len - q2 * kWave : kWave;      // entries held in register q2 (wave-uniform)
            for (int c = 0; c < lim; ++c) {
                const uint32_t other = (uint32_t)__shfl((int)key[q2], c);
                const int oc = q2 * kWave + c;
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    rank[q] += (other < key[q] || (other == key[q] && oc < q * kWave + lane)) ? 1 : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (q * kWave + lane < len) pos[s + rank[q]] = key[q];
    }
}

// Longer rows: one WORKGROUP per row of long_list.  Bitonic network in the form whose compare-exchanges all point the same way
// (stage k first pairs i with i ^ (k - 1), then with i ^ j for j = k / 4 ... 1): an index past the end of the row stands for
// +infinity and its exchanges are simply skipped, so a row needs no padding to a power of two and is sorted where it lies —
// in LDS up to kExclLds entries, in its own place in HBM beyond (a user who excludes a twelfth of a 96 k-item table and more).
__global__ __launch_bounds__(kBlock) void excl_sort_long_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                const int32_t *__restrict__ inv, uint32_t *__restrict__ pos,
                                                                const int32_t *__restrict__ long_list, const unsigned int *__restrict__ list_counts)
{
    __shared__ uint32_t lds[kExclLds];
    const int64_t n = list_counts[1];
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t r = long_list[i];
        const int64_t s = rowptr[r];
        const int len = (int)(rowptr[r + 1] - s);
        uint32_t *buf = len <= kExclLds ? lds : pos + s;
        for (int e = threadIdx.x; e < len; e += kBlock) buf[e] = (uint32_t)inv[col[s + e]];
        __syncthreads();
        int top = 1;
        while (top < len) top <<= 1;
        for (int k = 2; k <= top; k <<= 1) {
            for (int j = k - 1; j > 0; j = (j == k - 1) ? (k >> 2) : (j >> 1)) {     // k - 1 (the flip), then k / 4, k / 8, ..., 1
                for (int t = threadIdx.x; t < top / 2; t += kBlock) {
                    int lo, hi;
                    if (j == k - 1) {                                     // pairs (i, i ^ (k - 1)) inside blocks of k
                        const int blk = t / (k >> 1), off = t % (k >> 1);
                        lo = blk * k + off;
                        hi = blk * k + (k - 1 - off);
                    } else {                                              // pairs (i, i + j) with bit j of i clear
                        lo = ((t / j) * 2 * j) + (t % j);
                        hi = lo + j;
                    }
                    if (hi < len) {
                        const uint32_t a = buf[lo], b = buf[hi];
                        if (b < a) { buf[lo] = b; buf[hi] = a; }
                    }
                }
                __syncthreads();
                if (j == 1) break;
                if (j == k - 1 && (k >> 2) == 0) break;                  // k = 2: the flip was the whole stage
            }
        }
        if (len <= kExclLds)
            for (int e = threadIdx.x; e < len; e += kBlock) pos[s + e] = lds[e];
        __syncthreads();                                                  // the next row re-uses the buffer
    }
}

static int bits_for(int64_t n) { int b = 1; while (b < 31 && ((int64_t)1 << b) < n) ++b; return b; }
constexpr int kNormBeginBit = 15;                    // float bits [15, 31): exponent + 8 significant bits (the sign is 0)

int topk_order_layout(int64_t n_items, int64_t excl_rows, int64_t excl_nnz, TopkOrderLayout *L)
{
    if (n_items < 1 || n_items >= ((int64_t)1 << 31) || excl_rows < 0 || excl_nnz < 0 || excl_nnz >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    size_t t1 = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, t1, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const int32_t *)nullptr,
                                                  (int32_t *)nullptr, (size_t)n_items, kNormBeginBit, 31, (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    L->tmp_bytes = t1;
    int64_t off = 0;
    L->norm2 = off; off += al256(n_items * 4);
    L->keys = off; off += al256(n_items * 4);
    L->keys_sorted = off; off += al256(n_items * 4);
    L->iota = off; off += al256(n_items * 4);
    L->perm = off; off += al256(n_items * 4);
    L->inv = off; off += al256(n_items * 4);
    L->ekeys = off; off += excl_nnz > 0 ? al256(excl_rows * 4) * 2 + 256 : 0;     // [mid_list][long_list][two counters]
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
    int32_t *mid_list = reinterpret_cast<int32_t *>(ws + L.ekeys), *long_list = reinterpret_cast<int32_t *>(ws + L.ekeys + al256(excl_rows * 4));
    unsigned int *list_counts = reinterpret_cast<unsigned int *>(ws + L.ekeys + 2 * al256(excl_rows * 4));
    hipLaunchKernelGGL(invert_perm_kernel, dim3(ib), dim3(kBlock), 0, st, (const int32_t *)perm, n_items, inv,
                       excl_nnz > 0 ? list_counts : (unsigned int *)nullptr);
    *perm_out = perm;
    *excl_pos_out = nullptr;
    if (excl_nnz > 0) {
        if (excl_rows >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
        uint32_t *pos = reinterpret_cast<uint32_t *>(ws + L.excl_pos);
        const int64_t cus = cu_count();
        int64_t sb = (excl_rows * 32 + kBlock - 1) / kBlock;                    // a half-wave per row, at most 16 workgroups per CU
        if (sb > 16 * cus) sb = 16 * cus;
        hipLaunchKernelGGL(excl_sort_short_kernel, dim3((unsigned)sb), dim3(kBlock), 0, st, excl_rowptr, excl_col, excl_rows,
                           (const int32_t *)inv, pos, mid_list, long_list, list_counts);
        // the lists' lengths are only known on the device: fixed grids, every wave / workgroup walks its share of a list
        hipLaunchKernelGGL(excl_sort_mid_kernel, dim3((unsigned)(8 * cus)), dim3(kBlock), 0, st, excl_rowptr, excl_col,
                           (const int32_t *)inv, pos, (const int32_t *)mid_list, (const unsigned int *)list_counts);
        hipLaunchKernelGGL(excl_sort_long_kernel, dim3((unsigned)(4 * cus)), dim3(kBlock), 0, st, excl_rowptr, excl_col,
                           (const int32_t *)inv, pos, (const int32_t *)long_list, (const unsigned int *)list_counts);
        *excl_pos_out = reinterpret_cast<const int32_t *>(pos);
    }
    return launch_status();
}

}  // namespace igcn
