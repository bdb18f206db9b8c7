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
//   perm[pos] = item id at sweep position pos   (counting sort of the norms' upper 16 bits, descending — late round 4; rounds 2-4 ran
//               rocPRIM's radix sort: ten launches, 56 us for 96 k keys)
//   inv[item] = pos
//   excl_pos  = the exclusion CSR's entries as sweep positions, ascending inside every row (the sweep walks each
//               user's list with a cursor): inv[col], sorted row by row by two kernels of this file (round 4) —
//               rows of up to 256 entries by a wave each, in registers (bitonic networks of xor-shuffles, the <= 64-entry
//               rows of a turn interleaved), longer rows by a workgroup each (256-entry chunks sorted in registers and merged
//               through LDS up to 8 192 entries, a bitonic network in place in HBM beyond).  Which class a row falls into is decided
//               where it is sorted — NO host read.  Rounds 2-3 used rocprim::segmented_radix_sort_keys, which partitions its segments by size and copies
//               the partition sizes to the HOST before it can launch its sort kernels: a stream synchronisation inside an
//               entry point whose contract says there is none (it could not be captured into a HIP graph), 115 us of kernels
//               and 15-140 us of an idle GPU behind the read (profiles/r03ae_*, r04d_*).  A global sort of
//               (row << 32 | position) keys took 6 passes x 29 us on the Amazon-like lists.
// The norms are sorted on their upper 16 bits (exponent + 8 significant bits): any order is valid and a finer one buys nothing.  That
// makes it ONE counting pass over 65 536 bins: a kernel counts (the bins arrive zeroed), one workgroup turns the counts into start
// positions, a kernel hands out positions — and writes the inverse permutation in the same breath.  The lanes of a wave that hold
// the same key go to the bin together (one atomic per distinct key and wave: a table whose rows all have the same norm — normalised
// embeddings, an all-zero table — would otherwise queue 96 k atomics on one word).  Equal keys end up in no particular order (the
// lists do not depend on it; which users the completeness check hands to the fp32 sweep may, by a handful).
#include "topk_order.h"

namespace igcn {

static inline int64_t al256(int64_t n) { return (n + 255) / 256 * 256; }

constexpr int kOrderBins = 1 << 16;
constexpr int kNormBeginBit = 15;                    // float bits [15, 31): exponent + 8 significant bits (the sign is 0)

// bin of a squared norm, 0 = the longest rows (descending order = ascending bins)
__device__ __forceinline__ uint32_t order_bin(float n2) { return (kOrderBins - 1) - ((__float_as_uint(n2) >> kNormBeginBit) & (kOrderBins - 1)); }
// Where a bin's counter lives: neighbouring bins 256 bytes apart.  The norms of a table crowd into a few hundred neighbouring bins
// (random init: ~500), which in bin order are 16 cache lines — and device-scope atomics on one line queue up behind each other
// (~5 ns each: 34 us for 96 k items, measured); spread out, every hot bin has a line of its own.
__device__ __forceinline__ uint32_t order_slot(uint32_t bin) { return ((bin & 1023u) << 6) | (bin >> 10); }

// The lanes of the wave that hold the same bin as this one: `peers`; the lowest of them is the group's leader.
__device__ __forceinline__ unsigned long long same_bin_lanes(uint32_t bin, bool active) {
    unsigned long long todo = __ballot(active), mine = 0ull;
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t b0 = (uint32_t)__shfl((int)bin, src);
        const unsigned long long m = __ballot(active && bin == b0);
        if (active && bin == b0) mine = m;
        todo &= ~m;
    }
    return mine;
}

__global__ __launch_bounds__(kBlock) void order_count_kernel(const float *__restrict__ norm2, int64_t n, uint32_t *__restrict__ keys,
                                                             uint32_t *__restrict__ bins, int32_t *__restrict__ huge_count)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0 && huge_count) *huge_count = 0;       // (the list of long exclusion rows, filled by excl_sort_rows_kernel further down the stream)
    const bool active = i < n;
    const uint32_t bin = active ? order_bin(norm2[i]) : 0u;
    if (active) keys[i] = bin;
    const unsigned long long peers = same_bin_lanes(bin, active);
    const int lane = threadIdx.x & (kWave - 1);
    if (active && (peers & ((1ull << lane) - 1)) == 0) atomicAdd(bins + order_slot(bin), (uint32_t)__popcll(peers));
}

// counts -> start positions, in place, in two levels: workgroup q of 64 scans the 1 024 bins q << 10 ... (their counters are 64 slots
// apart: order_slot) and leaves their total in totals[q]; the 64 totals are scanned by whoever needs a position (order_place_kernel).
// (One workgroup scanning all 65 536 counters — 64 per thread, 384 shuffles each — took 45 us.)
__global__ __launch_bounds__(1024) void order_scan_kernel(uint32_t *__restrict__ bins, uint32_t *__restrict__ totals, uint32_t *__restrict__ status)
{
    __shared__ uint32_t wave_tot[1024 / kWave];
    const uint32_t slot = ((uint32_t)threadIdx.x << 6) | blockIdx.x;
    const uint32_t v = bins[slot];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += t; }
    if (lane == kWave - 1) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < 1024 / kWave; ++w) { const uint32_t t = wave_tot[w]; before += w < wave ? t : 0u; all += t; }
    bins[slot] = before + incl - v;
    if (threadIdx.x == 0) totals[blockIdx.x] = all;
    if (threadIdx.x == 0 && blockIdx.x == 0) status[0] = 0u;       // (order_place_kernel, queued behind, reports into it)
}

// positions: a wave's lanes of one bin take consecutive positions (ascending id among themselves); perm and its inverse.
// The counts are only right when the bins arrived ZEROED (the caller's job: igcn_score_topk_fast_f32 clears them with its own state).
// Should that ordering ever break again (round 4's captured memset node did it: profiles/r05b_*), the counts are inflated and the
// 64 group totals no longer add up to n — every workgroup sees that from the totals it scans anyway and falls back, all of them
// alike, to the IDENTITY order (the sweep then meets the items by id: the same lists, a slower sweep) and status[0] says so (1).
// The `pos < n` guard behind it cannot trip when the totals add up (status[0] = 2 if it ever does): kept so that no ordering bug
// can become a write outside the workspace.
__global__ __launch_bounds__(kBlock) void order_place_kernel(const uint32_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ bins,
                                                             const uint32_t *__restrict__ totals, int32_t *__restrict__ perm,
                                                             int32_t *__restrict__ inv, uint32_t *__restrict__ status)
{
    __shared__ uint32_t group_base[kOrderBins / 1024];   // items in the bins before group q of 1 024 bins
    __shared__ uint32_t grand_total;
    if (threadIdx.x < kWave) {
        const uint32_t t = totals[threadIdx.x];
        uint32_t incl = t;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if ((int)threadIdx.x >= o) incl += u; }
        group_base[threadIdx.x] = incl - t;
        if (threadIdx.x == kWave - 1) grand_total = incl;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = i < n;
    if ((int64_t)grand_total != n) {                     // (uniform over the whole grid: every workgroup reads the same totals)
        if (active) { perm[i] = (int32_t)i; inv[i] = (int32_t)i; }
        if (i == 0) status[0] = 1u;
        return;
    }
    const uint32_t bin = active ? keys[i] : 0u;
    const unsigned long long peers = same_bin_lanes(bin, active);
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned long long below = peers & ((1ull << lane) - 1);
    uint32_t start = 0;
    if (active && below == 0) start = atomicAdd(bins + order_slot(bin), (uint32_t)__popcll(peers));
    const int leader = peers ? __ffsll((long long)peers) - 1 : lane;
    start = (uint32_t)__shfl((int)start, leader);
    if (active) {
        const uint32_t pos = group_base[bin >> 10] + start + (uint32_t)__popcll(below);
        if ((int64_t)pos < n) { perm[pos] = (int32_t)i; inv[i] = (int32_t)pos; }
        else { inv[i] = (int32_t)i; status[0] = 2u; }
    }
}

__global__ __launch_bounds__(kBlock) void excl_mark_rows_kernel(const int64_t *__restrict__ user_ids, int64_t batch, int64_t n_rows,
                                                                uint8_t *__restrict__ needed)
{
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= batch) return;
    const int64_t r = user_ids ? user_ids[b] : b;
    if (r >= 0 && r < n_rows) needed[r] = 1;
}

// ---- the exclusion lists as sweep positions, ascending inside every row ------------------------------------------------
// No lists of rows, no counters, no atomics: a wave looks at kExclScan consecutive rows, sorts the ones of its classes and
// skips the rest; the workgroups of the second kernel look at 256 rows each for the (rare) huge ones.  (A first version
// appended the rows it did not take to two lists with one atomic each: 27 k atomics on two words — 240 us.)
constexpr int kExclShort = 64, kExclBig = 256, kExclLds = 8192, kExclScan = 8;     // (kExclBig: the longest row a wave sorts by itself)
constexpr uint32_t kPosNone = 0xFFFFFFFFu;

// One stage-by-stage bitonic network in the form whose compare-exchanges all point the same way (stage k first pairs i with
// i ^ (k - 1), then with i + j for j = k / 4 ... 1): an index past the end of the row stands for +infinity and its exchanges
// are simply skipped, so a row needs no padding to a power of two and can be sorted where it lies.  `sync`: what separates two
// steps (nothing for a single wave on LDS — its LDS instructions execute in order —, a barrier for a workgroup).
template <int THREADS, typename Buf, typename Sync>
__device__ __forceinline__ void bitonic_ascending(Buf buf, int len, int tid, Sync sync)
{
    int top = 1;
    while (top < len) top <<= 1;
    for (int k = 2; k <= top; k <<= 1) {
        for (int j = k - 1; j > 0;) {
            for (int t = tid; t < top / 2; t += THREADS) {
                int lo, hi;
                if (j == k - 1) {                                         // the flip: (i, i ^ (k - 1)) inside blocks of k
                    const int blk = t / (k >> 1), off = t % (k >> 1);
                    lo = blk * k + off;
                    hi = blk * k + (k - 1 - off);
                } else {                                                  // (i, i + j) with bit j of i clear
                    lo = (t / j) * 2 * j + (t % j);
                    hi = lo + j;
                }
                if (hi < len) {
                    const uint32_t a = buf[lo], b = buf[hi];
                    if (b < a) { buf[lo] = b; buf[hi] = a; }
                }
            }
            sync();
            j = (j == k - 1) ? (k >> 2) : (j >> 1);                       // k - 1, then k / 4, k / 8, ..., 1 (k = 2: the flip is all)
        }
    }
}

// Rows of up to kExclBig entries, sorted in REGISTERS by bitonic networks whose exchanges are xor-shuffles (padding = kPosNone,
// which sorts last).  A wave looks at kExclScan consecutive rows:
//   * their first 64 entries are loaded for all of them at once (two independent loads in flight per row: the column id, then
//     the position it maps to), one register slot per row, and all slots go through the 64-key network TOGETHER: 21 steps of
//     kExclScan independent shuffles, so the latency of a shuffle — and of the two dependent loads — is paid once per turn, not per
//     row (rank counting with one shuffle, or one LDS read, per key paid it per key: 240 and 170 us for the Amazon-like lists;
//     half-waves for rows of <= 32 entries with the 33 .. 64 ones taken one at a time: 74 us); the rows of <= 64 entries — 95 %
//     of them — are then stored;
//   * the rows of 65 .. 256 entries follow one at a time: four keys per lane (entry e = 64 q + lane), 36 steps of which the
//     distances >= 64 are exchanges between a lane's own registers.  (Until late round 4 the rows of 257 .. 1 024 followed here too,
//     sixteen keys per lane — ~7 us of the wave each, and the few turns that held two or three of them were the kernel's tail: with
//     those rows on the workgroup kernel's list, 72 -> 54 us here and no change there.)
__device__ __forceinline__ uint32_t bitonic_pick(uint32_t mine, uint32_t other, bool keep_min)
{
    const uint32_t lo = mine < other ? mine : other, hi = mine < other ? other : mine;
    return keep_min ? lo : hi;
}

// One row of up to 64 Q entries sorted by the whole wave: Q keys per lane (entry e = 64 q + lane), the bitonic network's distances
// below 64 as xor-shuffles, the ones from 64 up as exchanges between a lane's own registers (the loops unroll: every register index
// is a compile-time constant).
template <int Q, typename Load, typename Store>
__device__ __forceinline__ void sort_in_registers(int len, int lane, Load load, Store store)
{
    uint32_t kq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) kq[q] = load(q * kWave + lane, q * kWave + lane < len);
#pragma unroll
    for (int k = 2; k <= Q * kWave; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= kWave) {                                             // partners are registers of the same lane
                const int dq = j / kWave;
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    if ((q & dq) == 0 && (q | dq) < Q) {
                        const bool asc = (((q * kWave) & k) == 0);
                        const uint32_t a = kq[q], b = kq[q | dq];
                        const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
                        kq[q] = asc ? lo : hi;
                        kq[q | dq] = asc ? hi : lo;
                    }
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int e = q * kWave + lane;
                    const bool keep_min = ((e & j) == 0) == ((e & k) == 0);
                    kq[q] = bitonic_pick(kq[q], (uint32_t)__shfl_xor((int)kq[q], j), keep_min);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (q * kWave + lane < len) store(q * kWave + lane, kq[q]);
}

// ... a row of the exclusion CSR: column ids -> positions on the way in (two dependent loads per key, all Q of a lane in flight)
template <int Q>
__device__ __forceinline__ void sort_row_in_registers(const int32_t *__restrict__ col, const int32_t *__restrict__ inv,
                                                      uint32_t *__restrict__ pos, long long sr, int lr, int lane)
{
    sort_in_registers<Q>(lr, lane,
                         [&](int e, bool in) { return in ? (uint32_t)inv[col[sr + e]] : kPosNone; },
                         [&](int e, uint32_t key) { pos[sr + e] = key; });
}

__global__ __launch_bounds__(kBlock) void excl_sort_rows_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                int64_t n_rows, const int32_t *__restrict__ inv, uint32_t *__restrict__ pos,
                                                                const uint8_t *__restrict__ needed, int32_t *__restrict__ huge)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t w0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_w = ((int64_t)gridDim.x * kBlock) >> 6;
    constexpr int S = kExclScan;
    for (int64_t base = w0 * kExclScan; base < n_rows; base += n_w * kExclScan) {
        long long s_l = 0;
        int len_l = 0;
        if (lane < kExclScan && base + lane < n_rows && (!needed || needed[base + lane])) {    // (a row no user of the batch owns: skipped)
            s_l = rowptr[base + lane];
            len_l = (int)(rowptr[base + lane + 1] - s_l);
        }
        if (!__any(len_l > 0)) continue;
        // rows beyond this kernel's classes go on a list for excl_sort_huge_kernel (which used to scan every row length again: 27 us
        // for a handful of rows)
        if (len_l > kExclBig) huge[1 + atomicAdd(huge, 1)] = (int32_t)(base + lane);
        long long s[S];
        int len[S];
        uint32_t key[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            s[i] = __shfl(s_l, i);
            len[i] = __shfl(len_l, i);
        }
#pragma unroll
        for (int i = 0; i < S; ++i) key[i] = lane < len[i] ? (uint32_t)col[s[i] + lane] : kPosNone;
#pragma unroll
        for (int i = 0; i < S; ++i) key[i] = lane < len[i] ? (uint32_t)inv[key[i]] : kPosNone;
#pragma unroll
        for (int k = 2; k <= kExclShort; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
#pragma unroll
                for (int i = 0; i < S; ++i) key[i] = bitonic_pick(key[i], (uint32_t)__shfl_xor((int)key[i], j), keep_min);
            }
#pragma unroll
        for (int i = 0; i < S; ++i)
            if (len[i] <= kExclShort && lane < len[i]) pos[s[i] + lane] = key[i];
        // rows of 65 .. 256 entries (four keys per lane), one at a time
        unsigned long long todo = __ballot(len_l > kExclShort && len_l <= kExclBig);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            sort_row_in_registers<kExclBig / kWave>(col, inv, pos, __shfl(s_l, src), __shfl(len_l, src), lane);
        }
    }
}

// Rows of more than kExclBig entries, from the list excl_sort_rows_kernel left (huge[0] of them, ids behind it): a workgroup per row —
// up to kExclLds entries as register-sorted chunks merged through LDS, beyond that by the bitonic network above in place in HBM (a user
// who excludes a twelfth of a 96 k-item table and more).
__global__ __launch_bounds__(kBlock) void excl_sort_huge_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                const int32_t *__restrict__ inv, uint32_t *__restrict__ pos,
                                                                const int32_t *__restrict__ huge)
{
    __shared__ uint32_t lds[kExclLds];
    const int n_huge = huge[0];                                            // (written by the kernel before this one in the stream)
    for (int idx = blockIdx.x; idx < n_huge; idx += gridDim.x) {
        const int64_t r = huge[1 + idx];
        const long long si = rowptr[r];
        const int li = (int)(rowptr[r + 1] - si);
        if (li <= kExclLds) {
            // chunks of kExclBig entries sorted in registers, a wave each (column id -> position on the way in), left in LDS;
            // then every entry's place = its place in its own chunk + the entries of the other chunks that sort before it
            // (binary searches in LDS; an equal key of an earlier chunk goes first).  (A barrier-per-step bitonic network over
            // the whole row took 50 us for one row of 1 100 entries.)
            const int n_chunks = (li + kExclBig - 1) / kExclBig;
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
            for (int c = wave; c < n_chunks; c += kBlock / kWave) {
                const int c0 = c * kExclBig, cl = li - c0 < kExclBig ? li - c0 : kExclBig;
                sort_in_registers<kExclBig / kWave>(cl, lane,
                                                    [&](int e, bool in) { return in ? (uint32_t)inv[col[si + c0 + e]] : kPosNone; },
                                                    [&](int e, uint32_t key) { lds[c0 + e] = key; });
            }
            __syncthreads();
            for (int e = threadIdx.x; e < li; e += kBlock) {
                const uint32_t key = lds[e];
                const int c = e / kExclBig;
                int rank = e - c * kExclBig;
                for (int c2 = 0; c2 < n_chunks; ++c2) {
                    if (c2 == c) continue;
                    const int c0 = c2 * kExclBig, cl = li - c0 < kExclBig ? li - c0 : kExclBig;
                    int lo = 0, hi = cl;                              // entries of chunk c2 before `key`: < key, or <= key if c2 < c
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const uint32_t v = lds[c0 + mid];
                        if (v < key || (v == key && c2 < c)) lo = mid + 1; else hi = mid;
                    }
                    rank += lo;
                }
                pos[si + rank] = key;
            }
        } else {
            uint32_t *buf = pos + si;
            for (int e = threadIdx.x; e < li; e += kBlock) buf[e] = (uint32_t)inv[col[si + e]];
            __syncthreads();
            bitonic_ascending<kBlock>(buf, li, (int)threadIdx.x, [] { __syncthreads(); });
        }
        __syncthreads();                                              // the next row re-uses the buffer
    }
}

int topk_order_layout(int64_t n_items, int64_t excl_rows, int64_t excl_nnz, TopkOrderLayout *L)
{
    if (n_items < 1 || n_items >= ((int64_t)1 << 31) || excl_rows < 0 || excl_nnz < 0 || excl_nnz >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    int64_t off = 0;
    L->bins = off; off += kOrderBinsBytes;                                       // (first: see topk_order.h)
    L->totals = off; off += 256;                                                 // items per group of 1 024 bins
    L->status = off; off += 256;                                                 // [0]: 0 = ordered by norm; 1 = the bins did not arrive zeroed, identity order taken; 2 = a position out of range
    L->norm2 = off; off += al256(n_items * 4);
    L->keys = off; off += al256(n_items * 4);
    L->perm = off; off += al256(n_items * 4);
    L->inv = off; off += al256(n_items * 4);
    L->needed = off; off += excl_nnz > 0 ? al256(excl_rows) : 0;                // needed[row]: the rows this call's users own
    L->huge = off; off += excl_nnz > 0 ? al256((excl_rows + 1) * 4) : 0;        // [count][ids of rows longer than kExclBig]
    L->excl_pos = off; off += al256(excl_nnz * 4);
    L->total = off;
    return IGCN_OK;
}

int topk_order_build(const TopkOrderLayout &L, char *ws, int64_t n_items, const int64_t *excl_rowptr, const int32_t *excl_col,
                     int64_t excl_rows, int64_t excl_nnz, const int64_t *user_ids, int64_t batch, hipStream_t st,
                     const int32_t **perm_out, const int32_t **excl_pos_out)
{
    const float *norm2 = reinterpret_cast<const float *>(ws + L.norm2);
    uint32_t *keys = reinterpret_cast<uint32_t *>(ws + L.keys), *bins = reinterpret_cast<uint32_t *>(ws + L.bins);
    int32_t *perm = reinterpret_cast<int32_t *>(ws + L.perm), *inv = reinterpret_cast<int32_t *>(ws + L.inv);
    const unsigned ib = (unsigned)((n_items + kBlock - 1) / kBlock);
    int32_t *huge = excl_nnz > 0 ? reinterpret_cast<int32_t *>(ws + L.huge) : nullptr;
    hipLaunchKernelGGL(order_count_kernel, dim3(ib), dim3(kBlock), 0, st, norm2, n_items, keys, bins, huge);
    uint32_t *totals = reinterpret_cast<uint32_t *>(ws + L.totals);
    uint32_t *status = reinterpret_cast<uint32_t *>(ws + L.status);
    hipLaunchKernelGGL(order_scan_kernel, dim3(kOrderBins / 1024), dim3(1024), 0, st, bins, totals, status);
    hipLaunchKernelGGL(order_place_kernel, dim3(ib), dim3(kBlock), 0, st, (const uint32_t *)keys, n_items, bins, (const uint32_t *)totals, perm, inv, status);
    *perm_out = perm;
    *excl_pos_out = nullptr;
    if (excl_nnz > 0) {
        if (excl_rows >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
        uint32_t *pos = reinterpret_cast<uint32_t *>(ws + L.excl_pos);
        const int64_t cus = cu_count();
        // Only the rows of this call's users are sorted (a 512-user call against a 110 k-user CSR used to sort all of it: ADVICE r3):
        // unless the batch IS the CSR's rows in order, a kernel over the batch flags them first.
        const uint8_t *needed = nullptr;
        if (user_ids || batch < excl_rows) {
            uint8_t *flags = reinterpret_cast<uint8_t *>(ws + L.needed);
            const int me = zero_async(flags, (size_t)excl_rows, st);          // (a kernel, not a memset node: common.h)
            if (me != IGCN_OK) return me;
            hipLaunchKernelGGL(excl_mark_rows_kernel, dim3((unsigned)((batch + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, user_ids, batch,
                               excl_rows, flags);
            needed = flags;
        }
        const int64_t turns = (excl_rows + kExclScan - 1) / kExclScan;          // a wave per kExclScan rows
        int64_t rb = (turns + kBlock / kWave - 1) / (kBlock / kWave);
        if (rb > 8 * cus) rb = 8 * cus;
        hipLaunchKernelGGL(excl_sort_rows_kernel, dim3((unsigned)rb), dim3(kBlock), 0, st, excl_rowptr, excl_col, excl_rows,
                           (const int32_t *)inv, pos, needed, huge);
        int64_t hb = excl_rows < 5 * cus ? excl_rows : 5 * cus;                 // a workgroup per listed row (32 KiB of LDS each: five per CU); most leave at once
        hipLaunchKernelGGL(excl_sort_huge_kernel, dim3((unsigned)hb), dim3(kBlock), 0, st, excl_rowptr, excl_col,
                           (const int32_t *)inv, pos, (const int32_t *)huge);
        *excl_pos_out = reinterpret_cast<const int32_t *>(pos);
    }
    return launch_status();
}

}  // namespace igcn
