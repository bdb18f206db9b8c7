// Fused  score = U . I^T  ->  mask  ->  top-k  for MI355X (gfx950).
//
// Replaces, without ever materialising the [B, n_items] score matrix
// (Amazon-book: 42 GB for one full evaluation):
//   torch.mm(users_r, all_items_r.t())          model.py:120-122 / :70-71
//   scores[excl_u, excl_i] = -inf ; banned      trainer.py:149-161
//   torch.topk(scores, k)                       trainer.py:163
// and the membership loop of calculate_metrics  trainer.py:111-115 (igcn_hit_matrix).
//
// This is the one dense contraction of the path, so it runs on the matrix
// cores, in exact fp32: v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, no
// reduced-precision shortcut).  The fp32 MFMA is slow enough (64 cycles each,
// 2048 cycles per 32x32xd=64 tile) that operand traffic is negligible; what
// decides the speed is keeping every SIMD's matrix pipe busy.  Hence:
//   * one WAVE = one workgroup = 32 users for a whole item sweep: no LDS staging,
//     no barriers, nothing shared between waves.  The users' embeddings are the
//     MFMA B operand and stay in d/2 VGPRs per lane; the item rows (A operand)
//     are read straight from L2/Infinity Cache, the two lanes of a row taking
//     adjacent 16-byte pieces so that a load instruction touches 32 lines;
//   * <=128 VGPRs and k*512 B of LDS per wave, so 4 waves share a SIMD and cover
//     each other's loads and top-k bookkeeping with their MFMA chains; the waves
//     of a SIMD take different static priorities (by hardware wave slot) so that
//     they fall out of lock-step instead of all multiplying, then all selecting;
//   * the grid is sized so that ALL waves are resident at once and every SIMD
//     gets the same amount of MFMA work: G = ceil(B/32) user groups over S SIMDs;
//     floor(G/S)*S groups are swept by one wave each, the remaining groups are
//     cut into item-range parts (one wave per part) that fill the last wave slot
//     of every SIMD; parts write partial lists that a small kernel merges;
//   * the product is computed as S^T = I . U^T, so in the accumulator a lane
//     holds 16 item scores of ONE user (column = lane&31): the running top-k of
//     a user is private to a lane pair, no cross-lane traffic in the sweep;
//   * masking is exact and in-register: each lane walks its user's sorted
//     exclusion list with a cursor as the item sweep advances; banned items are
//     read as bytes per accumulator row;
//   * top-k: one compare of the tile maximum against the user's current k-th
//     best decides whether anything can enter.  Entries are 64-bit sortable keys
//     (order-preserving image of the fp32 score << 32 | ~item id), kept per lane
//     as a k-slot binary min-heap in LDS ([slot][lane]: lane l always hits its own
//     bank pair); a rare insert replaces the root and sifts down, O(log k).  The
//     wave handles "the first remaining candidate of every lane" per pass, so a
//     tile costs about one pass however its candidates are spread over lanes.
// Ties are broken towards the lower item id (torch.topk leaves them unspecified).
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace igcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kIdxNone = 0x7fffffff;

struct TopkPlan {
    int d_pad;               // 16 / 32 / 64 / 128
    int64_t groups;          // 32-user groups
    int64_t n_full;          // groups swept by a single wave (parts == 1 for them)
    int parts;               // item-range parts of the remaining groups (1 = none split)
    int64_t items_per_part;  // multiple of 32
    int64_t units;           // waves launched
    size_t lds_bytes;
};

static inline int topk_make_plan(int64_t batch, int64_t n_items, int32_t d, int32_t k, TopkPlan *p) {
    if (batch < 1 || n_items < 1) return IGCN_E_SHAPE;
    if (d < 4 || d > 128 || d % 4 != 0) return IGCN_E_SHAPE;
    if (k < 1 || k > IGCN_MAX_TOPK || k > n_items) return IGCN_E_RANGE;
    p->d_pad = d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : 128;
    p->lds_bytes = (size_t)k * kWave * 8;
    p->groups = (batch + 31) / 32;
    const int64_t simds = (int64_t)cu_count() * 4;
    // waves that can be resident per CU: 4 (3 for d > 64) per SIMD by registers, 160 KiB / lds by LDS
    int64_t per_cu = (160 * 1024) / (int64_t)p->lds_bytes;
    const int64_t by_regs = p->d_pad <= 64 ? 16 : 8;         // 114 / 179 VGPRs (hipcc, gfx950)
    if (per_cu > by_regs) per_cu = by_regs;
    if (per_cu < 1) return IGCN_E_RANGE;
    const int64_t slots = per_cu * cu_count();
    int64_t n_full, parts;
    if (p->groups >= simds) {
        // every SIMD gets floor(G/S) whole sweeps; the rest is cut so that each SIMD gets one part more
        n_full = (p->groups / simds) * simds;
        const int64_t rest = p->groups - n_full;
        parts = rest == 0 ? 1 : (simds + rest - 1) / rest;
        if (rest * 10 > simds * 9) { parts = 1; }              // nearly a whole extra round anyway
    } else {
        n_full = 0;                                            // small batch: split every group to fill the chip
        parts = (slots + p->groups - 1) / p->groups;
    }
    const int64_t max_by_items = n_items / 1024 > 1 ? n_items / 1024 : 1;
    if (parts > max_by_items) parts = max_by_items;
    if (parts > 64) parts = 64;
    if (const char *e = getenv("IGCN_TOPK_PARTS")) { int x = atoi(e); if (x >= 1 && x <= 64) parts = x; }   // developer knob
    if (parts <= 1) { parts = 1; n_full = p->groups; }
    int64_t per = (n_items + parts - 1) / parts;
    per = (per + 31) / 32 * 32;
    p->parts = (int)((n_items + per - 1) / per);
    if (p->parts <= 1) { p->parts = 1; n_full = p->groups; }
    p->items_per_part = per;
    p->n_full = n_full;
    p->units = n_full + (p->groups - n_full) * p->parts;
    return IGCN_OK;
}

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Sortable 64-bit key: larger key = ranks earlier (higher score, then lower item id).
__device__ __forceinline__ unsigned long long make_key(float s, int item) {
    unsigned int u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // order-preserving map of fp32 to uint32
    return ((unsigned long long)u << 32) | (unsigned int)(~item);
}
__device__ __forceinline__ float key_score(unsigned long long key) {
    unsigned int u = (unsigned int)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ int key_item(unsigned long long key) { return (int)~(unsigned int)key; }

// (v, i) ranks before (w, j): higher score first, then lower item id
__device__ __forceinline__ bool ranks_before(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

// Static issue priority from the hardware wave slot: the 3-4 waves of a SIMD get different
// priorities, so one of them always wins the matrix pipe and the others fill in behind it.
__device__ __forceinline__ void set_priority_by_wave_slot() {
    // s_getreg_b32 HW_REG_HW_ID (id 4), WAVE_ID = bits [3:0]: simm16 = (size-1) << 11 | offset << 6 | id
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3u;
    if (slot == 0) __builtin_amdgcn_s_setprio(0);
    else if (slot == 1) __builtin_amdgcn_s_setprio(1);
    else if (slot == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

#ifdef IGCN_TOPK_TRACE
// Developer build only (scripts/dev_topk_trace.py): shader-clock cycles per phase, summed over waves.
// [0] load wait  [1] MFMA chain  [2] masking  [3] selection  [4] whole wave  [5] tiles  [6] waves
__device__ unsigned long long g_topk_trace[8];
// clock read that cannot issue before `dep` (an SGPR derived from the results being timed) exists
__device__ __forceinline__ unsigned long long trace_clock(int dep) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(dep) : "memory");
    return t;
}
__device__ __forceinline__ int trace_dep(const f32x16 &acc) {
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
    return __builtin_amdgcn_readfirstlane(__float_as_int(m));
}
#endif

template <int D>
__global__ __launch_bounds__(kWave, (D <= 64 ? 4 : 2)) void score_topk_kernel(
    const float *__restrict__ user_rows, int64_t ldu, const int64_t *__restrict__ user_ids, int64_t batch,
    const float *__restrict__ item_rows, int64_t ldi, int64_t n_items, int d,
    const int64_t *__restrict__ excl_rowptr, const int32_t *__restrict__ excl_col, const uint8_t *__restrict__ banned,
    int k, int64_t n_full, int parts, int64_t items_per_part, int stagger,
    int64_t *__restrict__ out_idx, float *__restrict__ out_val, float *__restrict__ ws_val, int32_t *__restrict__ ws_idx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *heap = reinterpret_cast<unsigned long long *>(smem) + threadIdx.x;     // [k][64]

    if (stagger) set_priority_by_wave_slot();
#ifdef IGCN_TOPK_TRACE
    unsigned long long tr_load = 0, tr_chain = 0, tr_mask = 0, tr_sel = 0, tr_tiles = 0;
    const unsigned long long tr_begin = trace_clock(0);
#endif
    const int lane = threadIdx.x;
    const int j = lane & 31, h = lane >> 5;
    // unit -> (user group, item part)
    int64_t group;
    int part = 0, my_parts = 1;
    if ((int64_t)blockIdx.x < n_full) {
        group = blockIdx.x;
    } else {
        const int64_t v = (int64_t)blockIdx.x - n_full;
        group = n_full + v / parts;
        part = (int)(v % parts);
        my_parts = parts;
    }
    const int item_lo = my_parts == 1 ? 0 : (int)(part * items_per_part);
    const int item_hi = my_parts == 1 ? (int)n_items : (int)(item_lo + items_per_part < n_items ? item_lo + items_per_part : n_items);
    const int64_t b = group * 32 + j;
    const bool user_ok = b < batch;
    const int64_t uid = user_ok ? (user_ids ? user_ids[b] : b) : 0;

    // B operand: this lane's user.  Lane half h supplies k = 8q + 4h + c (q < D/8, c < 4): the two
    // lanes of a row read adjacent 16-B pieces, so one load instruction touches 32 lines, not 64.
    float bfrag[D / 2];
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
        float4 v = f4_zero();
        const int e = 8 * q + 4 * h;
        if (user_ok && e < d) v = *reinterpret_cast<const float4 *>(user_rows + uid * ldu + e);
        bfrag[4 * q + 0] = v.x; bfrag[4 * q + 1] = v.y; bfrag[4 * q + 2] = v.z; bfrag[4 * q + 3] = v.w;
    }

    // exclusion cursor: first excluded item >= item_lo (this user's list as a pointer + 32-bit cursor)
    const int32_t *ex_ptr = excl_col;
    int ex_pos = 0, ex_end = 0, ex_next = kIdxNone;
    if (excl_rowptr && user_ok) {
        const int64_t r0 = excl_rowptr[uid];
        ex_ptr = excl_col + r0;
        ex_end = (int)(excl_rowptr[uid + 1] - r0);
        int lo = 0, hi = ex_end;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ex_ptr[mid] < item_lo) lo = mid + 1; else hi = mid;
        }
        ex_pos = lo;
        if (ex_pos < ex_end) ex_next = ex_ptr[ex_pos];
    }

    // running top-k: per-lane min-heap of sortable keys in LDS; root (= k-th best so far) in registers.
    // Key 0 = empty slot: ranks below every real entry, masked (-inf) ones included.
    for (int s = 0; s < k; ++s) heap[s * kWave] = 0ull;
    unsigned long long root = 0ull;
    float thr = -INFINITY;                                   // score part of the root

    for (int tile_base = item_lo; tile_base < item_hi; tile_base += 32) {
        // A operand: item row (clamped at the ragged end, masked below), k-slice of this lane half
#ifdef IGCN_TOPK_TRACE
        const unsigned long long tr0 = trace_clock(tile_base);
#endif
        int arow_i = tile_base + j;
        if (arow_i >= item_hi) arow_i = item_hi - 1;
        const float *arow = item_rows + (int64_t)arow_i * ldi + 4 * h;
        float4 a[D / 8];
#pragma unroll
        for (int q = 0; q < D / 8; ++q)
            a[q] = (8 * q + 4 * h < d) ? *reinterpret_cast<const float4 *>(arow + 8 * q) : f4_zero();
#ifdef IGCN_TOPK_TRACE
        unsigned long long tr1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr1), "+v"(a[0].x) : : "memory");
#endif

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int q = 0; q < D / 8; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bfrag[4 * q + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bfrag[4 * q + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bfrag[4 * q + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bfrag[4 * q + 3], acc, 0, 0, 0);
        }

#ifdef IGCN_TOPK_TRACE
        const unsigned long long tr2 = trace_clock(trace_dep(acc));
#endif
        // --- masking -------------------------------------------------------------
        if (tile_base + 32 > item_hi) {                        // ragged last tile (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (tile_base + row_of(r, h) >= item_hi) acc[r] = -INFINITY;
        }
        if (banned) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int it = tile_base + 8 * g + 4 * h + c;
                    if (it < item_hi && banned[it]) acc[4 * g + c] = -INFINITY;
                }
            }
        }
        if (excl_rowptr) {
            const int tile_end = tile_base + 32;
            while (true) {
                const bool need = ex_next < tile_end;
                if (!__any(need)) break;
                if (need) {
                    const int rl = ex_next - tile_base;          // 0..31
                    if (((rl >> 2) & 1) == h) {
                        const int rr = (rl & 3) + 4 * (rl >> 3);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = (r == rr) ? -INFINITY : acc[r];
                    }
                    ++ex_pos;
                    ex_next = ex_pos < ex_end ? ex_ptr[ex_pos] : kIdxNone;
                }
            }
        }

#ifdef IGCN_TOPK_TRACE
        const unsigned long long tr3 = trace_clock(trace_dep(acc));
#endif
        // --- top-k ---------------------------------------------------------------
        while (true) {
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
            if (!__any(m >= thr)) break;                         // nothing in this tile can enter any list
            // first remaining candidate of this lane (float compare only); examined scores become
            // NaN (fmaxf skips NaN, NaN >= thr is false)
            float cs = 0.f;
            int cr = -1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s = acc[r];
                const bool take = cr < 0 && s >= thr;
                cs = take ? s : cs;
                cr = take ? row_of(r, h) : cr;
                acc[r] = take ? __uint_as_float(0x7fc00000u) : s;
            }
            const unsigned long long cand = cr >= 0 ? make_key(cs, tile_base + cr) : 0ull;
            if (cand > root) {                                   // (a tie on the score may still lose on the id)
                // replace the root (worst entry) and sift down
                int i = 0;
                unsigned long long first_up = 0ull;
                while (true) {
                    int c = 2 * i + 1;
                    if (c >= k) break;
                    unsigned long long kc = heap[c * kWave];
                    if (c + 1 < k) {
                        const unsigned long long k2 = heap[(c + 1) * kWave];
                        if (k2 < kc) { kc = k2; ++c; }
                    }
                    if (kc >= cand) break;
                    heap[i * kWave] = kc;
                    if (i == 0) first_up = kc;
                    i = c;
                }
                heap[i * kWave] = cand;
                root = i == 0 ? cand : first_up;
                thr = root ? key_score(root) : -INFINITY;    // heap not full yet: everything may enter
            }
        }
#ifdef IGCN_TOPK_TRACE
        const unsigned long long tr4 = trace_clock(__builtin_amdgcn_readfirstlane(__float_as_int(thr)));
        tr_load += tr1 - tr0; tr_chain += tr2 - tr1; tr_mask += tr3 - tr2; tr_sel += tr4 - tr3; ++tr_tiles;
#endif
    }
#ifdef IGCN_TOPK_TRACE
    if (lane == 0) {
        atomicAdd(&g_topk_trace[0], tr_load); atomicAdd(&g_topk_trace[1], tr_chain);
        atomicAdd(&g_topk_trace[2], tr_mask); atomicAdd(&g_topk_trace[3], tr_sel);
        atomicAdd(&g_topk_trace[4], trace_clock(0) - tr_begin); atomicAdd(&g_topk_trace[5], tr_tiles);
        atomicAdd(&g_topk_trace[6], 1ull);
    }
#endif

    // merge the two lanes of a user and emit best-first (k rounds of arg-max over 2k keys).
    // The heaps are private to this wave; a wave executes its LDS operations in order.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (h == 0 && user_ok) {
        unsigned long long *pheap = heap + 32;      // partner lane (l + 32), same wave
        for (int r = 0; r < k; ++r) {
            unsigned long long best = 0ull;
            int bp = 0, bwho = 0;
            for (int q = 0; q < k; ++q) {
                const unsigned long long k0 = heap[q * kWave], k1 = pheap[q * kWave];
                if (k0 > best) { best = k0; bp = q; bwho = 0; }
                if (k1 > best) { best = k1; bp = q; bwho = 1; }
            }
            if (bwho == 0) heap[bp * kWave] = 0ull; else pheap[bp * kWave] = 0ull;
            const float bv = best ? key_score(best) : -INFINITY;
            const int bi = best ? key_item(best) : kIdxNone;
            if (parts == 1) {
                out_idx[b * k + r] = bi == kIdxNone ? -1 : bi;
                out_val[b * k + r] = bv;
            } else {
                ws_val[(b * parts + part) * k + r] = bv;
                ws_idx[(b * parts + part) * k + r] = bi;
                if (my_parts == 1)                               // whole sweep by one wave: other part slots are empty
                    for (int pp = 1; pp < parts; ++pp) {
                        ws_val[(b * parts + pp) * k + r] = -INFINITY;
                        ws_idx[(b * parts + pp) * k + r] = kIdxNone;
                    }
            }
        }
    }
}

// One wave per user, lane = item-range split: k rounds of a wave-wide arg-best
// over the heads of the (already best-first) partial lists.
__global__ __launch_bounds__(kBlock) void topk_merge_kernel(const float *__restrict__ ws_val, const int32_t *__restrict__ ws_idx,
                                                            int64_t batch, int n_splits, int k,
                                                            int64_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t b = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (b >= batch) return;
    const float *v = ws_val + (b * n_splits + lane) * k;
    const int32_t *ix = ws_idx + (b * n_splits + lane) * k;
    int cur = 0;
    float hv = -INFINITY;
    int hi = kIdxNone;
    if (lane < n_splits) { hv = v[0]; hi = ix[0]; }
    for (int r = 0; r < k; ++r) {
        float bv = hv;
        int bi = hi, bl = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off), ol = __shfl_xor(bl, off);
            if (ranks_before(ov, oi, bv, bi) || (ov == bv && oi == bi && ol < bl)) { bv = ov; bi = oi; bl = ol; }
        }
        if (lane == 0) { out_idx[b * k + r] = bi == kIdxNone ? -1 : bi; out_val[b * k + r] = bv; }
        if (lane == bl && lane < n_splits) {
            ++cur;
            if (cur < k) { hv = v[cur]; hi = ix[cur]; } else { hv = -INFINITY; hi = kIdxNone; }
        }
    }
}

__global__ void hit_matrix_kernel(const int64_t *__restrict__ rec, int64_t n_users, int k,
                                  const int64_t *__restrict__ eval_rowptr, const int32_t *__restrict__ eval_col,
                                  float *__restrict__ hit)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_users * k) return;
    const int64_t u = i / k;
    const int64_t item = rec[i];
    int64_t lo = eval_rowptr[u], hi = eval_rowptr[u + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (eval_col[mid] < item) lo = mid + 1; else hi = mid;
    }
    hit[i] = (lo < eval_rowptr[u + 1] && eval_col[lo] == item) ? 1.f : 0.f;
}

template <int D>
static int launch_topk(const TopkPlan &p, hipStream_t st,
                       const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                       const float *item_rows, int64_t ldi, int64_t n_items, int d,
                       const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned, int k,
                       int64_t *out_idx, float *out_val, float *ws_val, int32_t *ws_idx)
{
    auto kern = score_topk_kernel<D>;
    static bool configured = false;
    if (!configured && p.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(64 * 1024));
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    if (p.units >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    static const int stagger = [] { const char *e = getenv("IGCN_TOPK_STAGGER"); return e ? atoi(e) : 1; }();   // developer knob
    hipLaunchKernelGGL(kern, dim3((unsigned)p.units), dim3(kWave), p.lds_bytes, st, user_rows, ldu, user_ids, batch,
                       item_rows, ldi, n_items, d, excl_rowptr, excl_col, banned, k, p.n_full, p.parts, p.items_per_part,
                       stagger, out_idx, out_val, ws_val, ws_idx);
    return launch_status();
}

}  // namespace igcn

using namespace igcn;

extern "C" int64_t igcn_score_topk_workspace_bytes(int64_t batch, int64_t n_items, int32_t d, int32_t k)
{
    TopkPlan p;
    if (topk_make_plan(batch, n_items, d, k, &p) != IGCN_OK) return -1;
    return p.parts > 1 ? (int64_t)batch * p.parts * k * 8 : 0;
}

extern "C" int igcn_score_topk_f32(const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                                   const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                                   const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                                   int32_t k, int64_t *out_idx, float *out_val, void *workspace, void *stream)
{
    if (!user_rows || !item_rows || !out_idx || !out_val) return IGCN_E_NULL;
    if ((excl_rowptr == nullptr) != (excl_col == nullptr)) return IGCN_E_NULL;
    TopkPlan p;
    int rc = topk_make_plan(batch, n_items, d, k, &p);
    if (rc != IGCN_OK) return rc;
    if (ldu < d || ldi < d || ldu % 4 || ldi % 4 || n_items >= ((int64_t)1 << 31) - 64) return IGCN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(user_rows) | reinterpret_cast<uintptr_t>(item_rows)) % 16) return IGCN_E_ALIGN;
    if (p.parts > 1 && !workspace) return IGCN_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *ws_val = static_cast<float *>(workspace);
    int32_t *ws_idx = reinterpret_cast<int32_t *>(ws_val ? ws_val + (int64_t)batch * p.parts * k : nullptr);

#define IGCN_TOPK_CASE(DD)                                                                                        \
    rc = launch_topk<DD>(p, st, user_rows, ldu, user_ids, batch, item_rows, ldi, n_items, (int)d, excl_rowptr,    \
                         excl_col, banned, (int)k, out_idx, out_val, ws_val, ws_idx)
    switch (p.d_pad) {
    case 16: IGCN_TOPK_CASE(16); break;
    case 32: IGCN_TOPK_CASE(32); break;
    case 64: IGCN_TOPK_CASE(64); break;
    default: IGCN_TOPK_CASE(128); break;
    }
#undef IGCN_TOPK_CASE
    if (rc != IGCN_OK) return rc;
    if (p.parts > 1) {
        const int64_t blocks = (batch + 3) / 4;
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, ws_val, ws_idx, batch,
                           p.parts, (int)k, out_idx, out_val);
        rc = launch_status();
    }
    return rc;
}

#ifdef IGCN_TOPK_TRACE
extern "C" int igcn_debug_topk_trace(unsigned long long *host8, int reset)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && host8) e = hipMemcpyFromSymbol(host8, HIP_SYMBOL(igcn::g_topk_trace), 64);
    if (e == hipSuccess && reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(igcn::g_topk_trace), z, 64);
    }
    return (int)e;
}
#endif

extern "C" int igcn_hit_matrix(const int64_t *rec, int64_t n_users, int32_t k,
                               const int64_t *eval_rowptr, const int32_t *eval_col, float *hit, void *stream)
{
    if (!rec || !eval_rowptr || !hit) return IGCN_E_NULL;
    if (n_users < 0 || k < 1) return IGCN_E_SHAPE;
    const int64_t n = n_users * k;
    if (n == 0) return IGCN_OK;
    if (!eval_col) return IGCN_E_NULL;
    hipLaunchKernelGGL(hit_matrix_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), rec, n_users, (int)k, eval_rowptr, eval_col, hit);
    return launch_status();
}
