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
// reduced-precision shortcut).  Layout of one workgroup (WAVES x 64 lanes):
//   * every wave owns 32 users for the whole item sweep; their embeddings are
//     the MFMA B operand and stay in D/2 VGPRs per lane;
//   * items stream through LDS in tiles of 64 rows (register-staged, double
//     buffered, rows padded by 16 B so the ds_read_b128 fragment reads are
//     bank-conflict free); each wave multiplies two 32x32xD sub-tiles per tile;
//   * the product is computed as S^T = I . U^T, so in the accumulator a lane
//     holds 16 item scores of ONE user (column = lane&31): the running top-k of
//     a user is private to a lane pair, no cross-lane traffic in the sweep;
//   * masking is exact and in-register: each lane walks its user's sorted
//     exclusion list with a cursor as the item sweep advances; banned items
//     arrive as a 0/-inf bias staged with the tile;
//   * top-k: one compare of the tile maximum against the user's current k-th
//     best decides whether anything can enter.  Entries are 64-bit sortable keys
//     (order-preserving image of the fp32 score << 32 | ~item id), kept per lane
//     as a k-slot binary min-heap in LDS ([slot][lane]: lane l always hits its own
//     bank pair); a rare insert replaces the root and sifts down, O(log k).  The
//     wave handles "the first remaining candidate of every lane" per pass, so a
//     tile costs about one pass however its candidates are spread over lanes;
//   * when the batch has too few 32-user groups to fill 256 CUs, the item range
//     is split across workgroups and a small kernel merges the partial lists.
// Ties are broken towards the lower item id (torch.topk leaves them unspecified).
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace igcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileItems = 64;            // items per staged tile
constexpr int kIdxNone = 0x7fffffff;

struct TopkPlan {
    int waves;               // waves per workgroup (32 users each)
    int d_pad;               // 16 / 32 / 64 / 128
    int n_splits;            // item-range splits
    int64_t items_per_split; // multiple of kTileItems
    int64_t user_tiles;
    size_t lds_bytes;
};

static inline size_t topk_lds_bytes(int waves, int d_pad, int k) {
    return (size_t)2 * kTileItems * (d_pad + 4) * 4 + (size_t)2 * kTileItems * 4 + (size_t)waves * k * kWave * 8;
}

static inline int topk_make_plan(int64_t batch, int64_t n_items, int32_t d, int32_t k, TopkPlan *p) {
    if (batch < 1 || n_items < 1) return IGCN_E_SHAPE;
    if (d < 4 || d > 128 || d % 4 != 0) return IGCN_E_SHAPE;
    if (k < 1 || k > IGCN_MAX_TOPK || k > n_items) return IGCN_E_RANGE;
    p->d_pad = d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : 128;
    p->waves = topk_lds_bytes(4, p->d_pad, k) <= 160 * 1024 ? 4 : 2;  // <= 80 KiB gives two workgroups per CU
    if (topk_lds_bytes(p->waves, p->d_pad, k) > 160 * 1024) return IGCN_E_RANGE;
    p->lds_bytes = topk_lds_bytes(p->waves, p->d_pad, k);
    const int64_t users_per_wg = 32 * p->waves;
    p->user_tiles = (batch + users_per_wg - 1) / users_per_wg;
    // Item-range splits only when the user tiles alone cannot fill the chip (2 workgroups per CU):
    // every split keeps its own k-entry lists, so splitting multiplies the insert work.
    const int64_t slots = (int64_t)cu_count() * 2;
    int64_t splits = p->user_tiles >= slots ? 1 : (2 * slots + p->user_tiles - 1) / p->user_tiles;
    if (const char *e = getenv("IGCN_TOPK_SPLITS")) { int x = atoi(e); if (x >= 1) splits = x; }   // developer knob
    const int64_t max_by_items = n_items / 4096 > 1 ? n_items / 4096 : 1;
    if (splits > max_by_items) splits = max_by_items;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    int64_t per = (n_items + splits - 1) / splits;
    per = (per + kTileItems - 1) / kTileItems * kTileItems;
    p->items_per_split = per;
    p->n_splits = (int)((n_items + per - 1) / per);
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

template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void score_topk_kernel(
    const float *__restrict__ user_rows, int64_t ldu, const int64_t *__restrict__ user_ids, int64_t batch,
    const float *__restrict__ item_rows, int64_t ldi, int64_t n_items, int d,
    const int64_t *__restrict__ excl_rowptr, const int32_t *__restrict__ excl_col, const uint8_t *__restrict__ banned,
    int k, int n_splits, int64_t items_per_split,
    int64_t *__restrict__ out_idx, float *__restrict__ out_val, float *__restrict__ ws_val, int32_t *__restrict__ ws_idx)
{
    constexpr int BLOCK = WAVES * kWave;
    constexpr int STRIDE = D + 4;                            // floats per LDS item row (16 B pad)
    constexpr int NLOAD = kTileItems * (D / 4) / BLOCK;      // float4 per thread per tile
    static_assert(NLOAD >= 1 && kTileItems * (D / 4) % BLOCK == 0, "tile must divide over the block");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_items = reinterpret_cast<float *>(smem);                   // [2][64][STRIDE]
    float *s_bias = s_items + 2 * kTileItems * STRIDE;                  // [2][64]
    unsigned long long *s_heap = reinterpret_cast<unsigned long long *>(s_bias + 2 * kTileItems);   // [WAVES][k][64]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t ut = blockIdx.x / n_splits;
    const int sp = (int)(blockIdx.x % n_splits);
    const int64_t item_lo = (int64_t)sp * items_per_split;
    const int64_t item_hi = item_lo + items_per_split < n_items ? item_lo + items_per_split : n_items;
    const int64_t b = ut * (32 * WAVES) + wid * 32 + j;
    const bool user_ok = b < batch;
    const int64_t uid = user_ok ? (user_ids ? user_ids[b] : b) : 0;

    // B operand: this lane's user, k-slice [h*D/2, (h+1)*D/2); MFMA step s uses k = h*D/2 + s
    float bfrag[D / 2];
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
        float4 v = f4_zero();
        const int e = h * (D / 2) + 4 * q;
        if (user_ok && e < d) v = *reinterpret_cast<const float4 *>(user_rows + uid * ldu + e);
        bfrag[4 * q + 0] = v.x; bfrag[4 * q + 1] = v.y; bfrag[4 * q + 2] = v.z; bfrag[4 * q + 3] = v.w;
    }

    // exclusion cursor: first excluded item >= item_lo
    int64_t ex_pos = 0, ex_end = 0;
    int ex_next = kIdxNone;
    if (excl_rowptr && user_ok) {
        int64_t lo = excl_rowptr[uid];
        ex_end = excl_rowptr[uid + 1];
        int64_t hi = ex_end;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (excl_col[mid] < item_lo) lo = mid + 1; else hi = mid;
        }
        ex_pos = lo;
        if (ex_pos < ex_end) ex_next = excl_col[ex_pos];
    }

    // running top-k: per-lane min-heap of sortable keys in LDS; root (= k-th best so far) in registers.
    // Key 0 = empty slot: ranks below every real entry, masked (-inf) ones included.
    unsigned long long *heap = s_heap + (wid * k) * kWave + lane;
    for (int s = 0; s < k; ++s) heap[s * kWave] = 0ull;
    unsigned long long root = 0ull;
    float thr = -INFINITY;                                   // score part of the root

    const int64_t n_tiles = (item_hi - item_lo + kTileItems - 1) / kTileItems;
    float4 st[NLOAD];
    float st_bias = 0.f;

    auto stage_load = [&](int64_t t) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = tid + i * BLOCK;
            const int row = idx / (D / 4), c4 = idx % (D / 4);
            const int64_t item = item_lo + t * kTileItems + row;
            st[i] = (item < item_hi && 4 * c4 < d) ? *reinterpret_cast<const float4 *>(item_rows + item * ldi + 4 * c4)
                                                    : f4_zero();
        }
        if (banned && tid < kTileItems) {
            const int64_t item = item_lo + t * kTileItems + tid;
            st_bias = (item < item_hi && banned[item]) ? -INFINITY : 0.f;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = tid + i * BLOCK;
            const int row = idx / (D / 4), c4 = idx % (D / 4);
            *reinterpret_cast<float4 *>(s_items + (buf * kTileItems + row) * STRIDE + 4 * c4) = st[i];
        }
        if (banned && tid < kTileItems) s_bias[buf * kTileItems + tid] = st_bias;
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int64_t t = 0; t < n_tiles; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < n_tiles) stage_load(t + 1);

#pragma unroll
        for (int sub = 0; sub < kTileItems / 32; ++sub) {
            const int64_t tile_base64 = item_lo + t * kTileItems + sub * 32;
            if (tile_base64 >= item_hi) continue;                  // workgroup-uniform
            const int tile_base = (int)tile_base64;

            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *arow = s_items + (buf * kTileItems + sub * 32 + j) * STRIDE + h * (D / 2);
#pragma unroll
            for (int q = 0; q < D / 8; ++q) {
                const float4 a = *reinterpret_cast<const float4 *>(arow + 4 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bfrag[4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bfrag[4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bfrag[4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bfrag[4 * q + 3], acc, 0, 0, 0);
            }

            // --- masking -------------------------------------------------------------
            if (tile_base64 + 32 > item_hi) {                      // ragged last tile
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (tile_base64 + row_of(r, h) >= item_hi) acc[r] = -INFINITY;
            }
            if (banned) {
                const float *bias = s_bias + buf * kTileItems + sub * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + 8 * g);
                    acc[4 * g + 0] += bv.x; acc[4 * g + 1] += bv.y; acc[4 * g + 2] += bv.z; acc[4 * g + 3] += bv.w;
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
                        ex_next = ex_pos < ex_end ? excl_col[ex_pos] : kIdxNone;
                    }
                }
            }

            // --- top-k ---------------------------------------------------------------
            while (true) {
                float m = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
                if (!__any(m >= thr)) break;                         // nothing in this tile can enter any list
                // first remaining candidate of this lane; examined scores become NaN (fmaxf skips NaN)
                unsigned long long cand = 0ull;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = acc[r];
                    if (cand == 0ull && s >= thr) {
                        const unsigned long long kr = make_key(s, tile_base + row_of(r, h));
                        if (kr > root) cand = kr;
                        acc[r] = __uint_as_float(0x7fc00000u);       // examined: NaN is never >= thr again
                    }
                }
                if (cand != 0ull) {
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
        }

        if (t + 1 < n_tiles) stage_store(buf ^ 1);
        __syncthreads();
    }

    // merge the two lanes of a user and emit best-first (k rounds of arg-max over 2k keys)
    __syncthreads();
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
            if (n_splits == 1) {
                out_idx[b * k + r] = bi == kIdxNone ? -1 : bi;
                out_val[b * k + r] = bv;
            } else {
                ws_val[(b * n_splits + sp) * k + r] = bv;
                ws_idx[(b * n_splits + sp) * k + r] = bi;
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

template <int D, int WAVES>
static int launch_topk(const TopkPlan &p, hipStream_t st,
                       const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                       const float *item_rows, int64_t ldi, int64_t n_items, int d,
                       const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned, int k,
                       int64_t *out_idx, float *out_val, float *ws_val, int32_t *ws_idx)
{
    auto kern = score_topk_kernel<D, WAVES>;
    static size_t configured = 0;
    if (p.lds_bytes > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(160 * 1024));
        if (e != hipSuccess) return (int)e;
        configured = 160 * 1024;
    }
    const int64_t grid = p.user_tiles * p.n_splits;
    if (grid >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * kWave), p.lds_bytes, st, user_rows, ldu, user_ids, batch,
                       item_rows, ldi, n_items, d, excl_rowptr, excl_col, banned, k, p.n_splits, p.items_per_split,
                       out_idx, out_val, ws_val, ws_idx);
    return launch_status();
}

}  // namespace igcn

using namespace igcn;

extern "C" int64_t igcn_score_topk_workspace_bytes(int64_t batch, int64_t n_items, int32_t d, int32_t k)
{
    TopkPlan p;
    if (topk_make_plan(batch, n_items, d, k, &p) != IGCN_OK) return -1;
    return p.n_splits > 1 ? (int64_t)batch * p.n_splits * k * 8 : 0;
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
    if (p.n_splits > 1 && !workspace) return IGCN_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *ws_val = static_cast<float *>(workspace);
    int32_t *ws_idx = reinterpret_cast<int32_t *>(ws_val ? ws_val + (int64_t)batch * p.n_splits * k : nullptr);

#define IGCN_TOPK_CASE(DD, WW)                                                                                        \
    rc = launch_topk<DD, WW>(p, st, user_rows, ldu, user_ids, batch, item_rows, ldi, n_items, (int)d, excl_rowptr,    \
                             excl_col, banned, (int)k, out_idx, out_val, ws_val, ws_idx)
    if (p.waves == 4) {
        switch (p.d_pad) {
        case 16: IGCN_TOPK_CASE(16, 4); break;
        case 32: IGCN_TOPK_CASE(32, 4); break;
        case 64: IGCN_TOPK_CASE(64, 4); break;
        default: IGCN_TOPK_CASE(128, 4); break;
        }
    } else {
        switch (p.d_pad) {
        case 16: IGCN_TOPK_CASE(16, 2); break;
        case 32: IGCN_TOPK_CASE(32, 2); break;
        case 64: IGCN_TOPK_CASE(64, 2); break;
        default: IGCN_TOPK_CASE(128, 2); break;
        }
    }
#undef IGCN_TOPK_CASE
    if (rc != IGCN_OK) return rc;
    if (p.n_splits > 1) {
        const int64_t blocks = (batch + 3) / 4;
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, ws_val, ws_idx, batch,
                           p.n_splits, (int)k, out_idx, out_val);
        rc = launch_status();
    }
    return rc;
}

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
