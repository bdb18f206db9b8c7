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
// reduced-precision shortcut): 64 cycles each, 2048 cycles per 32x32xd=64 tile.
// Operand traffic is negligible; what decides the speed is what a SIMD issues
// BESIDES the MFMAs.  Measured on this chip (scripts/probes/mfma_shadow_probe.hip,
// mfma_valu_overlap_probe.hip): the fp32 MFMA leaves no shadow — every vector
// instruction a wave puts between two dependent MFMAs costs its full ~5 cycles,
// and while one wave runs a dependent MFMA chain a co-resident wave's vector
// instructions do not issue at all.  So the bookkeeping per tile is written to be
// as few instructions as possible, and waves only help to hide memory latency:
//   * one WAVE = one workgroup = 32 users for a run of item tiles: no barriers,
//     nothing shared between waves.  The users' embeddings are the MFMA B operand
//     and stay in d/2 VGPRs per lane; the item rows (A operand) are read straight
//     from L2/Infinity Cache, addressed as a scalar tile base + a constant lane
//     offset; the two lanes of a row take adjacent 16-byte pieces so that a load
//     instruction touches 32 lines;
//   * the tile loop is software-pipelined by hand: while tile t's scores sit in one
//     accumulator, ONE pinned basic block runs the MFMA chain of tile t+1 into the
//     other, re-loads each a[q] for tile t+2 as soon as its four MFMAs have issued
//     (two chains ahead of its use) and stages tile t's candidates, a row or two
//     after every MFMA.  Left alone the compiler clusters the MFMAs and sinks the
//     loads to the end of the chain, one tile too late (+25 % time);
//   * the product is computed as S^T = I . U^T, so in the accumulator a lane
//     holds 16 item scores of ONE user (column = lane&31): the running top-k of
//     a user is private to a lane pair, no cross-lane traffic in the sweep;
//   * top-k entries are 64-bit sortable keys (order-preserving image of the fp32
//     score << 32 | ~item id), kept per lane as a k-slot binary min-heap in LDS
//     ([slot][lane]: lane l always hits its own bank pair).  In the tile loop a row
//     costs 4 vector instructions and one LDS write, no branch: compare with the
//     lane's (slightly stale) k-th best, add-with-carry into the slot counter,
//     clamp, address; (score, item) goes to the lane's next free STAGING slot
//     whether it is a candidate or not.  The replace-root/sift-down work is done
//     for all lanes together when some lane's staging area is full.  A lane that
//     had more candidates than free slots in one tile is rare: the tile's staged
//     entries are then discarded and the tile is redone with per-row branches;
//   * masking: each lane walks its user's sorted exclusion list with a cursor as
//     the item sweep advances; banned items are read as bytes; both only set
//     scores to -inf before the tile is staged;
//   * the grid never exceeds what is resident at once: a wave that starts late
//     runs its whole share after everybody else has finished.  Registers allow 2
//     waves per SIMD (two accumulators + operands: 207 VGPRs); LDS is handed out in
//     1280-B granules, and the staging depth is whatever they leave beside the heap.
//     Every wave gets the same number of tiles: with G 32-user groups and W resident
//     waves each wave sweeps floor(G/W) whole groups, and the tiles of the remaining
//     groups, laid end to end, are cut into W equal runs.  A group that is cut returns
//     one best-first list per (piece, lane half), merged by a small second kernel.
// Ties are broken towards the lower item id (torch.topk leaves them unspecified).
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace igcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- fp32 as the exact sum of three bf16 (MODE 1 of the kernel) -----------------------------------------
// v = h0 + h1 + h2 + r with |r| <= 2^-24 |v|: round to nearest bf16, subtract (exact in fp32), repeat.
__device__ __forceinline__ unsigned int bf16_rne_bits(float v) {
    const unsigned int u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float v, unsigned int &h0, unsigned int &h1, unsigned int &h2) {
    h0 = bf16_rne_bits(v);
    const float r1 = v - __uint_as_float(h0 << 16);
    h1 = bf16_rne_bits(r1);
    const float r2 = r1 - __uint_as_float(h1 << 16);
    h2 = bf16_rne_bits(r2);
}
// eight consecutive fp32 -> three bf16x8 planes (as float4 bit patterns: two bf16 per dword, low half first)
__device__ __forceinline__ void split3_x8(const float4 &lo, const float4 &hi, float4 out[3]) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned int w[3][4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        unsigned int a0, a1, a2, b0, b1, b2;
        split3(v[e], a0, a1, a2);
        split3(v[e + 1], b0, b1, b2);
        w[0][e / 2] = a0 | (b0 << 16); w[1][e / 2] = a1 | (b1 << 16); w[2][e / 2] = a2 | (b2 << 16);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
        out[p] = make_float4(__uint_as_float(w[p][0]), __uint_as_float(w[p][1]), __uint_as_float(w[p][2]), __uint_as_float(w[p][3]));
}
__device__ __forceinline__ bf16x8 as_bf16x8(const float4 &v) { return __builtin_bit_cast(bf16x8, v); }

// Item table -> MFMA-ready bf16 planes: [tile][plane 0..2][k-step 0..3][lane 0..63] x 16 B, lane (j, kg)
// holding k = 16 s + 8 kg .. + 7 of item 32 tile + j.  One thread per (tile, k-step, lane).
__global__ __launch_bounds__(kBlock) void topk_pack_items_kernel(const float *__restrict__ item_rows, int64_t ldi, int64_t n_items,
                                                                 int n_tiles, float4 *__restrict__ packed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_tiles * 4 * kWave) return;
    const int lane = (int)(i % kWave);
    const int s = (int)(i / kWave % 4);
    const int64_t tile = i / (4 * kWave);
    const int64_t item = tile * 32 + (lane & 31);
    float4 lo = f4_zero(), hi = f4_zero();
    if (item < n_items) {
        const float *src = item_rows + item * ldi + 16 * s + 8 * (lane >> 5);
        lo = *reinterpret_cast<const float4 *>(src);
        hi = *reinterpret_cast<const float4 *>(src + 4);
    }
    float4 planes[3];
    split3_x8(lo, hi, planes);
#pragma unroll
    for (int p = 0; p < 3; ++p) packed[((tile * 3 + p) * 4 + s) * kWave + lane] = planes[p];
}

constexpr int kIdxNone = 0x7fffffff;

struct TopkPlan {
    int d_pad;               // 16 / 32 / 64 / 128
    int64_t groups;          // 32-user groups
    int n_tiles;             // 32-item tiles of one sweep
    int64_t units;           // waves launched; all resident at once
    int64_t n_whole;         // whole sweeps per unit: groups [0, n_whole*units), unit u takes u, u+units, ...
    int64_t rest_tiles;      // tiles of the remaining groups (laid end to end), cut into runs of `run`
    int64_t run;             // tiles of the rest that one unit takes
    int p_max;               // bound on the pieces a rest group is cut into (1: runs are whole sweeps)
    int cap;                 // staging slots per lane
    size_t lds_bytes;
};

static inline int env_int(const char *name, int lo, int hi, int dflt) {
    const char *e = getenv(name);
    if (!e) return dflt;
    const int x = atoi(e);
    return x < lo || x > hi ? dflt : x;
}

static inline int topk_make_plan(int64_t batch, int64_t n_items, int32_t d, int32_t k, TopkPlan *p) {
    if (batch < 1 || n_items < 1) return IGCN_E_SHAPE;
    if (d < 4 || d > 128 || d % 4 != 0) return IGCN_E_SHAPE;
    if (k < 1 || k > IGCN_MAX_TOPK || k > n_items) return IGCN_E_RANGE;
    p->d_pad = d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : 128;
    p->groups = (batch + 31) / 32;
    const int64_t L = (n_items + 31) / 32;
    p->n_tiles = (int)L;
    // Resident waves per CU.  Registers allow 2 per SIMD (two accumulators + both operands).  LDS is handed out in
    // granules of 1280 B (1/128 of the CU's 160 KiB; measured: 12 x 13312 B do not fit, 12 x 12800 B do),
    // and the count is kept a multiple of 4 so that every SIMD of a CU carries the same number of
    // waves.  The grid must never exceed what is resident: a wave that starts late runs its whole
    // share after everybody else has finished.
    const int by_regs = 2;
    int per_simd = env_int("IGCN_TOPK_WAVES", 1, by_regs, by_regs);                    // developer knob
    int cap = 0;
    for (; per_simd >= 1; --per_simd) {
        const int granules = 128 / (4 * per_simd);
        cap = granules * 1280 / (kWave * 8) - k - 1;             // staging slots left beside the k list slots and a spare one
        if (cap >= 4) break;
    }
    if (per_simd < 1) return IGCN_E_RANGE;
    if (cap > 16) cap = 16;
    p->cap = env_int("IGCN_TOPK_CAP", 1, cap, cap);                                    // developer knob
    p->lds_bytes = (size_t)(k + p->cap + 1) * kWave * 8;
    const int64_t per_cu = 4 * per_simd;
    int64_t slots = per_cu * cu_count();
    slots = env_int("IGCN_TOPK_SLOTS", 1, (int)slots, (int)slots);                    // developer knob (tests: whole sweeps + cut rest at small sizes)
    int64_t rest;
    if (p->groups >= slots) {
        p->units = slots;
        p->n_whole = p->groups / slots;
        rest = p->groups - p->n_whole * slots;
    } else {
        p->units = 0;
        p->n_whole = 0;
        rest = p->groups;
    }
    p->rest_tiles = rest * L;
    p->run = 0;
    p->p_max = 1;
    if (rest > 0) {
        // a piece shorter than 32 tiles is mostly list warm-up; a group in more than 31 pieces does
        // not fit the 64 lanes of the merge
        int64_t min_run = L < 32 ? L : 32;
        if ((L + 29) / 30 > min_run) min_run = (L + 29) / 30;
        int64_t run = (p->rest_tiles + slots - 1) / slots;
        if (run < min_run) run = min_run;
        p->run = run;
        if (p->units == 0) p->units = (p->rest_tiles + run - 1) / run;
        p->p_max = run % L == 0 ? 1 : (int)((L - 1) / run + 2);
    }
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

// Static issue priority from the hardware wave slot: the waves of a SIMD get different
// priorities, so one of them always wins the matrix pipe and the others fill in behind it.
__device__ __forceinline__ void set_priority_by_wave_slot() {
    // s_getreg_b32 HW_REG_HW_ID (id 4), WAVE_ID = bits [3:0]: simm16 = (size-1) << 11 | offset << 6 | id
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3u;
    if (slot == 0) __builtin_amdgcn_s_setprio(0);
    else if (slot == 1) __builtin_amdgcn_s_setprio(1);
    else if (slot == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// min-heap of sortable keys in LDS, [slot][lane]: replace the root by `cand` and sift down.
// Returns the new root.
__device__ __forceinline__ unsigned long long heap_replace_root(unsigned long long *heap, int n, unsigned long long cand) {
    int i = 0;
    unsigned long long first_up = 0ull;
    while (true) {
        int c = 2 * i + 1;
        if (c >= n) break;
        unsigned long long kc = heap[c * kWave];
        if (c + 1 < n) {
            const unsigned long long k2 = heap[(c + 1) * kWave];
            if (k2 < kc) { kc = k2; ++c; }
        }
        if (kc >= cand) break;
        heap[i * kWave] = kc;
        if (i == 0) first_up = kc;
        i = c;
    }
    heap[i * kWave] = cand;
    return i == 0 ? cand : first_up;
}

#ifdef IGCN_TOPK_TRACE
// Developer build only (scripts/dev_topk_trace.py): shader-clock cycles per phase, summed over waves.
// [0] load wait  [1] MFMA chain  [2] masking  [3] selection  [4] whole wave  [5] tiles  [6] waves
// [7] whole wave in s_memrealtime ticks (100 MHz)
__device__ unsigned long long g_topk_trace[8];
__device__ unsigned long long g_topk_wave_times[4 * 8192];     // [begin, end] in s_memrealtime ticks, HW_ID, reserved per workgroup
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

// FULL: d == D, no k-slice of a row is padding.  MODE 0: fp32 MFMA, the exact fmaf chain.  MODE 1 (D = 64, FULL):
// both operands as three bf16 planes (exact split), 6 of the 9 plane products on the bf16 matrix cores with fp32
// accumulation — products good to 2^-23, i.e. fp32-grade scores, but not the bit pattern of the fmaf chain; the
// item planes come pre-packed (topk_pack_items_kernel).
template <int D, bool FULL, int MODE>
__global__ __launch_bounds__(kWave, 2) void score_topk_kernel(
    const float *__restrict__ user_rows, int64_t ldu, const int64_t *__restrict__ user_ids, int64_t batch,
    const float *__restrict__ item_rows, int64_t ldi, int64_t n_items, int d,
    const int64_t *__restrict__ excl_rowptr, const int32_t *__restrict__ excl_col, const uint8_t *__restrict__ banned,
    int k, int cap, int n_tiles, int64_t n_whole, int64_t rest_tiles, int64_t run, int p_max, int stagger,
    int64_t *__restrict__ out_idx, float *__restrict__ out_val, float *__restrict__ ws_val, int32_t *__restrict__ ws_idx,
    const float4 *__restrict__ packed)
{
    static_assert(MODE == 0 || (D == 64 && FULL), "the split mode is built for d = 64");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *heap = reinterpret_cast<unsigned long long *>(smem) + threadIdx.x;     // [k][64]
    unsigned long long *stage = heap + k * kWave;                                              // [cap][64]

    if (stagger) set_priority_by_wave_slot();
#ifdef IGCN_TOPK_TRACE
    unsigned long long tr_load = 0, tr_chain = 0, tr_mask = 0, tr_sel = 0, tr_tiles = 0;
    const unsigned long long tr_begin = trace_clock(0);
    unsigned long long tr_rt_begin;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_rt_begin) : : "memory");
#endif
    const int lane = threadIdx.x;
    const int j = lane & 31, h = lane >> 5;
    const float kNaN = __uint_as_float(0x7fc00000u);
    const int64_t units = gridDim.x;
    const int64_t n_full = n_whole * units;
    int64_t rx = (int64_t)blockIdx.x * run;                     // cursor in the rest groups' tile space
    const int64_t rx_end = rx + run < rest_tiles ? rx + run : rest_tiles;

    for (int64_t job = 0;; ++job) {
        // ---- next piece: users of `group`, item tiles [tin0, tin1) ----------------------------
        int64_t group;
        int tin0, tin1, pidx = 0;
        bool direct = true;
        if (job < n_whole) {
            group = (int64_t)blockIdx.x + job * units;
            tin0 = 0;
            tin1 = n_tiles;
        } else {
            if (rx >= rx_end) break;
            const int64_t rg = rx / n_tiles;
            group = n_full + rg;
            tin0 = (int)(rx - rg * n_tiles);
            const int64_t left = rx_end - rx;
            tin1 = left < n_tiles - tin0 ? (int)(tin0 + left) : n_tiles;
            pidx = (int)((int64_t)blockIdx.x - (rg * n_tiles) / run);
            direct = p_max == 1;
            rx += tin1 - tin0;
        }
        const int item_lo = tin0 * 32;
        const int item_hi = (int64_t)tin1 * 32 < n_items ? tin1 * 32 : (int)n_items;
        const int64_t b = group * 32 + j;
        const bool user_ok = b < batch;
        const int64_t uid = user_ok ? (user_ids ? user_ids[b] : b) : 0;

        // B operand: this lane's user.  Lane half h supplies k = 8q + 4h + c (q < D/8, c < 4): the two
        // lanes of a row read adjacent 16-B pieces, so one load instruction touches 32 lines, not 64.
        float bfrag[MODE == 0 ? D / 2 : 1];
        float4 ub[MODE == 1 ? 3 : 1][MODE == 1 ? 4 : 1];        // MODE 1: [plane][k-step], 8 bf16 each: k = 16 s + 8 h .. + 7
        if constexpr (MODE == 0) {
#pragma unroll
            for (int q = 0; q < D / 8; ++q) {
                float4 v = f4_zero();
                const int e = 8 * q + 4 * h;
                if (user_ok && (FULL || e < d)) v = *reinterpret_cast<const float4 *>(user_rows + uid * ldu + e);
                bfrag[4 * q + 0] = v.x; bfrag[4 * q + 1] = v.y; bfrag[4 * q + 2] = v.z; bfrag[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                float4 lo = f4_zero(), hi = f4_zero();
                if (user_ok) {
                    const float *src = user_rows + uid * ldu + 16 * st + 8 * h;
                    lo = *reinterpret_cast<const float4 *>(src);
                    hi = *reinterpret_cast<const float4 *>(src + 4);
                }
                float4 planes[3];
                split3_x8(lo, hi, planes);
                ub[0][st] = planes[0]; ub[1][st] = planes[1]; ub[2][st] = planes[2];
            }
        }

        // exclusion cursor: first excluded item >= item_lo; the entry after it is already on its way
        const int32_t *ex_ptr = excl_col;
        int ex_pos = 0, ex_end = 0, ex_next = kIdxNone, ex_after = kIdxNone;
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
            if (ex_pos + 1 < ex_end) ex_after = ex_ptr[ex_pos + 1];
        }

        // running top-k: per-lane min-heap of sortable keys in LDS; root (= k-th best so far) in registers.
        // Key 0 = empty slot: ranks below every real entry, masked (-inf) ones included.
        for (int s = 0; s < k; ++s) heap[s * kWave] = 0ull;
        unsigned long long root = 0ull;
        float thr = -INFINITY;                                   // score part of the root
        int cnt = 0;                                             // staged candidates of this lane

        // staged candidates -> heap, all lanes together
        auto flush = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");  // the hand-written staging stores
            const int n = cnt;
            cnt = 0;
            for (int i = 0; __any(i < n); ++i) {
                if (i < n) {
                    const unsigned long long raw = stage[i * kWave];
                    const unsigned long long cand = make_key(__uint_as_float((unsigned int)raw), (int)(raw >> 32));
                    if (cand > root) root = heap_replace_root(heap, k, cand);
                }
            }
            thr = root ? key_score(root) : -INFINITY;            // list not full yet: everything may enter
        };

        // A operand: item row of this lane, k-slice of its half.  Address = uniform tile base (scalar
        // registers, advanced by scalar adds) + a lane offset that never changes, so a tile's loads
        // cost no vector ALU work.  Rows past the end of the table are clamped (and masked below).
        const int lane_off = j * (int)ldi + 4 * h;
        int lane_off_last = lane_off;                            // for the ragged last tile of the table
        {
            const int64_t last_base = (int64_t)(n_tiles - 1) * 32;
            if (last_base + j >= n_items) lane_off_last = (int)(n_items - 1 - last_base) * (int)ldi + 4 * h;
        }
        float4 a[MODE == 0 ? D / 8 : 12];                        // MODE 1: [plane * 4 + k-step], 8 bf16 each
        auto tile_addr = [&](int t, const float *&tile_ptr, int &off) {
            tile_ptr = item_rows + (int64_t)t * 32 * ldi;
            off = t == n_tiles - 1 ? lane_off_last : lane_off;
        };
        auto load_a = [&](int t) {
            if constexpr (MODE == 0) {
                const float *tile_ptr; int off;
                tile_addr(t, tile_ptr, off);
#pragma unroll
                for (int q = 0; q < D / 8; ++q)
                    a[q] = (FULL || 8 * q + 4 * h < d) ? *reinterpret_cast<const float4 *>(tile_ptr + off + 8 * q) : f4_zero();
            } else {
                // planes 2, 1, 0 — the order in which the pipelined block re-loads them: the compiler sizes the
                // s_waitcnt before each MFMA for the worse of the two ways into the block, and with the planes in
                // storage order here it waited for all but the last 3 loads at the top of every block (+25 % time)
                const float4 *pk = packed + (int64_t)t * 12 * kWave + lane;
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                    for (int st = 0; st < 4; ++st) a[pl * 4 + st] = pk[(pl * 4 + st) * kWave];
            }
        };
        // MODE 1: the six plane products, smallest first: (item plane, user plane)
        constexpr int kTermA[6] = {2, 1, 0, 1, 0, 0};
        constexpr int kTermB[6] = {0, 1, 2, 0, 1, 0};
        // the whole chain of one tile, nothing interleaved (prologue of a piece)
        auto chain_plain = [&](f32x16 &acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int q = 0; q < D / 8; ++q) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bfrag[4 * q + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bfrag[4 * q + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bfrag[4 * q + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bfrag[4 * q + 3], acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int st = 0; st < 4; ++st)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[kTermA[tm] * 4 + st]), as_bf16x8(ub[kTermB[tm]][st]), acc, 0, 0, 0);
            }
        };

        // ---- the parts of a tile's bookkeeping ------------------------------------------------------
        // masks of tile t on its scores (rare events, branches): ragged end, exclusion cursor, banned items
        auto mask_tile = [&](f32x16 &acc, int tile_base) {
            // rows past the end of the piece: NaN = "already examined", never a candidate
            if (tile_base + 32 > item_hi) {                        // ragged last tile (wave-uniform)
                asm volatile("; ragged tile");                     // (a real branch: the common path skips all of this)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (tile_base + row_of(r, h) >= item_hi) acc[r] = kNaN;
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
                        ex_next = ex_after;
                        ex_after = ex_pos + 1 < ex_end ? ex_ptr[ex_pos + 1] : kIdxNone;
                    }
                }
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
        };
        // candidates of a tile -> staging slots, straight-line (predicated LDS writes, no branches), so
        // that it can be scheduled between the MFMAs of the NEXT tile's chain: while the matrix pipe of
        // a SIMD works on a wave's MFMA, the only instructions that issue are that same wave's own
        // independent ones (scripts/probes/mfma_valu_overlap_probe.hip).  Returns whether a lane had a
        // candidate but no free slot (those scores stay unmarked for stage_rows_slow).
        // One row of the fast staging: write (score, item id) to the lane's next free staging slot; a
        // candidate then advances the slot counter.  No branch, no select on the address: slots past
        // `cap` all map to one spare slot.  A lane that ran past it lost candidates; the caller then
        // discards this tile's staged entries and redoes the tile the slow way.
        unsigned int *const stage32 = reinterpret_cast<unsigned int *>(stage);
        const unsigned stage_base = (unsigned)(uintptr_t)stage;  // LDS byte address of this lane's first staging slot
        auto stage_row_fast = [&](float sc, int item, int &slot) {
            // 5 vector instructions and one LDS write per row (everything a wave issues besides MFMAs costs
            // matrix-pipe time on this chip, scripts/probes/mfma_shadow_probe.hip): clamp, address, item id,
            // compare, add-with-carry.  The store is written as ds_write2_b32 by hand: the compiler would
            // build a register pair with a move for ds_write_b64.  flush() waits for these stores itself.
#ifdef IGCN_X_NOSTAGE
            if (sc == 12345.f) slot += item;
#else
            const unsigned w = stage_base + (unsigned)(slot < cap ? slot : cap) * (unsigned)(kWave * 8);
            asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"(w), "v"(sc), "v"(item) : "memory");
            slot += sc >= thr ? 1 : 0;
            asm volatile("" : "+v"(slot));                       // one add-with-carry per row, no re-association
#endif
        };
        // the same with flushes in between, for a tile that overflowed some lane's staging slots
        auto stage_rows_slow = [&](f32x16 &acc, int tile_base) {
            bool full;
            do {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sc = acc[r];
                    const bool take = sc >= thr;
                    if (__any(take)) {                           // a row without candidates costs a compare and a branch
                        asm volatile("; row with candidates");
                        if (take && cnt < cap) {
                            stage[cnt * kWave] = ((unsigned long long)(unsigned int)(tile_base + row_of(r, h)) << 32) | __float_as_uint(sc);
                            ++cnt;
                            acc[r] = kNaN;
                        }
                    }
                }
                full = __any(cnt >= cap);                        // a full lane may have left candidates behind
                if (full) flush();
            } while (full);
        };
        // One step of the software pipeline: `cur` holds the raw scores of tile t, a[] the A operand of
        // tile t+1.  The chain of tile t+1 (into `nxt`), the loads of tile t+2 (each a[q] as soon as the
        // chain has consumed it) and the staging of tile t's candidates are ONE basic block.
        auto tile_step = [&](f32x16 &cur, f32x16 &nxt, int tile) {
            const int tile_base = tile * 32;
            mask_tile(cur, tile_base);
            const int item_h = tile_base + 4 * h;
            int slot = cnt;
            if (tile + 1 < tin1) {
                const float *tile_ptr = nullptr; int off = 0;
                if constexpr (MODE == 0) tile_addr(tile + 2 < tin1 ? tile + 2 : tin1 - 1, tile_ptr, off);   // (re-reads the last tile at the end)
#pragma unroll
                for (int r = 0; r < 16; ++r) nxt[r] = 0.f;
                // The schedule is written out by hand and pinned (sched_barrier after every MFMA): each MFMA of
                // the next tile's chain is followed by its share of this tile's 16 staging rows, and every
                // a[q] is re-loaded for the tile after next as soon as its four MFMAs have issued.
                if constexpr (MODE == 1) {
                    // 24 bf16 MFMAs; a staging row after each of the first 16 (the bf16 MFMA, unlike the fp32 one,
                    // has a shadow of ~5 vector instructions: scripts/probes/mfma_bf16_shadow_probe.hip); every
                    // item plane register is re-loaded for the tile after next right after its last use
                    const float4 *pk = packed + (int64_t)(tile + 2 < tin1 ? tile + 2 : tin1 - 1) * 12 * kWave + lane;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tm = 0; tm < 6; ++tm) {
#pragma unroll
                        for (int st = 0; st < 4; ++st) {
                            const int m = tm * 4 + st;
                            const int ai = kTermA[tm] * 4 + st;
                            nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[ai]), as_bf16x8(ub[kTermB[tm]][st]), nxt, 0, 0, 0);
                            if (m < 16) stage_row_fast(cur[m], item_h + row_of(m, 0), slot);
#ifndef IGCN_X_NOLOADA
                            if (tm == 0 || tm == 3 || tm == 5) a[ai] = pk[ai * kWave];   // last use of planes 2, 1, 0
#endif
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
                constexpr int RPG = 128 / D;                     // staging rows per group of four MFMAs (D = 64: 2)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < D / 8; ++q) {
                    const float4 aq = a[q];
                    nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.x, bfrag[4 * q + 0], nxt, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < (RPG + 3) / 4; ++i)
                        if (RPG * q + i < 16 && i < RPG) stage_row_fast(cur[RPG * q + i], item_h + row_of(RPG * q + i, 0), slot);
                    __builtin_amdgcn_sched_barrier(0);
                    nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.y, bfrag[4 * q + 1], nxt, 0, 0, 0);
#pragma unroll
                    for (int i = (RPG + 3) / 4; i < (RPG + 1) / 2; ++i)
                        if (i < RPG) stage_row_fast(cur[RPG * q + i], item_h + row_of(RPG * q + i, 0), slot);
                    __builtin_amdgcn_sched_barrier(0);
                    nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.z, bfrag[4 * q + 2], nxt, 0, 0, 0);
#pragma unroll
                    for (int i = (RPG + 1) / 2; i < (3 * RPG + 3) / 4; ++i)
                        if (i < RPG) stage_row_fast(cur[RPG * q + i], item_h + row_of(RPG * q + i, 0), slot);
                    __builtin_amdgcn_sched_barrier(0);
                    nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.w, bfrag[4 * q + 3], nxt, 0, 0, 0);
#pragma unroll
                    for (int i = (3 * RPG + 3) / 4; i < RPG; ++i)
                        stage_row_fast(cur[RPG * q + i], item_h + row_of(RPG * q + i, 0), slot);
                    a[q] = (FULL || 8 * q + 4 * h < d) ? *reinterpret_cast<const float4 *>(tile_ptr + off + 8 * q) : f4_zero();
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) stage_row_fast(cur[r], item_h + row_of(r, 0), slot);
            }
            if (__any(slot > cap)) {
                asm volatile("; staging overflow");
                flush();                                         // cnt still excludes this tile's entries
                stage_rows_slow(cur, tile_base);
            } else {
                cnt = slot;
                if (__any(cnt >= cap)) flush();
            }
#ifdef IGCN_TOPK_TRACE
            ++tr_tiles;
#endif
        };

        // prologue: scores of the first tile, A operand of the second
        f32x16 acc_a, acc_b;
        load_a(tin0);
        chain_plain(acc_a);
        if (tin0 + 1 < tin1) load_a(tin0 + 1);
        for (int tile = tin0; tile < tin1; tile += 2) {
            tile_step(acc_a, acc_b, tile);
            if (tile + 1 < tin1) tile_step(acc_b, acc_a, tile + 1);
        }
        flush();

        // ---- emit: heapsort each lane's list in place (best first), then either merge the two lanes
        // of a user into the output or hand both lists to the merge kernel ------------------------
        for (int n = k - 1; n > 0; --n) {
            const unsigned long long last = heap[n * kWave];
            heap[n * kWave] = heap[0];                           // current minimum goes to the end
            heap_replace_root(heap, n, last);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (direct) {
            if (h == 0 && user_ok) {
                const unsigned long long *pheap = heap + 32;     // partner lane (l + 32), same wave
                int i0 = 0, i1 = 0;
                unsigned long long k0 = heap[0], k1 = pheap[0];
                for (int r = 0; r < k; ++r) {
                    unsigned long long best;
                    if (k0 >= k1) { best = k0; ++i0; k0 = i0 < k ? heap[i0 * kWave] : 0ull; }
                    else          { best = k1; ++i1; k1 = i1 < k ? pheap[i1 * kWave] : 0ull; }
                    out_idx[b * k + r] = best ? key_item(best) : -1;
                    out_val[b * k + r] = best ? key_score(best) : -INFINITY;
                }
            }
        } else if (user_ok) {
            const int64_t slot = ((b - n_full * 32) * (2 * p_max) + 2 * pidx + h) * k;
            for (int r = 0; r < k; ++r) {
                const unsigned long long key = heap[r * kWave];
                ws_val[slot + r] = key ? key_score(key) : -INFINITY;
                ws_idx[slot + r] = key ? key_item(key) : kIdxNone;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#ifdef IGCN_TOPK_TRACE
    if (lane == 0) {
        atomicAdd(&g_topk_trace[0], tr_load); atomicAdd(&g_topk_trace[1], tr_chain);
        atomicAdd(&g_topk_trace[2], tr_mask); atomicAdd(&g_topk_trace[3], tr_sel);
        atomicAdd(&g_topk_trace[4], trace_clock(0) - tr_begin); atomicAdd(&g_topk_trace[5], tr_tiles);
        atomicAdd(&g_topk_trace[6], 1ull);
        unsigned long long tr_rt_end;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_rt_end) : : "memory");
        atomicAdd(&g_topk_trace[7], tr_rt_end - tr_rt_begin);      // wave lifetime in 100 MHz ticks
        if (blockIdx.x < 8192) {
            g_topk_wave_times[4 * blockIdx.x] = tr_rt_begin; g_topk_wave_times[4 * blockIdx.x + 1] = tr_rt_end;
            g_topk_wave_times[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            g_topk_wave_times[4 * blockIdx.x + 3] = 0;
        }
    }
#endif
}

// One wave per user of the groups that were cut: lane = one (piece, lane half) list, already
// best-first; k rounds of a wave-wide arg-best over the heads.
__global__ __launch_bounds__(kBlock) void topk_merge_kernel(const float *__restrict__ ws_val, const int32_t *__restrict__ ws_idx,
                                                            int64_t first_user, int64_t batch, int n_tiles, int64_t run,
                                                            int p_max, int k,
                                                            int64_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t rb = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int64_t b = first_user + rb;
    if (b >= batch) return;
    const int64_t rg = rb / 32;
    const int n_splits = 2 * (int)((((rg + 1) * n_tiles - 1) / run) - (rg * n_tiles) / run + 1);
    const float *v = ws_val + (rb * (2 * p_max) + lane) * k;
    const int32_t *ix = ws_idx + (rb * (2 * p_max) + lane) * k;
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
    const int64_t end = eval_rowptr[u + 1];
    int64_t lo = eval_rowptr[u], hi = end;
    if (!eval_col) lo = hi = end;                                // no column array: every list is empty
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (eval_col[mid] < item) lo = mid + 1; else hi = mid;
    }
    hit[i] = (lo < end && eval_col[lo] == item) ? 1.f : 0.f;
}

template <int D, bool FULL, int MODE = 0>
static int launch_topk(const TopkPlan &p, hipStream_t st,
                       const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                       const float *item_rows, int64_t ldi, int64_t n_items, int d,
                       const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned, int k,
                       int64_t *out_idx, float *out_val, float *ws_val, int32_t *ws_idx, const float4 *packed = nullptr)
{
    auto kern = score_topk_kernel<D, FULL, MODE>;
    static bool configured = false;
    if (!configured && p.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(64 * 1024));
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    if (p.units >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    const int stagger = env_int("IGCN_TOPK_STAGGER", 0, 1, 1);                          // developer knob
    hipLaunchKernelGGL(kern, dim3((unsigned)p.units), dim3(kWave), p.lds_bytes, st, user_rows, ldu, user_ids, batch,
                       item_rows, ldi, n_items, d, excl_rowptr, excl_col, banned, k, p.cap, p.n_tiles, p.n_whole,
                       p.rest_tiles, p.run, p.p_max, stagger, out_idx, out_val, ws_val, ws_idx, packed);
    return launch_status();
}

// users whose lists come back in pieces (rows of the workspace)
static inline int64_t topk_rest_users(const TopkPlan &p, int64_t batch) {
    if (p.p_max <= 1) return 0;
    const int64_t first = p.n_whole * p.units * 32;
    return batch > first ? batch - first : 0;
}

}  // namespace igcn

using namespace igcn;

extern "C" int64_t igcn_score_topk_workspace_bytes(int64_t batch, int64_t n_items, int32_t d, int32_t k)
{
    TopkPlan p;
    if (topk_make_plan(batch, n_items, d, k, &p) != IGCN_OK) return -1;
    return topk_rest_users(p, batch) * 2 * p.p_max * k * 8;
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
    if (ldu < d || ldi < d || ldu % 4 || ldi % 4 || n_items >= ((int64_t)1 << 31) - 64 || ldi > (1 << 20)) return IGCN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(user_rows) | reinterpret_cast<uintptr_t>(item_rows)) % 16) return IGCN_E_ALIGN;
    const int64_t rest_users = topk_rest_users(p, batch);
    if (rest_users > 0 && !workspace) return IGCN_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *ws_val = static_cast<float *>(workspace);
    int32_t *ws_idx = reinterpret_cast<int32_t *>(ws_val ? ws_val + rest_users * 2 * p.p_max * k : nullptr);

#define IGCN_TOPK_CASE(DD)                                                                                        \
    rc = (d == DD ? launch_topk<DD, true> : launch_topk<DD, false>)(                                             \
        p, st, user_rows, ldu, user_ids, batch, item_rows, ldi, n_items, (int)d, excl_rowptr, excl_col, banned,   \
        (int)k, out_idx, out_val, ws_val, ws_idx, nullptr)
    switch (p.d_pad) {
    case 16: IGCN_TOPK_CASE(16); break;
    case 32: IGCN_TOPK_CASE(32); break;
    case 64: IGCN_TOPK_CASE(64); break;
    default: IGCN_TOPK_CASE(128); break;
    }
#undef IGCN_TOPK_CASE
    if (rc != IGCN_OK) return rc;
    if (rest_users > 0) {
        const int64_t blocks = (rest_users + 3) / 4;
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, ws_val, ws_idx,
                           p.n_whole * p.units * 32, batch, p.n_tiles, p.run, p.p_max, (int)k, out_idx, out_val);
        rc = launch_status();
    }
    return rc;
}

// ---- the same evaluation with the products on the bf16 matrix cores (exact 3-way bf16 split of both operands,
// 6 of the 9 plane products, fp32 accumulation): fp32-grade scores (products to 2^-23), ~2.5x the throughput, but
// not the bit pattern of the fp32 fmaf chain.  d = 64 only.  Workspace = merge lists + the packed item planes.
static inline int64_t topk_split_packed_bytes(int64_t n_items) { return (n_items + 31) / 32 * 12 * kWave * 16; }

extern "C" int64_t igcn_score_topk_bf16x3_workspace_bytes(int64_t batch, int64_t n_items, int32_t d, int32_t k)
{
    if (d != 64) return -1;
    TopkPlan p;
    if (topk_make_plan(batch, n_items, d, k, &p) != IGCN_OK) return -1;
    const int64_t merge = (topk_rest_users(p, batch) * 2 * p.p_max * k * 8 + 255) / 256 * 256;
    return merge + topk_split_packed_bytes(n_items);
}

extern "C" int igcn_score_topk_bf16x3_f32(const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                                          const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                                          const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                                          int32_t k, int64_t *out_idx, float *out_val, void *workspace, void *stream)
{
    if (!user_rows || !item_rows || !out_idx || !out_val || !workspace) return IGCN_E_NULL;
    if ((excl_rowptr == nullptr) != (excl_col == nullptr)) return IGCN_E_NULL;
    if (d != 64) return IGCN_E_SHAPE;
    TopkPlan p;
    int rc = topk_make_plan(batch, n_items, d, k, &p);
    if (rc != IGCN_OK) return rc;
    if (ldu < d || ldi < d || ldu % 4 || ldi % 4 || n_items >= ((int64_t)1 << 31) - 64 || ldi > (1 << 20)) return IGCN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(user_rows) | reinterpret_cast<uintptr_t>(item_rows) | reinterpret_cast<uintptr_t>(workspace)) % 16)
        return IGCN_E_ALIGN;
    const int64_t rest_users = topk_rest_users(p, batch);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *ws_val = static_cast<float *>(workspace);
    int32_t *ws_idx = reinterpret_cast<int32_t *>(ws_val + rest_users * 2 * p.p_max * k);
    const int64_t merge_bytes = (rest_users * 2 * p.p_max * k * 8 + 255) / 256 * 256;
    float4 *packed = reinterpret_cast<float4 *>(static_cast<char *>(workspace) + merge_bytes);

    const int64_t threads = (int64_t)p.n_tiles * 4 * kWave;
    hipLaunchKernelGGL(topk_pack_items_kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       item_rows, ldi, n_items, p.n_tiles, packed);
    rc = launch_status();
    if (rc != IGCN_OK) return rc;
    rc = launch_topk<64, true, 1>(p, st, user_rows, ldu, user_ids, batch, item_rows, ldi, n_items, (int)d, excl_rowptr, excl_col,
                                  banned, (int)k, out_idx, out_val, ws_val, ws_idx, packed);
    if (rc != IGCN_OK) return rc;
    if (rest_users > 0) {
        const int64_t blocks = (rest_users + 3) / 4;
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, ws_val, ws_idx,
                           p.n_whole * p.units * 32, batch, p.n_tiles, p.run, p.p_max, (int)k, out_idx, out_val);
        rc = launch_status();
    }
    return rc;
}

#ifdef IGCN_TOPK_TRACE
extern "C" int igcn_debug_topk_wave_times(unsigned long long *host, int n_waves)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(host, HIP_SYMBOL(igcn::g_topk_wave_times), (size_t)n_waves * 32);
    return (int)e;
}
extern "C" int igcn_debug_topk_occupancy(int lds_bytes)
{
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igcn::score_topk_kernel<64, true, 0>, 64, (size_t)lds_bytes);
    return e == hipSuccess ? n : -(int)e;
}
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
    hipLaunchKernelGGL(hit_matrix_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), rec, n_users, (int)k, eval_rowptr, eval_col, hit);
    return launch_status();
}
