// Fused  score = U . I^T  ->  mask  ->  top-k  for MI355X (gfx950).
//
// Replaces, without ever materialising the [B, n_items] score matrix
// (Amazon-book: 42 GB for one full evaluation):
//   torch.mm(users_r, all_items_r.t())          model.py:120-122 / :70-71
//   scores[excl_u, excl_i] = -inf ; banned      trainer.py:149-161
//   torch.topk(scores, k)                       trainer.py:163
// and the membership loop of calculate_metrics  trainer.py:111-115 (igcn_hit_matrix).
//
// This is the one dense contraction of the path, so it runs on the matrix cores, in exact fp32:
// v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, no reduced-precision shortcut), 64 cycles each,
// 2048 cycles per 32 x 32 x (d = 64) tile.  The fp32 MFMA runs on the same lanes as the vector ALU: it
// leaves no shadow (scripts/probes/mfma_shadow_probe.hip: every vector instruction a wave puts between
// two MFMAs costs its full ~5 cycles, and a co-resident wave's vector instructions do not issue while a
// chain runs).  A SIMD's time is therefore 2048 cycles per tile PLUS everything else its waves issue,
// and the kernel is built to issue as little else as possible:
//   * one WAVE = one workgroup = 64 users (two 32-user groups; one group at d = 128) for a run of
//     32-item tiles: no barriers, nothing shared between waves.  The users' embeddings are the MFMA B
//     operands and stay in registers; an item tile (A operand) is read ONCE per 64 users, straight from
//     L2 / Infinity Cache, addressed as a scalar tile base + a constant lane offset.  The two groups'
//     chains are independent and interleaved MFMA by MFMA, so neither waits for its own accumulator;
//   * S^T = I . U^T: in the accumulator a lane holds 16 item scores of ONE user (column = lane & 31),
//     the two lanes l, l + 32 of a user hold its 32 scores of the tile;
//   * selection costs 12 vector instructions per 32 x 32 tile when nothing qualifies: the 16 scores of a
//     lane are folded to four quad maxima (v_max3 + v_max) and each is compared with the user's k-th
//     best so far (one threshold per USER, shared by its two lanes); the four wave-wide compare masks
//     go to scalar registers.  These 12 instructions are placed by hand between the MFMAs of the NEXT
//     tile's chain (one pinned basic block per tile: the loads of the tile after next into the second A
//     buffer, then MFMAs and selection).  Only quads whose mask is non-zero are looked at again, after the block, by scalar
//     branches: their four rows are appended — branch-free, 4 instructions and one LDS write a row —
//     to per-lane STAGING lists in LDS;
//   * the running top-k of a user is a k-slot 4-ary min-heap of 64-bit sortable keys
//     (order-preserving fp32 image << 32 | ~item id) in LDS, one per user, owned by ONE lane (lane
//     g * 32 + u owns user u of group g).  When some lane's staging list is nearly full every owner
//     lane drains the two lists of its user into its heap, all 64 lanes together, and the thresholds
//     are re-read;
//   * masking never touches the accumulators: each lane walks its users' sorted exclusion lists with a
//     cursor as the sweep advances, banned items arrive as one 32-bit word per tile through the scalar
//     cache (packed by a tiny pre-kernel); both become one bit per item of the current tile, looked at
//     only when a row is staged (a masked row is staged as -inf);
//   * the grid never exceeds what is resident: a wave that starts late runs its whole share after
//     everybody else has finished.  Registers allow 2 waves per SIMD; LDS is handed out in 1280-B
//     granules and the staging depth is what they leave beside the heaps (k <= 24: 8 waves per CU;
//     larger k: 4, 2 or 1).  Every wave gets the same number of tiles: with G 64-user groups and W
//     resident waves each wave sweeps floor(G / W) whole groups, and the tiles of the remaining groups,
//     laid end to end, are cut into W equal runs.  A group that is cut returns one best-first list per
//     piece, merged by a small second kernel.
// Ties are broken towards the lower item id (torch.topk leaves them unspecified).
#include <math.h>
#include <type_traits>
#include "common.h"
#include "topk_order.h"

namespace igcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// A pointer into the CONSTANT address space: a wave-uniform load through it is a scalar load (s_load).  Through a plain
// global pointer the compiler emits a vector load + s_waitcnt vmcnt(0) + readfirstlane, and that wait also drains every
// item tile in flight.  Only for memory no kernel of the same launch writes (the banned items packed by a pre-kernel).
typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
typedef const __attribute__((address_space(4))) float *const_f32_ptr;

constexpr int kIdxNone = 0x7fffffff;

// ---- candidate sweep on the bf16 matrix cores (MODE 1 of the kernel) --------------------------------------------
// Every fp32 x is split x = h + l + r, h = bf16(x), l = bf16(x - h), |l| <= 2^-8 |x|, |r| <= 2^-16 |x|.  The sweep forms
// h_i h_u + h_i l_u + l_i h_u (three bf16 MFMAs per 16 k, fp32 accumulate: 12 instead of 32 fp32 MFMAs per
// 32 x 32 x 64 tile, each ~38 instead of 64 cycles) — a score off by at most 2^-14 |u| |i| — and keeps the k + 4
// best candidates per user; topk_rescore_kernel then recomputes their scores exactly in fp32, orders them, and
// checks that no item the sweep dropped can reach the k-th exact score (else the user is handed to the fp32 sweep).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kFastExtra = 4;                                    // candidates kept beyond k (default; see topk_fast_extra)
__device__ __forceinline__ unsigned int bf16_rne_bits(float v) {
    const unsigned int u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// eight consecutive fp32 -> two bf16x8 planes (as float4 bit patterns: two bf16 per dword, low half first)
__device__ __forceinline__ void split2_x8(const float4 &lo, const float4 &hi, float4 out[2]) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned int w[2][4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const unsigned int a0 = bf16_rne_bits(v[e]), b0 = bf16_rne_bits(v[e + 1]);
        const unsigned int a1 = bf16_rne_bits(v[e] - __uint_as_float(a0 << 16));
        const unsigned int b1 = bf16_rne_bits(v[e + 1] - __uint_as_float(b0 << 16));
        w[0][e / 2] = a0 | (b0 << 16);
        w[1][e / 2] = a1 | (b1 << 16);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
        out[p] = make_float4(__uint_as_float(w[p][0]), __uint_as_float(w[p][1]), __uint_as_float(w[p][2]), __uint_as_float(w[p][3]));
}
__device__ __forceinline__ bf16x8 as_bf16x8(const float4 &v) { return __builtin_bit_cast(bf16x8, v); }

// ---- MODE 2: the candidate sweep in fp16.  Items as ONE fp16 plane (11 significant bits: half the bytes of the two
// bf16 planes — the sweep is bound by the L1 throughput of its item-tile loads), users as two (h = fp16(x),
// l = fp16(x - h)): h_i (h_u + l_u), 8 MFMAs per 32 x 32 x 64 tile, a score off by at most 2^-11 |u| |i|.  Both tables
// are scaled by a power of two that brings their largest element into [0.5, 1) (fp16 has 5 exponent bits); elements
// below 2^-14 of the largest lose relative, not absolute, accuracy (<= 2^-25 of the largest each).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ half8 as_half8(const float4 &v) { return __builtin_bit_cast(half8, v); }
__device__ __forceinline__ unsigned int f16_bits(_Float16 h) { return (unsigned int)__builtin_bit_cast(unsigned short, h); }
// power of two s with s * m in [0.5, 1) (m > 0), as its exponent: s = 2^-e
__device__ __forceinline__ int scale_exp(float m) { int e = 0; if (m > 0.f) (void)frexpf(m, &e); return e; }
// eight consecutive fp32, scaled by s -> one or two fp16x8 planes (float4 bit patterns, low half first)
__device__ __forceinline__ void split_f16_x8(const float4 &lo, const float4 &hi, float s, float4 out[2]) {
    const float v[8] = {lo.x * s, lo.y * s, lo.z * s, lo.w * s, hi.x * s, hi.y * s, hi.z * s, hi.w * s};
    unsigned int w[2][4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const _Float16 a0 = (_Float16)v[e], b0 = (_Float16)v[e + 1];
        const _Float16 a1 = (_Float16)(v[e] - (float)a0), b1 = (_Float16)(v[e + 1] - (float)b0);
        w[0][e / 2] = f16_bits(a0) | (f16_bits(b0) << 16);
        w[1][e / 2] = f16_bits(a1) | (f16_bits(b1) << 16);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
        out[p] = make_float4(__uint_as_float(w[p][0]), __uint_as_float(w[p][1]), __uint_as_float(w[p][2]), __uint_as_float(w[p][3]));
}

struct TopkPlan {
    int d_pad;               // 16 / 32 / 64 / 128 / 256
    int ng;                  // 32-user groups per wave
    int64_t groups;          // wave-groups (32 * ng users each)
    int n_tiles;             // 32-item tiles of one sweep
    int64_t units;           // waves launched; all resident at once
    int64_t n_whole;         // whole sweeps per unit: groups [0, n_whole*units), unit u takes u, u+units, ...
    int64_t rest_tiles;      // tiles of the remaining groups (laid end to end), cut into runs of `run`
    int64_t run;             // tiles of the rest that one unit takes
    int p_max;               // bound on the pieces a rest group is cut into (1: runs are whole sweeps)
    int cap;                 // staging slots per lane and group
    size_t lds_bytes;
};

constexpr int kMinCap = 8, kMaxCap = 16, kQuad = 4;
// Candidate sweep, early exit: a wave alive after this many tiles hands its remaining users to the fp32 sweep once three
// quarters of the waves have left AND one of its users is far from done (reach >= 1.5 x threshold: a wave about to leave by
// itself stays).  Trained LightGCN tables (Amazon-like, epochs 2-3): the stragglers hold ~20-40 users with low thresholds
// and crawl at ~6 us a tile; scoring 1.3-1.45 ms with 48 (the same with 24), 1.5-1.7 with 96, 1.8-2.0 with 192, 3.4-3.8
// when nobody gives up; after the first epoch (every wave leaves within ~50 tiles) nobody is handed over (0.41 ms).
constexpr int kGiveUpAfterTiles = 48;
constexpr int kEarlyCheckEvery = 6;      // exit checks every so many tiles up to tile 48 (a multiple of 3: the sweep's turn), give-up from twice that; 0: off
// A user that keeps filling its staging list while the rest of its wave has gone quiet makes the wave crawl (a drain is paid by all 64
// lanes: 2.5 drains and 15 us a tile against ~1 us).  Such users are handed to the fp32 sweep by THEIR OWN wave as soon as they show:
// drains a lane pair triggered between tiles 3 and 6 of the sweep (after the lists' warm-up), checked at tile 6 — at most kMaxEvict
// users of a wave, and only when at least half of the wave's users are already out of reach (at random init everybody is still
// reachable and nobody is handed over).  A local, deterministic rule: no counter of other waves is read.
constexpr int kHotDrains = 3, kMaxEvict = 4;
constexpr int kExitSlots = 64;       // counters of early leavers, one per whole sweep of a wave (256 B of the call's workspace)
__device__ __forceinline__ int exit_slot(int64_t job) { return job < kExitSlots ? (int)job : kExitSlots - 1; }
constexpr int kFastFallbackMaxInt = IGCN_FAST_FALLBACK_MAX;
constexpr int kWarmTiles = 128;     // tiles of the candidate sweep's warm-up pass ("topk_fast_warm"; see the kernel)
constexpr float kWarmFlat = 0.5f;   // ... taken only if the rows at its end are still this long relative to the first
constexpr int kMinCapSweep = 6;      // candidate sweeps: a shallower staging list (more drains) rather than half the resident waves (k + extra = 25..28)

// (the fp16 candidate sweep at d = 128 runs ONE wave per SIMD with 512 registers: both user groups stay)
// candidates the sweep keeps beyond k: more of them widen the gap the completeness check needs (fewer users handed to
// the fp32 sweep) and lengthen every list (more heap updates)
static inline int topk_fast_extra(int k, int mode) {
    int e = tuning_get(IGCN_TUNE_TOPK_FAST_EXTRA);
    // one fp16 plane each side doubles the error bound of the sweep: two more candidates keep the users handed to the
    // fp32 sweep as few (Amazon-like, k = 20: 1 484 of 109 730 with 4 extra, 43 with 6; the sweep's time is the same)
    if (e < 1) e = mode == 3 ? kFastExtra + 2 : kFastExtra;
    return k + e > kWave ? kWave - k : e;
}
// candidate sweep of the two-stage path: 3 = one fp16 plane each side (default), 2 = one fp16 item plane, two user planes,
// 1 = two bf16 planes each side (d = 64 only)
static inline int topk_fast_mode(int d) {
    const int t = tuning_get(IGCN_TUNE_TOPK_FAST_MODE);
    return t == 1 && d == 64 ? 1 : t == 2 ? 2 : 3;
}
static inline bool topk_wide_sweep() { return tuning_get(IGCN_TUNE_TOPK_FAST_WIDE) != 0; }
// sweep_mode: 0 the fp32 sweep, 1 / 2 / 3 the candidate sweeps (MODE of the kernel)
// (d_pad = 256, round 5: a whole item row and user row in registers need ~340; at two waves per SIMD the 90 beyond 256 went to
// SCRATCH, at one they sit in AGPRs — no kernel of the library carries a private segment, see capture_guard)
static inline bool topk_one_wave_per_simd(int d_pad, int sweep_mode) { return d_pad == 256 || (sweep_mode == 2 && d_pad == 128 && topk_wide_sweep()); }
// sweep_mode -1: the bounded fp32 sweep (igcn_score_topk_bounded_f32).  Its batches are the few users the two-stage path hands
// back: at d = 64 one 32-user group per wave then — a 64-user wave-group cut into the <= 58 pieces the merge takes cannot fill
// the chip below ~2 000 users, half-size groups give twice the waves for the same work (20-40 users: 243 -> ~130 us).
constexpr int64_t kNarrowBoundedBatch = 2048;
static inline bool topk_narrow_small_batches() { return tuning_get(IGCN_TUNE_TOPK_FAST_NARROW) != 0; }
static inline int topk_groups_per_wave(int d_pad, int sweep_mode = 0, int64_t batch = 0) {
    if (sweep_mode == -1 && d_pad == 64 && batch <= kNarrowBoundedBatch) return 1;
    // The candidate sweep of a batch that cannot give every second wave slot a 64-user group is cut into pieces, and a piece's
    // time is mostly the warm-up of its users' lists (k ln(n / k) candidates each, most of them in the first thousands of
    // items, whatever the piece's length): 32-user groups halve that per wave and double the waves.
    if (sweep_mode == 3 && d_pad == 64 && topk_narrow_small_batches() && (batch + 63) / 64 * 2 < 8 * (int64_t)cu_count()) return 1;
    return d_pad <= 64 || (d_pad == 128 && (sweep_mode == 3 || topk_one_wave_per_simd(d_pad, sweep_mode))) ? 2 : 1;
}

static inline int topk_make_plan(int64_t batch, int64_t n_items, int32_t d, int32_t k, TopkPlan *p, int sweep_mode = 0) {
    const bool candidate_sweep = sweep_mode > 0;
    if (batch < 1 || n_items < 1) return IGCN_E_SHAPE;
    if (d < 4 || d > 256 || d % 4 != 0) return IGCN_E_SHAPE;
    if (k < 1 || k > IGCN_MAX_TOPK || k > n_items) return IGCN_E_RANGE;
    p->d_pad = d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : d <= 128 ? 128 : 256;
    p->ng = topk_groups_per_wave(p->d_pad, sweep_mode, batch);
    const int upw = 32 * p->ng;
    p->groups = (batch + upw - 1) / upw;
    const int64_t L = (n_items + 31) / 32;
    p->n_tiles = (int)L;
    // Resident waves per CU.  Registers allow 2 per SIMD.  LDS is handed out in granules of 1280 B (1/128 of the
    // CU's 160 KiB; measured: 12 x 13312 B do not fit, 12 x 12800 B do): a wave gets 128 / per_cu of them, holds
    // the heaps (k slots x 64 owner lanes x 8 B) and gives the rest to the staging lists.  The grid must never
    // exceed what is resident: a wave that starts late runs its whole share after everybody else has finished.
    int per_cu = topk_one_wave_per_simd(p->d_pad, sweep_mode) ? 4 : 8;
    const int want = tuning_get(IGCN_TUNE_TOPK_WAVES_PER_CU);
    if ((want == 8 || want == 4 || want == 2 || want == 1) && want <= per_cu) per_cu = want;
    int cap = 0;
    for (; per_cu >= 1; per_cu >>= 1) {
        const int64_t budget = (int64_t)(128 / per_cu) * 1280;
        cap = (int)((budget - (int64_t)k * kWave * 8) / (p->ng * kWave * 8));
        if (cap >= (candidate_sweep ? kMinCapSweep : kMinCap)) break;
    }
    if (per_cu < 1) return IGCN_E_RANGE;
    if (cap > kMaxCap) cap = kMaxCap;
    const int cap_t = tuning_get(IGCN_TUNE_TOPK_CAP);
    if (cap_t >= kMinCap && cap_t <= cap) cap = cap_t;
    p->cap = cap;
    p->lds_bytes = (size_t)(k + p->ng * cap) * kWave * 8;
    int64_t slots = (int64_t)per_cu * cu_count();
    const int slots_t = tuning_get(IGCN_TUNE_TOPK_SLOTS);       // tests: whole sweeps + cut rest at small sizes
    if (slots_t >= 1 && slots_t <= slots) slots = slots_t;
    int64_t rest;
    if (p->groups >= slots) {
        p->units = slots;
        p->n_whole = p->groups / slots;
        rest = p->groups - p->n_whole * slots;
    } else if (candidate_sweep && 2 * p->groups >= slots) {
        // The candidate sweeps of the two-stage path are not bound by the matrix cores but by the handling of their
        // candidates: filling every wave slot by cutting the sweeps into pieces buys them nothing and costs a list
        // warm-up per piece plus the merge — one whole sweep per wave-group instead (Amazon-like evaluation, 1715
        // groups on 2048 slots: 5.2 -> 4.3 ms; the fp32 sweep, which does need every slot: 12.1 -> 13.5 ms).
        p->units = p->groups;
        p->n_whole = 1;
        rest = 0;
    } else {
        p->units = 0;
        p->n_whole = 0;
        rest = p->groups;
    }
    p->rest_tiles = rest * L;
    p->run = 0;
    p->p_max = 1;
    if (rest > 0) {
        // a piece shorter than 32 tiles is mostly list warm-up; a group in more than 60 pieces does
        // not fit the 64 lanes of the merge
        int64_t min_run = L < 32 ? L : 32;
        if ((L + 57) / 58 > min_run) min_run = (L + 57) / 58;
        // The narrow bounded sweep (the two-stage path's fall-back: a handful of users, one 32-user group per wave, ONE wave per SIMD
        // at best — each piece is a latency chain of dependent fp32 MFMAs and exposed tile loads, ~3 us per tile where the matrix pipe
        // alone would need 0.9) takes pieces a quarter as long: four times the waves in flight for the same work, and the lists start
        // from the caller's bounds, so there is little warm-up to repeat.  Its merge handles four lists per lane (topk_merge_wide_kernel).
        if (sweep_mode == -1 && p->ng == 1 && tuning_get(IGCN_TUNE_TOPK_FAST_PIECES) != 0) {
            min_run = L < 12 ? L : 12;
            if ((L + 229) / 230 > min_run) min_run = (L + 229) / 230;
        }
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
    const unsigned int u = (unsigned int)(key >> 32);
    // inverse of the map in make_key, written as one xor: clearing the top bit with an AND reads as fabs() to the
    // instruction selector, which then crashes folding it into the threshold compare (ROCm 7.2 clang, gfx950)
    const unsigned int m = (unsigned int)((int)u >> 31);          // all ones: the score was >= +0
    return __uint_as_float(u ^ (~m | 0x80000000u));
}
__device__ __forceinline__ int key_item(unsigned long long key) { return (int)~(unsigned int)key; }

// (v, i) ranks before (w, j): higher score first, then lower item id
__device__ __forceinline__ bool ranks_before(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

// Issue priority.  Base priority 0 or 1 from the hardware wave slot: the two waves of a SIMD get different ones, so
// one of them always wins the matrix pipe and the other fills in behind it.  A wave in its slow path (staging
// candidates, draining them into the heaps) raises itself to 3: its few vector instructions then go ahead of the
// partner's next MFMA instead of queueing behind every one of them, the slow path is over sooner, and both waves
// are back to feeding the matrix pipe.
__device__ __forceinline__ unsigned wave_slot() {
    // s_getreg_b32 HW_REG_HW_ID (id 4), WAVE_ID = bits [3:0]: simm16 = (size-1) << 11 | offset << 6 | id
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;
}
__device__ __forceinline__ void set_base_priority(unsigned slot) {
    if (slot) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
}

// 4-ary min-heap of sortable keys in LDS, [slot][owner lane] (children of slot i: 4i+1 .. 4i+4): replace the root
// by `cand` and sift down.  The four children of a level are read together, so a level costs one LDS round trip
// and a 20-slot heap has two levels below the root (a binary heap: four, with two dependent reads each).
// Returns the new root.
__device__ __forceinline__ unsigned long long heap_replace_root(unsigned long long *heap, int n, unsigned long long cand) {
    int i = 0;
    unsigned long long first_up = 0ull;
    while (true) {
        const int c = 4 * i + 1;
        if (c >= n) break;
        const unsigned long long none = ~0ull;
        unsigned long long k0 = heap[c * kWave];
        unsigned long long k1 = c + 1 < n ? heap[(c + 1) * kWave] : none;
        unsigned long long k2 = c + 2 < n ? heap[(c + 2) * kWave] : none;
        unsigned long long k3 = c + 3 < n ? heap[(c + 3) * kWave] : none;
        int m01 = c, m23 = c + 2;
        if (k1 < k0) { k0 = k1; m01 = c + 1; }
        if (k3 < k2) { k2 = k3; m23 = c + 3; }
        if (k2 < k0) { k0 = k2; m01 = m23; }
        if (k0 >= cand) break;
        heap[i * kWave] = k0;
        if (i == 0) first_up = k0;
        i = m01;
    }
    heap[i * kWave] = cand;
    return i == 0 ? cand : first_up;
}

// hand-written vector instructions of the pinned block (the compiler would canonicalise the operands of an fmaxf
// with v_max_f32 x, x first).
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float m;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}
__device__ __forceinline__ void vmax_into(float &m, float x) {      // m = max(m, x) in m's own register
    asm volatile("v_max_f32 %0, %0, %1" : "+v"(m) : "v"(x));
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float m;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return m;
}

#ifdef IGCN_TOPK_STATS
// Developer build only (scripts/dev_topk_variants.py): event counts summed over waves.
// [0] tiles  [1] tiles with a hit quad  [2] hit quads  [3] flushes  [4] flush iterations  [5] staged candidates  [6] waves
// [7] shader cycles inside flush()  [8] inside stage_hits()  [9] whole wave  [10] inside build_masks()
__device__ unsigned long long g_topk_stats[16];   // ... [12] cycles before the first tile of a job (operands, cursors, heaps)  [13] cycles after the last (heapsort, emit)  [14] / [15] cycles from the first tile to tile 63 / 255
__device__ unsigned long long g_topk_wave_times[3 * 4096];   // begin, end (100 MHz ticks), HW_ID per workgroup
#define IGCN_CLOCK() __builtin_amdgcn_s_memtime()
#define IGCN_STAT(i, v) (st_##i += (v))
#else
#define IGCN_STAT(i, v) ((void)0)
#endif

struct TopkArgs {
    const float *user_rows; int64_t ldu; const int64_t *user_ids; int64_t batch;
    const float *item_rows; int64_t ldi; int64_t n_items; int d;
    const int64_t *excl_rowptr; const int32_t *excl_col; const uint32_t *banned_bits;
    int k, cap, n_tiles, p_max, stagger;
    int64_t n_whole, rest_tiles, run;
    int64_t *out_idx; float *out_val; float *ws_val; int32_t *ws_idx;
    const float4 *packed;     // MODE 1: item planes [tile][plane 0..1][k-step 0..3][lane] x 16 B; MODE 2: [tile][k-step][lane] x 16 B
    const unsigned int *stats; // MODE 2: bit patterns of max |item row|^2, max |item element|, max |user element|
    const float *init_thr;     // NULL, or per batch position a LOWER BOUND of the user's k-th best score: only items that reach it are looked at
    // MODE 2, descending-norm order: tile_bound[t] >= s_i |i| (1 + margin) for every item of tile t AND of every later
    // tile; unorm2[b] = |u_b|^2.  |approximate score| <= s_u |u| tile_bound: once no user of the wave can be reached, the
    // rest of the sweep is skipped.  NULL: no early exit.
    const float *tile_bound; const float *unorm2;
    // early exit bookkeeping (MODE 2 / 3, NULL: none): waves that left early; users a wave gave up on (see the sweep loop)
    unsigned int *exit_count; uint8_t *unfinished;
    unsigned int *shared_thr;   // NULL, or per batch position the best k-th-best any piece of the user's sweep has published (see flush())
    // BOUNDED only (the two-stage path's fall-back, planned on the device): the batch is a LIST in device memory — count_dev[0]
    // users (at most `batch`: the plan's size), user b of it sits at batch position rows[b] of the caller's arrays (user_ids,
    // out_idx / out_val); init_thr stays indexed by b.  NULL: the batch is positions 0 .. batch - 1.
    const int32_t *rows; const int32_t *count_dev;
    // BOUNDED, NG = 1, pieces (the two-stage path's fall-back): per user k slots of sortable score bits, slot s = the best score any
    // of the pieces s P / k .. (s + 1) P / k - 1 has met — k slots certify k DIFFERENT items, so their minimum is a lower bound of the
    // user's k-th best (see flush()).  NULL: off.
    unsigned int *piece_best;
    // Whole candidate sweeps (MODE 3, d = 64): a warm-up pass over the first warm_tiles tiles keeps, per user, the best score of each
    // of the 32 accumulator slots (32 different items); the kc-th largest of them is a lower bound of the user's kc-th best, kept in
    // warm_thr[batch position] — the sweep proper then starts with that threshold instead of -inf (see the kernel).  0 / NULL: off.
    int warm_tiles; float *warm_thr;
    int early_checks;          // candidate sweep: exit checks every so many tiles up to tile 48, give-up from twice that (0: every 24 tiles / from tile 48)
};

// FULL: d == D, no k-slice of a row is padding.  NG: 32-user groups of a wave.  MODE 0: fp32 MFMA, the exact fmaf
// chain.  MODE 1 (D = 64, FULL): the candidate sweep on the bf16 matrix cores described above.
// BOUNDED: A.init_thr holds a lower bound of every user's k-th best score (igcn_score_topk_bounded_f32); a variant of
// its own so that the plain sweeps carry nothing of it (two more live values cost the fp32 sweep 4 %).
// MODE 2 at D = 128: one wave per SIMD (512 registers: 128 of user planes, 64 of accumulators, 96 of item-tile ring).
template <int D, int NG, bool FULL, int MODE = 0, bool BOUNDED = false>
__global__ __launch_bounds__(kWave, ((MODE == 2 && D == 128 && NG == 2) || D == 256) ? 1 : 2) void score_topk_kernel(const TopkArgs A)
{
    static_assert(MODE == 0 || (FULL && ((D == 64 && (NG == 2 || MODE == 3)) || ((MODE == 2 || MODE == 3) && D == 128))), "the candidate sweeps are built for d = 64 (and fp16: d = 128)");
    constexpr int KS = D / 16;                                   // MODE 1 / 2 / 3: k-steps of 16 per row
    constexpr bool kF16 = MODE == 2 || MODE == 3;                // fp16 candidate sweep; MODE 3: ONE user plane (h only)
    constexpr int kUserPlanes = MODE == 3 ? 1 : 2;
    // MODE 3 at d = 128: 64 registers of user plane + 64 of accumulators leave room for 12 item-tile quads, not 16 or 24: a
    // ring of three HALF tiles (4 k-steps each).  A tile is (lo, hi); while a block multiplies (lo, hi) the third half takes
    // the next tile's lo at the block's start and lo itself is re-loaded with the next tile's hi as soon as its four
    // k-steps are consumed: loads run one block ahead, as in the fp32 sweep's two-buffer scheme.
    constexpr bool kRing12 = MODE == 3 && D == 128;
    static_assert(!BOUNDED || MODE == 0, "a lower bound goes with the exact sweep");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *const heap_base = reinterpret_cast<unsigned long long *>(smem);       // [k][64 owner lanes]
    unsigned long long *const heap = heap_base + threadIdx.x;                                  // this lane's own heap
    unsigned long long *const stage_all = heap_base + A.k * kWave;                             // [NG][cap][64 lanes]

    const unsigned prio_slot = (A.stagger & 1) ? wave_slot() : 0u;
    const bool prio_boost = (A.stagger & 2) != 0;
    set_base_priority(prio_slot);
#ifdef IGCN_TOPK_STATS
    unsigned long long st_0 = 0, st_1 = 0, st_2 = 0, st_3 = 0, st_4 = 0, st_5 = 0, st_7 = 0, st_8 = 0, st_10 = 0, st_12 = 0, st_13 = 0, st_14 = 0, st_15 = 0, st_job = 0, st_first = 0;
    const unsigned long long st_begin = IGCN_CLOCK();
    const unsigned long long st_rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x;
    const int j = lane & 31, h = lane >> 5;
    const int k = A.k, cap = A.cap, n_tiles = A.n_tiles;
    const int64_t ldi = A.ldi, n_items = A.n_items;
    constexpr int UPW = 32 * NG;                                 // users of a wave
    const int own_g = NG == 2 ? h : 0;                           // the group whose user (own_g, j) this lane's heap belongs to
    const bool owner = NG == 2 || h == 0;
    const int64_t units = gridDim.x;
    const int64_t n_full = A.n_whole * units;
    int64_t batch_n = A.batch;                                   // users of the batch; BOUNDED with a device list: as many as it holds
    if constexpr (BOUNDED) {
        // (a SCALAR load: the count was written by an earlier kernel of the stream; a vector load would put the batch size, and every
        // comparison with it, into vector registers)
        if (A.count_dev) { const int64_t c = (int32_t)((const_u32_ptr)(uintptr_t)A.count_dev)[0]; batch_n = c < A.batch ? c : A.batch; }
    }
    int64_t rx = (int64_t)blockIdx.x * A.run;                    // cursor in the rest groups' tile space
    const int64_t rx_end = rx + A.run < A.rest_tiles ? rx + A.run : A.rest_tiles;

    for (int64_t job = 0;; ++job) {
        // ---- next piece: users of wave-group `group`, item tiles [tin0, tin1) -------------------
#ifdef IGCN_TOPK_STATS
        st_job = IGCN_CLOCK();
#endif
        int64_t group;
        int tin0, tin1, pidx = 0;
        bool direct = true;
        if (job < A.n_whole) {
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
            pidx = (int)((int64_t)blockIdx.x - (rg * n_tiles) / A.run);
            direct = A.p_max == 1;
            rx += tin1 - tin0;
        }
        if constexpr (BOUNDED) {
            if (group * UPW >= batch_n) break;                   // planned for more users than the list holds: groups ascend, nothing follows
        }
        const int item_lo = tin0 * 32;
        const int item_hi = (int64_t)tin1 * 32 < n_items ? tin1 * 32 : (int)n_items;

        // B operands: user (g, j) of this lane.  Lane half h supplies k = 8q + 4h + c (q < D/8, c < 4): the two
        // lanes of a row read adjacent 16-B pieces, so one load instruction touches 32 lines, not 64.
        float bfrag[NG][MODE == 0 ? D / 2 : 1];
        float4 ub[MODE != 0 ? NG : 1][2][MODE != 0 ? KS : 1];   // MODE 1 / 2: [group][plane][k-step], 8 halves each: k = 16 s + 8 h .. + 7
        int64_t uid[NG];
        bool user_ok[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int64_t b = group * UPW + g * 32 + j;
            user_ok[g] = b < batch_n;
            int64_t brow = b;                                     // position in the caller's arrays
            if constexpr (BOUNDED) { if (A.rows && user_ok[g]) brow = A.rows[b]; }
            uid[g] = user_ok[g] ? (A.user_ids ? A.user_ids[brow] : brow) : 0;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int q = 0; q < D / 8; ++q) {
                    float4 v = f4_zero();
                    const int e = 8 * q + 4 * cold(h);
                    if (user_ok[g] && (FULL || e < A.d)) v = *reinterpret_cast<const float4 *>(A.user_rows + uid[g] * A.ldu + e);
                    bfrag[g][4 * q + 0] = v.x; bfrag[g][4 * q + 1] = v.y; bfrag[g][4 * q + 2] = v.z; bfrag[g][4 * q + 3] = v.w;
                }
            } else {
                const float su = kF16 ? ldexpf(1.f, -scale_exp(__uint_as_float(A.stats[2]))) : 1.f;
#pragma unroll
                for (int st = 0; st < KS; ++st) {
                    float4 lo = f4_zero(), hi = f4_zero();
                    if (user_ok[g]) {
                        const float *src = A.user_rows + uid[g] * A.ldu + 16 * st + 8 * cold(h);
                        lo = *reinterpret_cast<const float4 *>(src);
                        hi = *reinterpret_cast<const float4 *>(src + 4);
                    }
                    float4 planes[2];
                    if constexpr (kF16) split_f16_x8(lo, hi, su, planes); else split2_x8(lo, hi, planes);
                    ub[g][0][st] = planes[0]; ub[g][1][st] = planes[1];
                }
            }
        }

        // exclusion cursors: ex_next = first excluded item >= item_lo of each of this lane's users, ex_after = the entry behind it
        // (already on its way), ex_cur -> that entry's place in the list, ex_left = entries from ex_cur to the list's end (ex_next is
        // real iff ex_left >= 0, ex_after iff ex_left >= 1).  Five registers per user; a (base, position, end) triple took six.
        const int32_t *ex_cur[NG];
        int ex_left[NG], ex_next[NG], ex_after[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            ex_cur[g] = A.excl_col;
            ex_left[g] = -1; ex_next[g] = kIdxNone; ex_after[g] = kIdxNone;
            if (A.excl_rowptr && user_ok[g]) {
                const int64_t r0 = A.excl_rowptr[uid[g]];
                const int32_t *ptr = A.excl_col + r0;
                const int end = (int)(A.excl_rowptr[uid[g] + 1] - r0);
                int lo = 0, hi = end;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (ptr[mid] < item_lo) lo = mid + 1; else hi = mid;
                }
                ex_cur[g] = ptr + lo + 1;
                ex_left[g] = end - (lo + 1);
                if (lo < end) ex_next[g] = ptr[lo];
                if (lo + 1 < end) ex_after[g] = ptr[lo + 1];
            }
        }

        // running top-k: one min-heap of sortable keys per user in LDS, owned by one lane; its root (= the user's
        // k-th best so far) in the owner's registers.  Key 0 = empty slot: ranks below every real entry, masked
        // (-inf) ones included.
        {
            // (a zero made HERE: the compiler otherwise keeps one 64-bit zero pair for every job of the kernel — hoisted, carried
            // across the sweep and, in the variants with the fewest registers to spare, spilled)
            const unsigned z = (unsigned)cold(0);
            const unsigned long long zero = ((unsigned long long)z << 32) | z;
            for (int s = 0; s < k; ++s) heap[s * kWave] = zero;
        }
        // (the root is NOT carried in registers across the sweep: flush() reads heap[0] back — two registers the sweep loop needs)
        unsigned long long best = 0ull;                          // (BOUNDED, NG = 1: the best key this piece has met, see A.piece_best)
        float thr[NG];                                           // k-th best score of user (g, j): the same in both lanes of a user
        // Staging lists: wpos[g] = LDS byte address of this lane's next free slot of group g (slots of a lane are 512 bytes apart).  The
        // list's first slot is re-derived from the lane id where it is needed (stage_first: one instruction) instead of being carried
        // beside a counter: two registers less across the sweep, and the store address of a staged row is wpos itself.
        unsigned wpos[NG];
        auto stage_first = [&](int g) -> unsigned { return (unsigned)(uintptr_t)(stage_all + (g * cap) * kWave) + ((unsigned)cold(lane) << 3); };
        auto staged = [&](int g) -> int { return (int)((wpos[g] - stage_first(g)) >> 9); };      // staged candidates of this lane
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // a lane without a user (past the end of the batch) never has a candidate: its zero embedding would
            // otherwise tie every score with its threshold and flood the staging lists
            thr[g] = user_ok[g] ? (BOUNDED ? A.init_thr[group * UPW + g * 32 + cold(j)] : -INFINITY) : INFINITY;
            wpos[g] = stage_first(g);
        }

        constexpr bool kWarm = MODE == 3 && D == 64;             // warm-up pass before a whole sweep (see below)
        bool warm = false;
        constexpr bool kHotTrack = MODE == 3 && D == 64;         // (the default candidate sweep; the others have no registers to spare)
        int hot[kHotTrack ? NG : 1];                             // drains this lane's staging list has triggered (see kHotDrains)
#pragma unroll
        for (int g = 0; g < (kHotTrack ? NG : 1); ++g) hot[g] = 0;
        unsigned exm[NG];                                        // mask bits of the current tile (see build_masks)
#pragma unroll
        for (int g = 0; g < NG; ++g) exm[g] = 0u;
        bool mflag = false;

        // staged candidates -> heaps: every owner lane drains the two lists of its user (lane halves 0 and 1)
        auto flush = [&]() {
#ifdef IGCN_TOPK_STATS
            const unsigned long long t_in = IGCN_CLOCK();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");  // the hand-written staging stores
            int n0, n1;
            if constexpr (NG == 2) {
                const int c0 = staged(0), c1 = staged(1);
                const int a0 = __shfl(c0, j), a1 = __shfl(c0, j + 32);
                const int b0 = __shfl(c1, j), b1 = __shfl(c1, j + 32);
                n0 = h ? b0 : a0;
                n1 = h ? b1 : a1;
            } else {
                const int c0 = staged(0);
                n0 = __shfl(c0, j);
                n1 = __shfl(c0, j + 32);
                if (!owner) n0 = n1 = 0;
            }
            const unsigned long long *l0 = stage_all + (own_g * cap) * kWave + j;
            const int n = n0 + n1;
            unsigned long long root = heap[0];
            IGCN_STAT(3, 1);
            for (int i = 0; __any(i < n); ++i) {
                IGCN_STAT(4, 1);
                if (i < n) {
                    IGCN_STAT(5, 1);
                    const unsigned long long raw = i < n0 ? l0[i * kWave] : l0[(i - n0) * kWave + 32];
                    const unsigned long long cand = make_key(__uint_as_float((unsigned int)raw), (int)(raw >> 32));
                    if (cand > root) root = heap_replace_root(heap, k, cand);
                    if constexpr (BOUNDED && NG == 1) { if (cand > best && (unsigned int)raw != 0xff800000u) best = cand; }   // (-inf: a masked fill-in)
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                wpos[g] = stage_first(g);
                const unsigned long long r = heap_base[g * 32 + j];     // root of user (g, j), kept by lane g * 32 + j
                thr[g] = !user_ok[g] ? INFINITY : r ? key_score(r) : -INFINITY;   // list not full yet: everything may enter
                if constexpr (kWarm) {                                  // ... that reaches the warm-up pass's bound (>= kc items do)
                    if (warm && !r && user_ok[g])
                        thr[g] = __hip_atomic_load(A.warm_thr + group * UPW + g * 32 + cold(j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // ... that reaches the caller's lower bound (the list fills from the items above it: there are >= k)
                if constexpr (BOUNDED) { if (user_ok[g]) thr[g] = fmaxf(thr[g], A.init_thr[group * UPW + g * 32 + cold(j)]); }
            }
            // (compiled into the default candidate sweep and into the narrow bounded sweep — the two-stage path's fall-back, whose
            // few users are always cut into ~58 pieces: without sharing each piece warms a list of its own, k ln(1 700 / k) ~ 110
            // candidates per user and piece instead of a handful; the other variants have no registers to spare)
            if constexpr ((MODE == 3 && D == 64) || (BOUNDED && NG == 1)) if (A.shared_thr) {
                // The pieces a user's sweep is cut into run at the same time, each with a list of its own: they share their
                // thresholds.  A piece's k-th best is a lower bound of the user's k-th best over the whole table, so every
                // piece may use the largest one any of them has reached — the pieces over the short rows then stop staging
                // almost at once instead of warming up a list nobody will look at.  (Sortable score bits, atomicMax; 0 = none.)
                const int64_t b_own = group * UPW + cold(lane);
                if (owner && root && b_own < batch_n) atomicMax(A.shared_thr + b_own, (unsigned int)(root >> 32));
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const unsigned int sh = user_ok[g] ? A.shared_thr[group * UPW + g * 32 + cold(j)] : 0u;
                    if (sh) thr[g] = fmaxf(thr[g], key_score((unsigned long long)sh << 32));
                }
            }
            if constexpr (BOUNDED && NG == 1) if (A.piece_best && !direct) {
                // A piece's own k-th best is a poor bound for a user whose sweep is cut into ~58 pieces of 1 700 items: rank ~1 000 of
                // the table at the piece's end, ~10 000 at its start — every piece stages ~100 candidates where one sweep would
                // stage 170 in all.  The pieces' BEST scores combine to much more: k slots, slot s collecting the maximum over
                // a k-th of the pieces (disjoint item ranges: k different items), and the minimum over the slots — valid as soon
                // as every slot is set — is the k-th largest of k sample maxima: rank ~1 000 after a few tiles, below 100 at the end.
                const int64_t rg = group - n_full;
                const int n_lists = (int)((((rg + 1) * n_tiles - 1) / A.run) - (rg * n_tiles) / A.run + 1);
                const int64_t b_own = group * UPW + cold(lane);
                if (n_lists >= k) {
                    if (owner && best && b_own < batch_n)
                        atomicMax(A.piece_best + b_own * k + (int)((int64_t)pidx * k / n_lists), (unsigned int)(best >> 32));
                    if (user_ok[0]) {
                        const unsigned int *slots = A.piece_best + (group * UPW + cold(j)) * k;
                        unsigned int m = ~0u;
                        for (int sl = 0; sl < k; ++sl) { const unsigned int v = slots[sl]; m = v < m ? v : m; }
                        if (m) thr[0] = fmaxf(thr[0], key_score((unsigned long long)m << 32));
                    }
                }
            }
#ifdef IGCN_TOPK_STATS
            asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(thr[0]) : "memory");
            st_7 += IGCN_CLOCK() - t_in;
#endif
        };

        // A operand: item row of this lane, k-slice of its half.  Address = uniform tile base (scalar
        // registers, advanced by scalar adds) + a lane offset that never changes, so a tile's loads
        // cost no vector ALU work.  Rows past the end of the table are clamped (and masked below).
        const unsigned lane_off = (unsigned)(j * (int)ldi + 4 * h) * 4u;       // bytes
        unsigned lane_off_last = lane_off;                       // for the ragged last tile of the table
        {
            const int64_t last_base = (int64_t)(n_tiles - 1) * 32;
            if (last_base + j >= n_items) lane_off_last = (unsigned)((int)(n_items - 1 - last_base) * (int)ldi + 4 * h) * 4u;
        }
        // Two A buffers at d <= 64 (registers allow it): the loads of the tile after next are issued at the START of
        // a block into the buffer the block does not use, so they have a whole block to land and nothing the
        // bookkeeping between two blocks waits for (vmcnt counts in order) is ever behind a load just issued.
        // At d = 128 one buffer: each piece is re-loaded as soon as the block has consumed it.
        constexpr bool kTwoBuffers = D <= 64;
        float4 a[D / 8], a2[(kTwoBuffers || kF16) ? D / 8 : 1];
        float4 a3[kF16 ? D / 8 : 1];                         // MODE 2: a third buffer (its tiles are half the size): loads run two tile steps ahead
        auto tile_addr = [&](int t, const char *&tile_ptr, unsigned &off) {
            tile_ptr = reinterpret_cast<const char *>(A.item_rows + (int64_t)t * 32 * ldi);
            off = t == n_tiles - 1 ? lane_off_last : lane_off;
        };
        auto load_into = [&](float4 (&buf)[D / 8], int t) {
            if constexpr (MODE == 1) {                             // [plane * 4 + k-step], one coalesced 1 KiB line set each
                const float4 *pk = A.packed + (int64_t)t * 8 * kWave + lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) buf[i] = pk[i * kWave];
            } else if constexpr (kF16) {                           // [k-step]
                const float4 *pk = A.packed + (int64_t)t * KS * kWave + lane;
#pragma unroll
                for (int i = 0; i < KS; ++i) buf[i] = pk[i * kWave];
            } else {
                const char *tile_ptr; unsigned off;
                tile_addr(t, tile_ptr, off);
#pragma unroll
                for (int q = 0; q < D / 8; ++q)
                    buf[q] = (FULL || 8 * q + 4 * h < A.d) ? *reinterpret_cast<const float4 *>(tile_ptr + off + 32 * q) : f4_zero();
            }
        };
        // MODE 1: the three plane products of a k-step, smallest first: (item plane, user plane)
        constexpr int kTermA[3] = {1, 0, 0};
        constexpr int kTermB[3] = {0, 1, 0};
        auto load_a = [&](int t) { load_into(a, t); };
        auto load_half = [&](float4 (&buf)[D / 8], int t, int half) {      // kRing12: k-steps 4 half .. 4 half + 3 of tile t
            const float4 *pk = A.packed + ((int64_t)t * KS + 4 * half) * kWave + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) buf[i] = pk[i * kWave];
        };
        // the chains of one tile, nothing interleaved (prologue of a piece)
        auto chain_plain = [&](f32x16 (&acc)[NG]) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
            if constexpr (MODE == 1) {
#pragma unroll
                for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[kTermA[tm] * 4 + st]),
                                                                             as_bf16x8(ub[g][kTermB[tm]][st]), acc[g], 0, 0, 0);
            } else if constexpr (kF16) {
#pragma unroll
                for (int tm = 0; tm < kUserPlanes; ++tm)         // the small term (user plane l) first
#pragma unroll
                    for (int st = 0; st < KS; ++st)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(kRing12 && st >= 4 ? a2[st - 4] : a[st]),
                                                                            as_half8(ub[g][kUserPlanes - 1 - tm][st]), acc[g], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < D / 8; ++q) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bfrag[g][4 * q + 0], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bfrag[g][4 * q + 1], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bfrag[g][4 * q + 2], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bfrag[g][4 * q + 3], acc[g], 0, 0, 0);
                    }
                }
            }
        };

        // ---- the parts of a tile's bookkeeping ------------------------------------------------------
        // Masks of a tile (ragged end of the piece, banned items, the users' exclusion cursors) never touch the
        // accumulators: they are collected as one bit per item in exm[g], already shifted so that bit
        // (r & 3) + 8 (r >> 2) belongs to register r of this lane, and are honoured when a row is staged (a masked
        // row goes in as -inf, so it can only fill a list that has fewer than k real entries).  A masked item with a
        // high score costs one needless look at its quad, nothing else.  mflag (wave-uniform): some exm is non-zero.
        auto build_masks = [&](int tile, int tile_base) {
#ifdef IGCN_TOPK_STATS
            const unsigned long long t_m = IGCN_CLOCK();
#endif
            if (mflag) {
#pragma unroll
                for (int g = 0; g < NG; ++g) exm[g] = 0u;
                mflag = false;
            }
            if (tile_base + 32 > item_hi) {                        // ragged last tile (wave-uniform)
                const unsigned beyond = item_hi > tile_base ? ~0u << (item_hi - tile_base) : ~0u;
#pragma unroll
                for (int g = 0; g < NG; ++g) exm[g] |= beyond >> (4 * h);
                mflag = true;
            }
            if (A.banned_bits) {
                const unsigned bm = ((const_u32_ptr)(uintptr_t)A.banned_bits)[tile];   // wave-uniform: one SCALAR load per tile
                if (bm) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) exm[g] |= bm >> (4 * h);
                    mflag = true;
                }
            }
            if (A.excl_rowptr) {
                const int tile_end = tile_base + 32;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    bool need = ex_next[g] < tile_end;
                    if (__any(need)) {
                        mflag = true;
                        do {
                            if (need) {
                                exm[g] |= (1u << (ex_next[g] - tile_base)) >> (4 * h);
                                ex_next[g] = ex_after[g];
                                ++ex_cur[g]; --ex_left[g];
                                ex_after[g] = ex_left[g] >= 1 ? *ex_cur[g] : kIdxNone;
                            }
                            need = ex_next[g] < tile_end;
                        } while (__any(need));
                    }
                }
            }
#ifdef IGCN_TOPK_STATS
            asm volatile("" : : "v"(exm[0]) : "memory");
            st_10 += IGCN_CLOCK() - t_m;
#endif
#ifdef IGCN_X_KEEPEXM
#pragma unroll
            for (int g = 0; g < NG; ++g) asm volatile("" : : "v"(exm[g]));      // developer ablation: the masks are computed, then not used
#endif
        };
        // One part of the selection of group g, quad q4 (rows 4 q4 .. 4 q4 + 3 of the lane): s = 0, 1 fold the
        // quad's maximum, s = 2 compares it with the user's threshold into a wave-wide mask (scalar registers).
        float qmax[NG];
        unsigned long long qmask[NG][kQuad];
#ifdef IGCN_X_NOSELECT
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int q4 = 0; q4 < kQuad; ++q4) qmask[g][q4] = 0ull;
#endif
        auto select_part = [&](const f32x16 &cur, int g, int q4, int s) {
            if (s == 0) qmax[g] = vmax3(cur[4 * q4], cur[4 * q4 + 1], cur[4 * q4 + 2]);
            else if (s == 1) qmax[g] = vmax2(qmax[g], cur[4 * q4 + 3]);
            else qmask[g][q4] = __builtin_amdgcn_fcmpf(qmax[g], thr[g], 3 /* oge */);
        };
        // The four rows of a quad that holds a candidate -> staging: (score, item id) goes to the lane's next free
        // staging slot whether it is a candidate or not; a candidate then advances the slot counter.  No branch.
        // The store is written as ds_write2_b32 by hand: the compiler would build a register pair with a move
        // for ds_write_b64.  flush() waits for these stores itself.  The caller has made sure of 4 free slots.
        auto stage_quad = [&](const f32x16 &cur, int g, int q4, int item_h, bool masks) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float sc = cur[4 * q4 + rr];
                if (masks && ((exm[g] >> (8 * q4 + rr)) & 1u)) sc = -INFINITY;
                const int item = item_h + 8 * q4 + rr;
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"(wpos[g]), "v"(sc), "v"(item) : "memory");
                wpos[g] += sc >= thr[g] ? 512u : 0u;
            }
        };
        // after a tile's selection: the quads with candidates, in order, each behind a scalar branch on its mask,
        // draining the staging lists first whenever some lane has fewer than 4 free slots
        auto stage_hits = [&](f32x16 (&cur)[NG], int tile_base) {
            IGCN_STAT(0, 1);
#ifdef IGCN_TOPK_STATS
            const unsigned long long t_hits = IGCN_CLOCK();
#endif
#ifdef IGCN_X_NOHITS
            if (tile_base != 0x7fffff00) return;                 // developer ablation build (never shipped)
#endif
            unsigned long long any = 0ull;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int q4 = 0; q4 < kQuad; ++q4) any |= qmask[g][q4];
            if (!any) return;
            IGCN_STAT(1, 1);
            if (prio_boost) __builtin_amdgcn_s_setprio(3);
            const int item_h = tile_base + 4 * h;
            // ONE copy of this loop (round 5: with a masked and an unmasked instantiation of it, chosen per tile, the default sweep was
            // 99 KB of code against a 64 KB instruction cache: one copy is 64 KB and 1-2 % faster); the tile's mask state picks the form
            // of the four staging stores only.
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int q4 = 0; q4 < kQuad; ++q4) {
                    if (qmask[g][q4]) {
                        IGCN_STAT(2, 1);
#ifdef IGCN_X_NOFLUSH
                        { const unsigned lim = stage_first(g) + ((unsigned)(cap - 4) << 9); wpos[g] = wpos[g] > lim ? lim : wpos[g]; }   // developer ablation: wrong results, no drain
#else
                        const bool full = wpos[g] > stage_first(g) + ((unsigned)(cap - 4) << 9);     // fewer than 4 free slots
                        if (__any(full)) {
                            if constexpr (kHotTrack) hot[g] += full ? 1 : 0;
                            flush();
                        }
#endif
#ifdef IGCN_X_NOMASKSTAGE
                        stage_quad(cur[g], g, q4, item_h, false);           // developer ablation build (never shipped): masked items are staged as they are
#else
                        if (mflag) stage_quad(cur[g], g, q4, item_h, true); else stage_quad(cur[g], g, q4, item_h, false);
#endif
                    }
                }
            }
            if (prio_boost) set_base_priority(prio_slot);
#ifdef IGCN_TOPK_STATS
            asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(wpos[0]) : "memory");
            st_8 += IGCN_CLOCK() - t_hits;
#endif
        };
        // One step of the software pipeline: `cur` holds the raw scores of tile t, a[] the A operand of tile
        // t+1.  The chains of tile t+1 (into `nxt`), the loads of tile t+2 (each a[q] as soon as the chains have
        // consumed it) and the selection of tile t are ONE basic block, its schedule written out by hand and
        // pinned (sched_barrier after every MFMA).
        auto tile_step = [&](f32x16 (&cur)[NG], f32x16 (&nxt)[NG], float4 (&ause)[D / 8], float4 (&aload)[D / 8], int tile, float4 (&ahi)[D / 8]) {
            const int tile_base = tile * 32;
            build_masks(tile, tile_base);
            constexpr int kSlots = MODE == 1 ? 12 * NG : kF16 ? kUserPlanes * KS * NG : (D / 2) * NG;   // MFMAs of the block
            constexpr int kParts = 3 * kQuad * NG;                // selection instructions of the block
            // the first ones wait until the previous block's MFMAs have long retired (a block of only 4 MFMAs — one 32-user group,
            // one fp16 plane per side at d = 64 — keeps two for that: the staging code between two blocks adds to the distance)
            constexpr int kFirst = kSlots >= 8 ? 4 : 2;
            if (tile + 1 < tin1) {
                const char *tile_ptr = nullptr; unsigned off = 0;
                if constexpr (MODE == 0) tile_addr(tile + 2 < tin1 ? tile + 2 : tin1 - 1, tile_ptr, off);   // (re-reads the last tile at the end)
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r) nxt[g][r] = 0.f;
                if constexpr (MODE == 1) {
#ifndef IGCN_X_NOLOADA
#ifdef IGCN_X_SAMETILE
                    const float4 *pk = A.packed + lane;           // developer build: every load hits the same lines
#else
                    const float4 *pk = A.packed + (int64_t)(tile + 2 < tin1 ? tile + 2 : tin1 - 1) * 8 * kWave + lane;
#endif
#pragma unroll
                    for (int i = 0; i < 8; ++i) aload[i] = pk[i * kWave];
#endif
                } else if constexpr (kF16) {
#ifndef IGCN_X_NOLOADA
#ifdef IGCN_X_SAMETILE
                    const float4 *pk = A.packed + lane;           // developer build: every load hits the same lines
#else
                    const float4 *pk = A.packed + (int64_t)(kRing12 ? (tile + 2 < tin1 ? tile + 2 : tin1 - 1)
                                                                     : (tile + 3 < tin1 ? tile + 3 : tin1 - 1)) * KS * kWave + lane;
#endif
#pragma unroll
                    for (int i = 0; i < (kRing12 ? 4 : KS); ++i) aload[i] = pk[i * kWave];
#endif
                } else if constexpr (kTwoBuffers) {
#pragma unroll
                    for (int q = 0; q < D / 8; ++q)
                        aload[q] = (FULL || 8 * q + 4 * h < A.d) ? *reinterpret_cast<const float4 *>(tile_ptr + off + 32 * q) : f4_zero();
                }
                __builtin_amdgcn_sched_barrier(0);
                auto select_share = [&](int slot) {             // this MFMA's share of tile t's selection instructions
#ifndef IGCN_X_NOSELECT
                    if (slot >= kFirst) {
                        const int p0 = (slot - kFirst) * kParts / (kSlots - kFirst);
                        const int p1 = (slot - kFirst + 1) * kParts / (kSlots - kFirst);
#pragma unroll
                        for (int p = 0; p < kParts; ++p)
                            if (p >= p0 && p < p1) {
                                const int pg = p % NG, pp = p / NG;
                                select_part(cur[pg], pg, pp / 3, pp % 3);
                            }
                    }
#endif
                };
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm) {
#pragma unroll
                        for (int st = 0; st < 4; ++st) {
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                nxt[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ause[kTermA[tm] * 4 + st]),
                                                                                 as_bf16x8(ub[g][kTermB[tm]][st]), nxt[g], 0, 0, 0);
                                select_share((tm * 4 + st) * NG + g);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                } else if constexpr (kF16) {
#pragma unroll
                    for (int tm = 0; tm < kUserPlanes; ++tm) {
#pragma unroll
                        for (int st = 0; st < KS; ++st) {
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                nxt[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(kRing12 && st >= 4 ? ahi[st - 4] : ause[st]),
                                                                                as_half8(ub[g][kUserPlanes - 1 - tm][st]), nxt[g], 0, 0, 0);
                                select_share((tm * KS + st) * NG + g);
#ifndef IGCN_X_NOLOADA
                                if constexpr (kRing12) {
                                    if (st == 3 && g == NG - 1) {      // lo is consumed: it takes the hi half of the tile after next
                                        const float4 *pk = A.packed + ((int64_t)(tile + 2 < tin1 ? tile + 2 : tin1 - 1) * KS + 4) * kWave + lane;
#pragma unroll
                                        for (int i = 0; i < 4; ++i) ause[i] = pk[i * kWave];
                                    }
                                }
#endif
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                } else {
#pragma unroll
                for (int q = 0; q < D / 8; ++q) {
                    const float4 aq = ause[q];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float av = c == 0 ? aq.x : c == 1 ? aq.y : c == 2 ? aq.z : aq.w;
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            nxt[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfrag[g][4 * q + c], nxt[g], 0, 0, 0);
                            select_share((4 * q + c) * NG + g);
#ifndef IGCN_X_NOLOADA
                            if (!kTwoBuffers && c == 3 && g == NG - 1)
                                aload[q] = (FULL || 8 * q + 4 * h < A.d) ? *reinterpret_cast<const float4 *>(tile_ptr + off + 32 * q) : f4_zero();
#endif
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                }
            } else {
                // last tile of the piece: no chain to hide behind; the MFMA results must have landed (the compiler
                // does not see the hand-written instructions' operands as MFMA results)
                asm volatile("s_nop 15\n\ts_nop 15" : : : "memory");
#pragma unroll
                for (int p = 0; p < kParts; ++p) {
                    const int pg = p % NG, pp = p / NG;
                    select_part(cur[pg], pg, pp / 3, pp % 3);
                }
            }
            stage_hits(cur, tile_base);
        };

#ifdef IGCN_TOPK_STATS
        st_12 += IGCN_CLOCK() - st_job;
#endif
        if constexpr (kWarm) {
            // ---- warm-up pass (round 4, late) ---------------------------------------------------------------------------------------
            // A sweep that starts from empty lists stages k (1 + ln(n / k)) ~ 240 candidates per user (k + extra = 26, 96 k items), half
            // of them in its first 2 % — where every quad holds candidates and a drain follows every other quad: the first 63 tiles took
            // 16 % of a wave's life, the first 255 29 % (profiles/r04zw_*).  So the first P tiles are multiplied TWICE: once here with no
            // selection at all — each of the lane's 16 accumulator registers keeps the best unmasked score it has held, 32 slots per user
            // over its two lanes, 32 different items — and the kc-th largest of the 32 slot maxima is a lower bound tau of the user's kc-th
            // best that >= kc items of these tiles reach.  The sweep proper then starts at tile 0 with thr = tau: ~45 candidates in
            // the first P tiles instead of ~140 (P = 64), and the same stream as before after them (simulated: 240 -> 143 per user at
            // P = 64, 108 at P = 256).  The scores are the sweep's own, bit for bit (same operands, same MFMA order), so tau is exact.
            // A list that is not full keeps tau as its threshold (flush()), and keeps its wave in the sweep (the exit check below).
            const int P = A.warm_tiles;
            warm = A.warm_thr != nullptr && P > 0 && job < A.n_whole && k <= 32 && tin1 - tin0 > P;
            // Only where nobody can leave during those tiles anyway: a wave leaves once |u| x (longest row still to come) < thr for
            // all its users, and on a trained table (rows at tile 128 a quarter as long as the first: 0.23-0.26 after 1-3 epochs,
            // 0.78 at random init) most waves have left after 6-12 tiles — the pass then costs more tiles than the sweep
            // (measured: 0.58 -> 0.72 ms after one epoch at P = 128, no gain at any P; profiles/r04zy_*).
            if (warm && A.tile_bound) {
                const float b0 = ((const_f32_ptr)(uintptr_t)A.tile_bound)[tin0], bp = ((const_f32_ptr)(uintptr_t)A.tile_bound)[tin0 + P];
                warm = bp >= kWarmFlat * b0;
            }
            if (warm) {
                int sv_left[NG];                                 // where the cursors stood (the pass walks them, the sweep proper starts over)
                float slot[NG][16];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    sv_left[g] = ex_left[g];
#pragma unroll
                    for (int r = 0; r < 16; ++r) slot[g][r] = -INFINITY;
                }
                auto warm_step = [&](float4 (&buf)[D / 8], int t) {
                    build_masks(t, t * 32);
                    f32x16 sc[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[g][r] = 0.f;
#pragma unroll
                    for (int st = 0; st < KS; ++st)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            sc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(buf[st]), as_half8(ub[g][0][st]), sc[g], 0, 0, 0);
                    if (mflag) {
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                slot[g][r] = fmaxf(slot[g][r], ((exm[g] >> (8 * (r >> 2) + (r & 3))) & 1u) ? -INFINITY : sc[g][r]);
                    } else {
                        // (fmaxf would canonicalise both operands first — three v_max_f32 per slot, 93 + 8 MFMAs per step; written by
                        // hand the compiler does not see the operands as MFMA results: the wait is spelled out, as before the last tile's
                        // selection)
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 15\n\ts_nop 15" : : : "memory");
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int r = 0; r < 16; ++r) vmax_into(slot[g][r], sc[g][r]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                const int t_end = tin0 + P;
                auto clamp_t = [&](int t) { return t < tin1 ? t : tin1 - 1; };
                load_into(a, tin0); load_into(a2, clamp_t(tin0 + 1)); load_into(a3, clamp_t(tin0 + 2));
                for (int t = tin0; t < t_end; t += 3) {
                    warm_step(a, t);
                    load_into(a, clamp_t(t + 3));
                    if (t + 1 < t_end) { warm_step(a2, t + 1); load_into(a2, clamp_t(t + 4)); }
                    if (t + 2 < t_end) { warm_step(a3, t + 2); load_into(a3, clamp_t(t + 5)); }
                }
                // the kc-th largest of a user's 32 slot maxima: drop the 32 - kc smallest (one per round, from whichever of the
                // user's two lanes holds it), the smallest left is tau.  Fewer than kc unmasked slots: tau = -inf, nothing changes.
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    for (int it = 0; it < 32 - k; ++it) {
                        float m = slot[g][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) m = fminf(m, slot[g][r]);
                        const float o = __shfl_xor(m, 32);
                        bool done = !(m < o || (m == o && h == 0));
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const bool hit = !done && slot[g][r] == m;
                            slot[g][r] = hit ? INFINITY : slot[g][r];
                            done = done || hit;
                        }
                    }
                    float m = slot[g][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fminf(m, slot[g][r]);
                    const float tau = fminf(m, __shfl_xor(m, 32));
                    if (user_ok[g]) {
                        thr[g] = tau;
                        if (h == 0) A.warm_thr[group * UPW + g * 32 + cold(j)] = tau;
                    }
                    // the sweep proper starts over at the piece's first tile
                    ex_cur[g] -= sv_left[g] - ex_left[g];
                    ex_left[g] = sv_left[g];
                    ex_next[g] = ex_left[g] >= 0 ? ex_cur[g][-1] : kIdxNone;
                    ex_after[g] = ex_left[g] >= 1 ? ex_cur[g][0] : kIdxNone;
                    exm[g] = 0u;
                }
                mflag = false;
                asm volatile("s_waitcnt vmcnt(0)" : : : "memory");     // (flush() reads the bound back, either lane of the user)
            }
        }
        // prologue: scores of the first tile, A operand of the second
        f32x16 acc_a[NG], acc_b[NG];
        if constexpr (kRing12) { load_half(a, tin0, 0); load_half(a2, tin0, 1); } else load_a(tin0);
        chain_plain(acc_a);
        if constexpr (kF16) {
            // ring of three item-tile buffers: step t multiplies tile t + 1 and requests tile t + 3 into the buffer step
            // t - 1 consumed; three steps per turn, the accumulator pair changes roles every step, so the odd turn
            // ends with a copy (32 moves per three tile steps)
            if constexpr (kRing12) {                           // next tile = (a3, a)
                const int t1 = tin0 + 1 < tin1 ? tin0 + 1 : tin1 - 1;
                load_half(a3, t1, 0);
                load_half(a, t1, 1);
            } else {
                load_into(a2, tin0 + 1 < tin1 ? tin0 + 1 : tin1 - 1);
                load_into(a3, tin0 + 2 < tin1 ? tin0 + 2 : tin1 - 1);
            }
            unsigned gone_next = 0u;
            float ureach[NG];                                  // s_u |u| of this lane's users: what a unit row can score at most
#pragma unroll
            for (int g = 0; g < NG; ++g)
                ureach[g] = A.tile_bound && user_ok[g] ? sqrtf(A.unorm2[group * UPW + g * 32 + j]) * ldexpf(1.f, -scale_exp(__uint_as_float(A.stats[2]))) : 0.f;
#ifdef IGCN_X_TURN6
            constexpr int kTurn = 6;
#else
            constexpr int kTurn = 3;
#endif
            for (int tile = tin0; tile < tin1; tile += kTurn) {
                // Exit check every 24 tiles — and every 6 up to tile 48 (round 4): on trained tables 99 % of the users are out of reach after
                // 8 tiles, yet the median wave swept 24 (154 us) because that was its first check, and the few crawling waves (2.5 drains
                // a tile, 15 us a tile) reached their first give-up opportunity at tile 48 after 720 us — the kernel's whole tail
                // (profiles/r04s_*).  At random init nobody leaves and the eight extra checks cost nothing measurable.
                const int rel = tile - tin0;
#ifdef IGCN_TOPK_STATS
                if (rel == 0) st_first = IGCN_CLOCK();
                if (rel == 63) st_14 += IGCN_CLOCK() - st_first;       // the lists' warm-up: 2 % of the table, half of the candidates
                if (rel == 255) st_15 += IGCN_CLOCK() - st_first;
#endif
                if constexpr (kHotTrack) {
                    if (rel == 3) {                                       // the lists' warm-up (every first item is a candidate) is over
#pragma unroll
                        for (int g = 0; g < NG; ++g) hot[g] = 0;
                    }
                }
                if (A.tile_bound && rel > 0 && (rel % 24 == 0 || (A.early_checks && rel < 48 && rel % A.early_checks == 0))) {
                    // Cauchy-Schwarz exit (every 24 tiles): the items come by descending norm, so if no user of this wave
                    // can still be reached by a row as long as this tile's longest, none of the remaining tiles matters
                    // (a user whose list is not full yet has thr = -inf and keeps the sweep alive)
                    // the thresholds in registers date from the last drain: early in the sweep, where they still move fast
                    // and where most waves of a trained model leave, they are refreshed first
                    if (tile - tin0 <= 192) flush();
                    // (a scalar load: the table was written by the packing kernel; a vector load here would wait for every
                    // item tile in flight)
                    const float reach = ((const_f32_ptr)(uintptr_t)A.tile_bound)[tile];
                    bool alive[NG], any_alive = false, far = false;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        alive[g] = user_ok[g] && ureach[g] * reach >= thr[g];
                        // (a list the warm-up bound has not filled yet: its items are among the first warm_tiles tiles)
                        if constexpr (kWarm) { if (warm) alive[g] = alive[g] || (user_ok[g] && heap_base[g * 32 + j] == 0ull); }
                        any_alive |= alive[g];
                        // far from done: the rows would have to get another third shorter (norms fall slowly in the tail).  (Round 4 also
                        // tried "still reachable N tiles further on" from the table of tile bounds, N = 24 ... 384: the same users within a
                        // few, the same time — profiles/r04q_*.)
                        far |= alive[g] && ureach[g] * reach >= 1.5f * thr[g];
                    }
                    if constexpr (kHotTrack) {
                        if (A.unfinished && job < A.n_whole && rel == 6 && A.early_checks) {
                            bool evict[NG];
                            unsigned long long ev = 0ull, al = 0ull;
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                const int other = __shfl_xor(hot[g], 32);
                                const int hu = hot[g] > other ? hot[g] : other;          // the user's two lanes stage different rows
                                evict[g] = alive[g] && ureach[g] * reach >= 1.5f * thr[g] && hu >= kHotDrains;
                                ev |= __ballot(evict[g] && h == 0) << (g ? 32 : 0);
                                al |= __ballot(alive[g] && h == 0) << (g ? 32 : 0);
                            }
                            const int n_evict = __popcll(ev), n_alive = __popcll(al);
                            if (n_evict > 0 && n_evict <= kMaxEvict && 2 * n_alive <= UPW) {
                                any_alive = false; far = false;
#pragma unroll
                                for (int g = 0; g < NG; ++g) {
                                    if (evict[g]) {
                                        if (h == 0) A.unfinished[group * UPW + g * 32 + cold(j)] = 1;
                                        user_ok[g] = false; thr[g] = INFINITY; alive[g] = false;
                                    }
                                    any_alive |= alive[g];
                                    far |= alive[g] && ureach[g] * reach >= 1.5f * thr[g];
                                }
                            }
                        }
                    }
                    if (!__any(any_alive)) {
                        if (A.exit_count && job < A.n_whole && lane == 0) atomicAdd(A.exit_count + exit_slot(job), 1u);
                        break;
                    }
                    // Giving up: a wave that is still alive long after three quarters of the waves have left would hold the
                    // kernel (one wave alone needs as long for a whole sweep as the full chip for all of them) for the sake
                    // of a few users.  Those users are handed to the fp32 sweep instead (re-scoring flags them; their
                    // candidates so far bound it from below), which is planned across the whole chip.  Which users take
                    // that way depends on timing; the lists do not.  With nobody leaving early (norms all alike) nobody gives up.
                    // (not before kGiveUpAfterTiles: handing users over costs a launch of its own, ~0.15 ms; the count is the one read at
                    // the previous check — its latency would otherwise stall the wave at every check)
                    // One counter per JOB (a wave's j-th whole sweep; kExitSlots of them, later jobs share the last): the leavers of
                    // a wave's first sweep must not count against its second (above 2 048 slots x 64 users per call
                    // every wave of job 1 that was alive at tile 48 handed its users over).  Only waves that LEFT EARLY or gave up
                    // are counted, not the ones that reached the end of the table: with those counted too, the last quarter
                    // of a random-init sweep — where all waves end together — gave up a few tiles before its end
                    // (measured: scoring 3.4 -> 5.0 ms, profiles/r04a_*).
                    const unsigned gone = gone_next;
                    const bool counting = A.exit_count && job < A.n_whole;
                    if (counting) gone_next = __hip_atomic_load(A.exit_count + exit_slot(job), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (counting && rel >= (A.early_checks ? 2 * A.early_checks : kGiveUpAfterTiles)) {
                        if ((uint64_t)gone * 4 >= (uint64_t)gridDim.x * 3 && __any(far)) {   // (a wave about to leave by itself stays)
                            if (lane == 0) atomicAdd(A.exit_count + exit_slot(job), 1u);
#pragma unroll
                            for (int g = 0; g < NG; ++g)
                                if (alive[g] && h == 0) A.unfinished[group * UPW + g * 32 + cold(j)] = 1;
                            break;
                        }
                    }
                }
                if constexpr (kRing12) {
                    tile_step(acc_a, acc_b, a3, a2, tile, a);
                    if (tile + 1 < tin1) tile_step(acc_b, acc_a, a2, a, tile + 1, a3);
                } else {
                    tile_step(acc_a, acc_b, a2, a, tile, a);
                    if (tile + 1 < tin1) tile_step(acc_b, acc_a, a3, a2, tile + 1, a);
                }
                if (tile + 2 < tin1) {
                    if constexpr (kRing12) tile_step(acc_a, acc_b, a, a3, tile + 2, a2);
                    else tile_step(acc_a, acc_b, a, a3, tile + 2, a);
#ifndef IGCN_X_TURN6
                    // a turn of three steps ends with the accumulator pair in swapped roles: 32 moves
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc_a[g] = acc_b[g];
#else
                    // Developer A/B (round 4, rejected): the second half of a SIX-step turn — the same three buffer roles with
                    // the accumulator pair swapped, so that the turn ends where it began and no accumulator is ever copied.
                    // 244 -> 238 VGPRs, scratch of the d = 128 variant 44 -> 12 bytes, and the same time within the run-to-run
                    // spread (d = 64: 2.93-3.47 vs 3.26-3.40 ms without masks, 3.47-3.58 vs 3.47-3.49 with; d = 128 4.3-4.7 vs
                    // 4.4-4.6: profiles/r04c_*): the moves are not on the wave's critical path, and the library took twice as long to compile.
                    if constexpr (kRing12) {
                        if (tile + 3 < tin1) tile_step(acc_b, acc_a, a3, a2, tile + 3, a);
                        if (tile + 4 < tin1) tile_step(acc_a, acc_b, a2, a, tile + 4, a3);
                        if (tile + 5 < tin1) tile_step(acc_b, acc_a, a, a3, tile + 5, a2);
                    } else {
                        if (tile + 3 < tin1) tile_step(acc_b, acc_a, a2, a, tile + 3, a);
                        if (tile + 4 < tin1) tile_step(acc_a, acc_b, a3, a2, tile + 4, a);
                        if (tile + 5 < tin1) tile_step(acc_b, acc_a, a, a3, tile + 5, a);
                    }
#endif
                }
            }
        } else {
        if (tin0 + 1 < tin1) load_a(tin0 + 1);
        for (int tile = tin0; tile < tin1; tile += 2) {
            if constexpr (kTwoBuffers) {
                tile_step(acc_a, acc_b, a, a2, tile, a);
                if (tile + 1 < tin1) tile_step(acc_b, acc_a, a2, a, tile + 1, a);
            } else {
                tile_step(acc_a, acc_b, a, a, tile, a);
                if (tile + 1 < tin1) tile_step(acc_b, acc_a, a, a, tile + 1, a);
            }
        }
        }
        flush();
#ifdef IGCN_TOPK_STATS
        st_job = IGCN_CLOCK();
#endif

        // ---- emit: every owner lane heapsorts its user's list in place (best first) and writes it to the
        // output, or to the workspace when the user's sweep was cut into pieces ----------------------
        for (int n = k - 1; n > 0; --n) {
            const unsigned long long last = heap[n * kWave];
            heap[n * kWave] = heap[0];                           // current minimum goes to the end
            heap_replace_root(heap, n, last);
        }
        // The lists leave through the WAVE, not through their owner lanes: user l of the wave-group sits in column l of the heap
        // array ([slot][lane]), the wave's users are consecutive batch positions, so [users here][k] is one contiguous block of the
        // output (a run per user in the pieces' workspace) and lane e mod 64 writes its element e — 512 contiguous bytes per store
        // instruction instead of 64 scattered 8-byte pieces.  (Round 4: on trained tables every wave finishes within the same
        // ~100 us and the owner lanes' scattered stores took 178 k cycles per wave — half of the wave's life; profiles/r04y_*.)
        const int64_t b0 = group * UPW;
        const int n_here = batch_n - b0 < UPW ? (int)(batch_n - b0) : UPW;
        bool by_rows = false;
        if constexpr (BOUNDED) by_rows = A.rows != nullptr;
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        __builtin_amdgcn_wave_barrier();
        if (!by_rows) {
            const int total = n_here * k;
            for (int e = cold(lane); e < total; e += kWave) {
                const int u = e / k, r = e - u * k;
                const unsigned long long key = heap_base[r * kWave + u];
                if (direct) {
                    A.out_idx[b0 * k + e] = key ? key_item(key) : -1;
                    A.out_val[b0 * k + e] = key ? key_score(key) : -INFINITY;
                } else {
                    const int64_t slot = ((b0 + u - n_full * UPW) * A.p_max + pidx) * k + r;
                    A.ws_val[slot] = key ? key_score(key) : -INFINITY;
                    A.ws_idx[slot] = key ? key_item(key) : kIdxNone;
                }
            }
        } else {
            // (the two-stage path's fall-back: a handful of users whose output rows are scattered over the batch)
            const int64_t b_own = b0 + cold(lane);
            if (owner && b_own < batch_n) {
                if (direct) {
                    int64_t orow = b_own;
                    if constexpr (BOUNDED) orow = A.rows[b_own];
                    for (int r = 0; r < k; ++r) {
                        const unsigned long long key = heap[r * kWave];
                        A.out_idx[orow * k + r] = key ? key_item(key) : -1;
                        A.out_val[orow * k + r] = key ? key_score(key) : -INFINITY;
                    }
                } else {
                    const int64_t slot = ((b_own - n_full * UPW) * A.p_max + pidx) * k;
                    for (int r = 0; r < k; ++r) {
                        const unsigned long long key = heap[r * kWave];
                        A.ws_val[slot + r] = key ? key_score(key) : -INFINITY;
                        A.ws_idx[slot + r] = key ? key_item(key) : kIdxNone;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef IGCN_TOPK_STATS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");
        st_13 += IGCN_CLOCK() - st_job;
#endif
    }
#ifdef IGCN_TOPK_STATS
    {
        // st_5 is per lane (staged candidates this owner lane drained): sum over the wave
        unsigned long long c5 = st_5;
        for (int off = 32; off > 0; off >>= 1) c5 += __shfl_xor(c5, off);
        if (lane == 0) {
            atomicAdd(&g_topk_stats[0], st_0); atomicAdd(&g_topk_stats[1], st_1); atomicAdd(&g_topk_stats[2], st_2);
            atomicAdd(&g_topk_stats[3], st_3); atomicAdd(&g_topk_stats[4], st_4); atomicAdd(&g_topk_stats[5], c5);
            atomicAdd(&g_topk_stats[6], 1ull);
            atomicAdd(&g_topk_stats[7], st_7); atomicAdd(&g_topk_stats[8], st_8);
            atomicAdd(&g_topk_stats[9], IGCN_CLOCK() - st_begin); atomicAdd(&g_topk_stats[10], st_10);
            atomicAdd(&g_topk_stats[11], __builtin_amdgcn_s_memrealtime() - st_rt_begin);   // 100 MHz ticks
            atomicAdd(&g_topk_stats[12], st_12); atomicAdd(&g_topk_stats[13], st_13);
            atomicAdd(&g_topk_stats[14], st_14); atomicAdd(&g_topk_stats[15], st_15);
            if (blockIdx.x < 4096) {
                g_topk_wave_times[3 * blockIdx.x] = st_rt_begin;
                g_topk_wave_times[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
                g_topk_wave_times[3 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |
                                                        ((st_0 & 0xFFFFull) << 32) | ((st_3 & 0xFFFFull) << 48);   // + tiles swept, flushes
            }
        }
    }
#endif
}

// One wave per user of the groups that were cut: lane = one piece's list, already best-first; k rounds of a
// wave-wide arg-best over the heads.
__global__ __launch_bounds__(kBlock) void topk_merge_kernel(const float *__restrict__ ws_val, const int32_t *__restrict__ ws_idx,
                                                            int64_t first_user, int64_t batch, int n_tiles, int64_t run,
                                                            int p_max, int k, int upw,
                                                            int64_t *__restrict__ out_idx, float *__restrict__ out_val,
                                                            const int32_t *__restrict__ rows, const int32_t *__restrict__ count_dev)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t rb = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    int64_t b = first_user + rb;
    if (count_dev && *count_dev < batch) batch = *count_dev;      // the batch is a device list (TopkArgs::rows / count_dev)
    if (b >= batch) return;
    if (rows) b = rows[b];
    const int64_t rg = rb / upw;
    const int n_lists = (int)((((rg + 1) * n_tiles - 1) / run) - (rg * n_tiles) / run + 1);
    const bool live = lane < n_lists;
    const float *v = ws_val + (rb * p_max + (live ? lane : 0)) * k;
    const int32_t *ix = ws_idx + (rb * p_max + (live ? lane : 0)) * k;
    int cur = 0;
    float hv = -INFINITY;
    int hi = kIdxNone;
    if (live) { hv = v[0]; hi = ix[0]; }
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
        if (lane == bl && live) {
            ++cur;
            if (cur < k) { hv = v[cur]; hi = ix[cur]; } else { hv = -INFINITY; hi = kIdxNone; }
        }
    }
}

// ... and with up to Q lists per lane, for the narrow bounded sweep's up to 232 pieces: a lane's best head, then the wave's.
template <int Q>
__global__ __launch_bounds__(kBlock) void topk_merge_wide_kernel(const float *__restrict__ ws_val, const int32_t *__restrict__ ws_idx,
                                                                 int64_t first_user, int64_t batch, int n_tiles, int64_t run,
                                                                 int p_max, int k, int upw,
                                                                 int64_t *__restrict__ out_idx, float *__restrict__ out_val,
                                                                 const int32_t *__restrict__ rows, const int32_t *__restrict__ count_dev)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t rb = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    int64_t b = first_user + rb;
    if (count_dev && *count_dev < batch) batch = *count_dev;
    if (b >= batch) return;
    if (rows) b = rows[b];
    const int64_t rg = rb / upw;
    const int n_lists = (int)((((rg + 1) * n_tiles - 1) / run) - (rg * n_tiles) / run + 1);
    const float *v = ws_val + rb * p_max * k;
    const int32_t *ix = ws_idx + rb * p_max * k;
    float hv[Q];
    int hi[Q], cur[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int list = lane * Q + q;
        const bool live = list < n_lists;
        hv[q] = live ? v[(int64_t)list * k] : -INFINITY;
        hi[q] = live ? ix[(int64_t)list * k] : kIdxNone;
        cur[q] = 0;
    }
    for (int r = 0; r < k; ++r) {
        float mv = hv[0];
        int mi = hi[0], mq = 0;
#pragma unroll
        for (int q = 1; q < Q; ++q)
            if (ranks_before(hv[q], hi[q], mv, mi)) { mv = hv[q]; mi = hi[q]; mq = q; }
        float bv = mv;
        int bi = mi, bl = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off), ol = __shfl_xor(bl, off);
            if (ranks_before(ov, oi, bv, bi) || (ov == bv && oi == bi && ol < bl)) { bv = ov; bi = oi; bl = ol; }
        }
        if (lane == 0) { out_idx[b * k + r] = bi == kIdxNone ? -1 : bi; out_val[b * k + r] = bv; }
        if (lane == bl) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (q == mq && lane * Q + q < n_lists) {
                    ++cur[q];
                    const bool more = cur[q] < k;
                    hv[q] = more ? v[(int64_t)(lane * Q + q) * k + cur[q]] : -INFINITY;
                    hi[q] = more ? ix[(int64_t)(lane * Q + q) * k + cur[q]] : kIdxNone;
                }
        }
    }
}

// The same merge with one LANE per user, for plans that cut a group's sweep into at most P pieces (the usual
// case: 2-3): the lane keeps the heads of its user's lists in registers and emits the best of them k times.  A wave
// per user spends 18 cross-lane operations per output on 2-3 live lanes; this one spends P compares.
template <int P>
__global__ __launch_bounds__(kBlock) void topk_merge_lanes_kernel(const float *__restrict__ ws_val, const int32_t *__restrict__ ws_idx,
                                                                  int64_t first_user, int64_t batch, int n_tiles, int64_t run,
                                                                  int p_max, int k, int upw,
                                                                  int64_t *__restrict__ out_idx, float *__restrict__ out_val,
                                                                  const int32_t *__restrict__ rows, const int32_t *__restrict__ count_dev)
{
    const int64_t rb = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int64_t b = first_user + rb;
    if (count_dev && *count_dev < batch) batch = *count_dev;
    if (b >= batch) return;
    if (rows) b = rows[b];
    const int64_t rg = rb / upw;
    const int n_lists = (int)((((rg + 1) * n_tiles - 1) / run) - (rg * n_tiles) / run + 1);
    const float *v = ws_val + rb * p_max * k;
    const int32_t *ix = ws_idx + rb * p_max * k;
    float hv[P];
    int hi[P], cur[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const bool live = q < n_lists;
        hv[q] = live ? v[q * k] : -INFINITY;
        hi[q] = live ? ix[q * k] : kIdxNone;
        cur[q] = 0;
    }
    for (int r = 0; r < k; ++r) {
        float bv = hv[0];
        int bi = hi[0], bq = 0;
#pragma unroll
        for (int q = 1; q < P; ++q)
            if (ranks_before(hv[q], hi[q], bv, bi)) { bv = hv[q]; bi = hi[q]; bq = q; }    // ties: the lower list wins
        out_idx[b * k + r] = bi == kIdxNone ? -1 : bi;
        out_val[b * k + r] = bv;
#pragma unroll
        for (int q = 0; q < P; ++q)
            if (q == bq && q < n_lists) {
                ++cur[q];
                const bool more = cur[q] < k;
                hv[q] = more ? v[q * k + cur[q]] : -INFINITY;
                hi[q] = more ? ix[q * k + cur[q]] : kIdxNone;
            }
    }
}

// banned uint8 [n_items] -> one bit per item, one 32-bit word per 32-item tile
__global__ __launch_bounds__(kBlock) void topk_pack_banned_kernel(const uint8_t *__restrict__ banned, int64_t n_items, int n_tiles,
                                                                  const int32_t *__restrict__ perm, uint32_t *__restrict__ bits)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t pair = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);     // a wave packs two tiles
    const int64_t item = pair * kWave + lane;                                              // sweep position
    const bool b = item < n_items && banned[perm ? perm[item] : item] != 0;
    const unsigned long long m = __ballot(b);
    if (lane == 0) {
        if (2 * pair < n_tiles) bits[2 * pair] = (uint32_t)m;
        if (2 * pair + 1 < n_tiles) bits[2 * pair + 1] = (uint32_t)(m >> 32);
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

// calculate_metrics (trainer.py:109-138) in one pass over the recommended lists: per user the membership tests of
// hit_matrix_kernel, then for every cut-off k_t: hits / k_t, hits / |list|, DCG / IDCG in float32 as the reference forms
// them; users with an empty list do not count (trainer.py:133).  Sums over users in float64: per-workgroup partial sums
// to the workspace, added up in workgroup order by the last kernel — the same bits on every run, no float atomics.
struct MetricCuts { int k[IGCN_MAX_METRIC_CUTS]; int n; };
__global__ __launch_bounds__(kBlock) void eval_metrics_kernel(const int64_t *__restrict__ rec, int64_t n_users, int k_rec,
                                                              const int64_t *__restrict__ eval_rowptr, const int32_t *__restrict__ eval_col,
                                                              const MetricCuts cuts, double *__restrict__ partial)
{
    constexpr int NV = 3 * IGCN_MAX_METRIC_CUTS + 1;
    __shared__ double sh[kBlock / kWave][NV];
    const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.0;
    if (u < n_users) {
        const int64_t lo0 = eval_rowptr[u], end = eval_rowptr[u + 1];
        const int len = (int)(end - lo0);
        if (len > 0 && eval_col) {
            float hits[IGCN_MAX_METRIC_CUTS], dcg[IGCN_MAX_METRIC_CUTS], idcg[IGCN_MAX_METRIC_CUTS];
#pragma unroll
            for (int t = 0; t < IGCN_MAX_METRIC_CUTS; ++t) hits[t] = dcg[t] = idcg[t] = 0.f;
            for (int j = 0; j < k_rec; ++j) {
                const int64_t item = rec[u * k_rec + j];
                int64_t lo = lo0, hi = end;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (eval_col[mid] < item) lo = mid + 1; else hi = mid;
                }
                const float h = (lo < end && eval_col[lo] == item) ? 1.f : 0.f;
                const float denom = log2f((float)(j + 2));
                const float hd = h / denom, id = (j < len ? 1.f : 0.f) / denom;
#pragma unroll
                for (int t = 0; t < IGCN_MAX_METRIC_CUTS; ++t)
                    if (t < cuts.n && j < cuts.k[t]) { hits[t] += h; dcg[t] += hd; idcg[t] += id; }
            }
#pragma unroll
            for (int t = 0; t < IGCN_MAX_METRIC_CUTS; ++t)
                if (t < cuts.n) {
                    v[3 * t] = (double)(hits[t] / (float)cuts.k[t]);
                    v[3 * t + 1] = (double)(hits[t] / (float)len);
                    v[3 * t + 2] = (double)(dcg[t] / idcg[t]);
                }
            v[NV - 1] = 1.0;
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double x = v[i];
        for (int o = kWave / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if ((threadIdx.x & (kWave - 1)) == 0) sh[threadIdx.x >> 6][i] = x;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double x = 0.0;
        for (int w = 0; w < kBlock / kWave; ++w) x += sh[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * NV + threadIdx.x] = x;
    }
}
__global__ __launch_bounds__(kWave) void eval_metrics_finish_kernel(const double *__restrict__ partial, int64_t n_blocks, double *__restrict__ out)
{
    constexpr int NV = 3 * IGCN_MAX_METRIC_CUTS + 1;
    if (threadIdx.x >= NV) return;
    double x = 0.0;
    for (int64_t b = 0; b < n_blocks; ++b) x += partial[b * NV + threadIdx.x];
    out[threadIdx.x] = x;
}

// Item table -> MFMA-ready bf16 planes for MODE 1: [tile][plane 0..1][k-step 0..3][lane 0..63] x 16 B, lane (j, kg)
// holding k = 16 s + 8 kg .. + 7 of item 32 tile + j (rows past the end: zeros).  One thread per (tile, k-step, lane).
__global__ __launch_bounds__(kBlock) void topk_pack_items_kernel(const float *__restrict__ item_rows, int64_t ldi, int64_t n_items,
                                                                 int n_tiles, const int32_t *__restrict__ perm, float4 *__restrict__ packed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_tiles * 4 * kWave) return;
    const int lane = (int)(i % kWave);
    const int s = (int)(i / kWave % 4);
    const int64_t tile = i / (4 * kWave);
    const int64_t item = tile * 32 + (lane & 31);                 // sweep position
    float4 lo = f4_zero(), hi = f4_zero();
    if (item < n_items) {
        const float *src = item_rows + (int64_t)(perm ? perm[item] : item) * ldi + 16 * s + 8 * (lane >> 5);
        lo = *reinterpret_cast<const float4 *>(src);
        hi = *reinterpret_cast<const float4 *>(src + 4);
    }
    float4 planes[2];
    split2_x8(lo, hi, planes);
#pragma unroll
    for (int p = 0; p < 2; ++p) packed[((tile * 2 + p) * 4 + s) * kWave + lane] = planes[p];
}

// stats[0] = max over the items of |row|^2, stats[1] = max |item element| (a group of 2^lg = d / 4 lanes per row: 16 at
// d = 64, 32 at d = 128), as the bit
// patterns of non-negative floats; norm2_out (optional): |row|^2 of every row (what the sweep order sorts by).  A small
// fixed grid walks the table; the maxima are reduced inside the workgroup and ONE lane per workgroup issues the atomics
// (one atomic per wave from 2 048 waves on the same two words took 43 us).
// With ids: the rows ids[0..n) of the table, and only the element maximum, into stats[2] (the users of a call).
__device__ __forceinline__ void row_stats_body(const float *__restrict__ rows, int64_t ld, int64_t n,
                                               const int64_t *__restrict__ ids, int of_users, int lg,
                                               unsigned int *__restrict__ stats, float *__restrict__ norm2_out,
                                               unsigned block, unsigned blocks)
{
    const int lpr = 1 << lg;
    __shared__ float sh[2][kBlock / kWave];
    const int64_t t0 = (int64_t)block * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)blocks * blockDim.x;
    float best_n2 = 0.f, best_el = 0.f;
    for (int64_t t = t0; (t >> lg) < n + 3; t += stride) {                // (+3: the groups of a wave stay together)
        const int64_t r = t >> lg;
        float n2 = 0.f;
        if (r < n) {
            const int64_t row = ids ? ids[r] : r;
            const float4 v = *reinterpret_cast<const float4 *>(rows + row * ld + 4 * (t & (lpr - 1)));
            n2 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            best_el = fmaxf(best_el, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        for (int o = 1; o < lpr; o <<= 1) n2 += __shfl_xor(n2, o);
        if (norm2_out && r < n && (t & (lpr - 1)) == 0) norm2_out[r] = n2;
        best_n2 = fmaxf(best_n2, n2);
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) best_el = fmaxf(best_el, __shfl_xor(best_el, o));
    for (int o = lpr; o < kWave; o <<= 1) best_n2 = fmaxf(best_n2, __shfl_xor(best_n2, o));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & (kWave - 1)) == 0) { sh[0][w] = best_n2; sh[1][w] = best_el; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kBlock / kWave; ++i) { best_n2 = fmaxf(best_n2, sh[0][i]); best_el = fmaxf(best_el, sh[1][i]); }
        // (a maximum that does not beat what is already there needs no atomic: after the first few workgroups almost none does,
        // which is what lets the grid be eight workgroups per CU instead of one — round 4: 23 + 33 us -> see profiles/r04*)
        auto raise = [](unsigned int *p, float v) {
            const unsigned int b = __float_as_uint(v);
            if (b > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, b);
        };
        if (of_users) {
            raise(stats + 2, best_el);
        } else {
            raise(stats, best_n2);
            raise(stats + 1, best_el);
        }
    }
}
// The two tables of igcn_score_topk_fast_f32 in ONE launch (late round 4: the users' pass used to wait behind the order build for
// no reason, 20 us of a mostly idle GPU): workgroups [0, item_blocks) walk the items, the others the batch's users.  Thread 0 also
// clears the caller's flagged count (one memset launch less; the re-scoring kernel that counts into it is queued behind).
__global__ __launch_bounds__(kBlock) void topk_row_stats_both_kernel(const float *__restrict__ item_rows, int64_t ldi, int64_t n_items,
                                                                     float *__restrict__ item_norm2, unsigned item_blocks,
                                                                     const float *__restrict__ user_rows, int64_t ldu, int64_t batch,
                                                                     const int64_t *__restrict__ user_ids, float *__restrict__ user_norm2,
                                                                     int lg, unsigned int *__restrict__ stats, int32_t *__restrict__ flagged)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && flagged) flagged[0] = 0;
    if (blockIdx.x < item_blocks) row_stats_body(item_rows, ldi, n_items, nullptr, 0, lg, stats, item_norm2, blockIdx.x, item_blocks);
    else row_stats_body(user_rows, ldu, batch, user_ids, 1, lg, stats, user_norm2, blockIdx.x - item_blocks, gridDim.x - item_blocks);
}

// Item table -> one MFMA-ready fp16 plane for MODE 2: [tile][k-step 0..ks-1][lane 0..63] x 16 B (ks = d / 16), scaled by
// the power of two that brings the largest element (stats[1]) into [0.5, 1).  One thread per (tile, k-step, lane).
__global__ __launch_bounds__(kBlock) void topk_pack_items_f16_kernel(const float *__restrict__ item_rows, int64_t ldi, int64_t n_items,
                                                                     int n_tiles, int ks, const unsigned int *__restrict__ stats,
                                                                     const int32_t *__restrict__ perm, float4 *__restrict__ packed,
                                                                     const float *__restrict__ norm2, float *__restrict__ tile_bound,
                                                                     const uint32_t *__restrict__ order_status)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_tiles * ks * kWave) return;
    const int lane = (int)(i % kWave);
    const int s = (int)(i / kWave % ks);
    const int64_t tile = i / (ks * kWave);
    const int64_t item = tile * 32 + (lane & 31);                 // sweep position
    if (tile_bound && lane == 0 && s == 0)
        // the tile's first row is its longest up to the sort's granularity (norms ordered on 16 bits: < 2^-8 apart
        // in |row|^2 inside a bucket); 1 % covers that, the fp16 rounding of the plane and the fp32 accumulation
        // (order_status != 0: the order build fell back to the id order, topk_order.hip — rows no longer get shorter along the sweep,
        // so no tile bounds the ones behind it: a bound nobody gets below keeps every wave in the sweep to its end)
        tile_bound[tile] = order_status && order_status[0] ? 3.0e38f
                                                           : 1.01f * sqrtf(norm2[perm[item]]) * ldexpf(1.f, -scale_exp(__uint_as_float(stats[1])));
    float4 lo = f4_zero(), hi = f4_zero();
    if (item < n_items) {
        const float *src = item_rows + (int64_t)(perm ? perm[item] : item) * ldi + 16 * s + 8 * (lane >> 5);
        lo = *reinterpret_cast<const float4 *>(src);
        hi = *reinterpret_cast<const float4 *>(src + 4);
    }
    float4 planes[2];
    split_f16_x8(lo, hi, ldexpf(1.f, -scale_exp(__uint_as_float(stats[1]))), planes);
    packed[(tile * ks + s) * kWave + lane] = planes[0];
}

// Second stage of the bf16 path, one wave per user, lane c = candidate c of the sweep (kc = k + kFastExtra <= 64 of
// them, best approximate score first).  The candidate's score is recomputed in fp32 in the order the fp32 sweep adds
// the products (k = 8q + c, 8q + 4 + c), the candidates are ranked by (exact score, lower id first) and the best k
// written out.  Then the check that makes the result that of the fp32 sweep: an item the sweep dropped has an
// approximate score <= a_min (the smallest kept) and an exact one <= a_min + eps, eps = c |u| max|i| bounding what the
// reduced-precision operands and the fp32 accumulation can be off by; if a_min + eps does not stay below the k-th
// exact score, a dropped item could belong to the list (or tie with its tail) and the user is flagged:
// flagged[1 + n] = position of the user in the batch, flagged[0] = n.
template <int LANES>                      // lanes of a user: 64, or 32 (two users per wave) when kc <= 32
__global__ __launch_bounds__(kBlock) void topk_rescore_kernel(const float *__restrict__ user_rows, int64_t ldu,
                                                              const int64_t *__restrict__ user_ids, int64_t batch,
                                                              const float *__restrict__ item_rows, int64_t ldi,
                                                              const int64_t *__restrict__ cand_idx, const float *__restrict__ cand_val,
                                                              int kc, int k, int d, const unsigned int *__restrict__ stats, int mode,
                                                              const int32_t *__restrict__ perm,
                                                              int64_t *__restrict__ out_idx, float *__restrict__ out_val,
                                                              int32_t *__restrict__ flagged, float *__restrict__ flagged_thr,
                                                              const uint8_t *__restrict__ unfinished, uint8_t *__restrict__ flagged_weak)
{
    constexpr int UPW = kWave / LANES;                            // users per wave
    const int lane = threadIdx.x & (LANES - 1);
    const int base = (threadIdx.x & (kWave - 1)) - lane;          // first lane of this user's lane group
    const int64_t b = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * UPW + (base ? 1 : 0);
    if (b >= batch) return;                                       // (a whole lane group leaves: the shuffles below stay inside a group)
    const int64_t uid = user_ids ? user_ids[b] : b;
    const float *u = user_rows + uid * ldu;
    const bool live = lane < kc;
    int64_t item = live ? cand_idx[b * kc + lane] : -1;           // a sweep position
    if (perm && item >= 0) item = perm[item];
    const float approx = live ? cand_val[b * kc + lane] : -INFINITY;
    const bool real = item >= 0 && approx > -INFINITY;            // -inf: a masked item filling a short list, or an empty slot
    float exact = -INFINITY, un2 = 0.f;
    {
        const float *it = item_rows + (real ? item : 0) * ldi;
        float acc = 0.f;
#pragma unroll 8
        for (int q = 0; q < d / 8; ++q) {
            const float4 ua = *reinterpret_cast<const float4 *>(u + 8 * q), ub = *reinterpret_cast<const float4 *>(u + 8 * q + 4);
            const float4 ia = *reinterpret_cast<const float4 *>(it + 8 * q), ib = *reinterpret_cast<const float4 *>(it + 8 * q + 4);
            acc = fmaf(ia.x, ua.x, acc); acc = fmaf(ib.x, ub.x, acc);
            acc = fmaf(ia.y, ua.y, acc); acc = fmaf(ib.y, ub.y, acc);
            acc = fmaf(ia.z, ua.z, acc); acc = fmaf(ib.z, ub.z, acc);
            acc = fmaf(ia.w, ua.w, acc); acc = fmaf(ib.w, ub.w, acc);
            un2 += ua.x * ua.x + ua.y * ua.y + ua.z * ua.z + ua.w * ua.w + ub.x * ub.x + ub.y * ub.y + ub.z * ub.z + ub.w * ub.w;
        }
        if (real) exact = acc;
    }
    const int id32 = item >= 0 ? (int)item : kIdxNone;
    int rank = 0;
    for (int c = 0; c < kc; ++c) {
        const float ov = __shfl(exact, base + c);
        const int oi = __shfl(id32, base + c);
        rank += (ranks_before(ov, oi, exact, id32) || (ov == exact && oi == id32 && c < lane)) ? 1 : 0;
    }
    if (live && rank < k) {
        out_idx[b * k + rank] = item >= 0 ? item : -1;
        out_val[b * k + rank] = exact;
    }
    // the k-th exact score, the smallest kept approximate score, how many candidates are real
    float e_k = live && rank == k - 1 ? exact : -INFINITY;
    float a_min = real ? approx : INFINITY;
    int n_real = real ? 1 : 0;
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) {
        e_k = fmaxf(e_k, __shfl_xor(e_k, o));
        a_min = fminf(a_min, __shfl_xor(a_min, o));
        n_real += __shfl_xor(n_real, o);
    }
    if (lane == 0) {
        // MODE 1: two bf16 planes each side (8 significant bits each: x = h + l + r, |l| <= 2^-8 |x|, |r| <= 2^-16 |x|), three
        // products kept, l_i l_u and the r terms dropped: 3 * 2^-16 per product, 2^-14 with the accumulation.  MODE 2: items
        // rounded to fp16 once (11 bits: 2^-11 each), users
        // exact to 2^-22, fp32 accumulation of 2 d products 2^-18 (d = 64) / 2^-17 (d = 128): 2^-11 (1 + 2^-5) / 2^-11 (1 + 2^-4);
        // its approximate scores carry the two tables' scales.
        // mode 3: users rounded to fp16 once too: 2^-10 (1 + 2^-11) + the accumulation
        const float coef = mode == 3 ? (d > 64 ? 0x1.08p-10f : 0x1.04p-10f) : mode == 2 ? (d > 64 ? 0x1.1p-11f : 0x1.08p-11f) : 0x1p-14f;
        if (mode >= 2) a_min = ldexpf(a_min, scale_exp(__uint_as_float(stats[1])) + scale_exp(__uint_as_float(stats[2])));
        float eps = coef * sqrtf(un2 * __uint_as_float(stats[0]));
        // MODE 2: scaled elements below 2^-14 are fp16 subnormals (absolute error <= 2^-25 each, both sides): at most
        // 2 d 2^-25 per score in scaled units (2^-18 at d = 64) = 2^-16 (d / 64) max|i_j| max|u_j| — matters only for a
        // user far smaller than the largest of the batch, whom it then sends to the fp32 sweep
        if (mode >= 2 && eps > 0.f) eps += (d > 64 ? 0x1p-15f : 0x1p-16f) * __uint_as_float(stats[1]) * __uint_as_float(stats[2]);
        // fewer real candidates than slots: the sweep dropped nothing real.  eps == 0 (an all-zero user): scores are exact —
        // enough in an id-order sweep (ties are kept by lower id there too); in a permuted sweep the ties at the end of
        // the candidate list were kept by POSITION, so the user goes to the fp32 sweep (a_min + 0 < e_k fails on a tie).
        // Fewer than k real candidates: the list ends with masked fill-ins, which the fp32 sweep picks by lower id and a
        // permuted sweep by lower position: that user goes to the fp32 sweep too.
        // (unfinished: the sweep gave up on this user before its end — nothing is known about the items it did not see)
        const bool ok = !(unfinished && unfinished[b]) &&
                        ((n_real < kc && (!perm || n_real >= k)) || (eps == 0.f && !perm) || a_min + eps < e_k);
        if (!ok) {
            // e_k, the k-th exact score among real (unmasked) candidates, is a lower bound of the user's k-th best:
            // the fp32 sweep that re-does this user starts from it instead of from an empty list
            const int slot = atomicAdd(flagged, 1);
            flagged[1 + slot] = (int32_t)b;
            if (flagged_thr) flagged_thr[slot] = e_k;
            // (a user its wave gave up on has seen a few tiles only: its bound admits thousands of items — not a case for the streaming filter)
            if (flagged_weak && slot < (int)kFastFallbackMaxInt) flagged_weak[slot] = unfinished && unfinished[b] ? 1 : 0;
        }
    }
}

// ---- the flagged users of the two-stage path, as a streaming filter (late round 4) ------------------------------------------------
// A flagged user comes with a lower bound of its k-th best score that is nearly always that score itself (the k-th exact score among
// its candidates).  The bounded fp32 sweep needs ~85 us (random init, ~55 users) to ~190 us (trained tables) for them plus a merge of
// 58 partial lists per user — a handful of users cut into pieces of 52 tiles, each piece a latency chain of 32 fp32 MFMAs per tile.
// With bounds that tight the job is a filter: score every (flagged user, item) pair once — plain fmaf chains in the order the fp32
// sweep adds its products (the re-scoring kernel's: bit-identical values), 55 x 96 k x 64 = 0.3 G — and keep the pairs that reach the
// user's bound and are neither banned nor excluded: k of them, a few more on ties.  topk_filter_kernel appends those to a list per
// user (as sortable keys), topk_filter_select_kernel ranks each list and writes the best k.  A user whose list overflows kFilterCap
// entries (a weak bound: fewer than k real candidates, or a user its wave gave up on early) or holds fewer than k goes on to the
// bounded sweep as before — through a second device-side list, usually empty.
constexpr int kFilterCap = 256;        // entries per user: 4 per lane of the selecting wave
constexpr int kFilterUsers = 32;       // users staged in LDS at a time

template <int D>
__global__ __launch_bounds__(kBlock) void topk_filter_kernel(const float *__restrict__ user_rows, int64_t ldu, const int64_t *__restrict__ user_ids,
                                                             const float *__restrict__ item_rows, int64_t ldi, int64_t n_items,
                                                             const int64_t *__restrict__ excl_rowptr, const int32_t *__restrict__ excl_col,
                                                             const uint8_t *__restrict__ banned, const int32_t *__restrict__ flagged,
                                                             const float *__restrict__ bound, const uint8_t *__restrict__ weak, int max_users,
                                                             int32_t *__restrict__ cnt, unsigned long long *__restrict__ lists)
{
    __shared__ float4 urow[kFilterUsers][D / 4];
    __shared__ int64_t uid_s[kFilterUsers];
    __shared__ float bound_s[kFilterUsers];
    const int n_f = flagged[0] < max_users ? flagged[0] : max_users;
    if ((int)blockIdx.y * kFilterUsers >= n_f) return;            // (planned for max_users: blockIdx.y = a chunk of kFilterUsers of them)
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = item < n_items;
    {
        const int u0 = (int)blockIdx.y * kFilterUsers;
        const int nu = n_f - u0 < kFilterUsers ? n_f - u0 : kFilterUsers;
        bool any_user = false;
        if ((int)threadIdx.x < nu) {
            const int64_t b = flagged[1 + u0 + threadIdx.x];
            uid_s[threadIdx.x] = user_ids ? user_ids[b] : b;
            // (no bound to speak of — fewer than k real candidates, or a user given up on early —: every item would queue up at this
            // user's counter; and a list that has overflowed by now takes no more: these users go to the bounded sweep)
            const float bd = bound[u0 + threadIdx.x];
            any_user = bd > -INFINITY && !weak[u0 + threadIdx.x] && cnt[u0 + threadIdx.x] <= kFilterCap;
            bound_s[threadIdx.x] = any_user ? bd : __builtin_nanf("");   // (nothing reaches a NaN)
        }
        if (!__syncthreads_or(any_user)) return;                   // (trained tables: every flagged user of the chunk is one its wave gave up on)
        float4 it[D / 4];
#pragma unroll
        for (int q = 0; q < D / 4; ++q) it[q] = live ? *reinterpret_cast<const float4 *>(item_rows + item * ldi + 4 * q) : f4_zero();
        const bool is_banned = live && banned && banned[item];
        for (int e = threadIdx.x; e < nu * (D / 4); e += kBlock) {
            const int uu = e / (D / 4), q = e - uu * (D / 4);
            urow[uu][q] = *reinterpret_cast<const float4 *>(user_rows + uid_s[uu] * ldu + 4 * q);
        }
        __syncthreads();
        for (int uu = 0; uu < nu; ++uu) {
            if (!(bound_s[uu] == bound_s[uu])) continue;           // (NaN: not a user for the filter; uniform over the workgroup)
            float acc = 0.f;                                       // the chain of topk_rescore_kernel: k = 8 q + c, 8 q + 4 + c
#pragma unroll
            for (int q = 0; q < D / 8; ++q) {
                const float4 ua = urow[uu][2 * q], ub = urow[uu][2 * q + 1];
                const float4 ia = it[2 * q], ib = it[2 * q + 1];
                acc = fmaf(ia.x, ua.x, acc); acc = fmaf(ib.x, ub.x, acc);
                acc = fmaf(ia.y, ua.y, acc); acc = fmaf(ib.y, ub.y, acc);
                acc = fmaf(ia.z, ua.z, acc); acc = fmaf(ib.z, ub.z, acc);
                acc = fmaf(ia.w, ua.w, acc); acc = fmaf(ib.w, ub.w, acc);
            }
            if (live && !is_banned && acc >= bound_s[uu]) {
                bool ok = true;
                if (excl_rowptr) {
                    const int64_t r0 = excl_rowptr[uid_s[uu]], r1 = excl_rowptr[uid_s[uu] + 1];
                    int64_t lo = r0, hi = r1;
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (excl_col[mid] < (int32_t)item) lo = mid + 1; else hi = mid;
                    }
                    ok = !(lo < r1 && excl_col[lo] == (int32_t)item);
                }
                if (ok) {
                    const int slot = atomicAdd(cnt + u0 + uu, 1);
                    if (slot < kFilterCap) lists[(int64_t)(u0 + uu) * kFilterCap + slot] = make_key(acc, (int)item);
                }
            }
        }
    }
}

// one wave per flagged user: the best k of its list -> the caller's rows; overflowed / short lists -> the slow list
__global__ __launch_bounds__(kBlock) void topk_filter_select_kernel(const int32_t *__restrict__ flagged, const float *__restrict__ bound, int max_users,
                                                                    const int32_t *__restrict__ cnt, const unsigned long long *__restrict__ lists, int k,
                                                                    int64_t *__restrict__ out_idx, float *__restrict__ out_val,
                                                                    int32_t *__restrict__ slow_count, int32_t *__restrict__ slow_rows,
                                                                    float *__restrict__ slow_bound)
{
    __shared__ unsigned long long keys[kBlock / kWave][kFilterCap];
    const int n_f = flagged[0] < max_users ? flagged[0] : max_users;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    const int f = blockIdx.x * (kBlock / kWave) + wave;
    if (f >= n_f) return;                                          // (a whole wave leaves; no block-wide barrier below)
    const int n = cnt[f];
    const int64_t row = flagged[1 + f];
    if (n > kFilterCap || n < k) {
        if (lane == 0) {
            const int s = atomicAdd(slow_count, 1);
            slow_rows[s] = (int32_t)row;
            slow_bound[s] = bound[f];
        }
        return;
    }
    unsigned long long mine[kFilterCap / kWave];
#pragma unroll
    for (int j = 0; j < kFilterCap / kWave; ++j) {
        const int e = lane + j * kWave;
        mine[j] = e < n ? lists[(int64_t)f * kFilterCap + e] : 0ull;
        keys[wave][e] = mine[j];
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    int rank[kFilterCap / kWave];
#pragma unroll
    for (int j = 0; j < kFilterCap / kWave; ++j) rank[j] = 0;
    for (int e = 0; e < n; ++e) {
        const unsigned long long other = keys[wave][e];            // (an LDS broadcast; keys are distinct: the item id is part of them)
#pragma unroll
        for (int j = 0; j < kFilterCap / kWave; ++j) rank[j] += other > mine[j] ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < kFilterCap / kWave; ++j) {
        if (lane + j * kWave < n && rank[j] < k) {
            out_idx[row * k + rank[j]] = key_item(mine[j]);
            out_val[row * k + rank[j]] = key_score(mine[j]);
        }
    }
}

template <int D, int NG, bool FULL, int MODE = 0, bool BOUNDED = false>
static int launch_topk(const TopkPlan &p, hipStream_t st, const TopkArgs &args)
{
    auto kern = score_topk_kernel<D, NG, FULL, MODE, BOUNDED>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(160 * 1024));
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    if (p.units >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    static int scratch_bytes = -1;                             // (per instantiation; see capture_guard)
    const int cg = capture_guard(reinterpret_cast<const void *>(kern), st, &scratch_bytes);
    if (cg != IGCN_OK) return cg;
    hipLaunchKernelGGL(kern, dim3((unsigned)p.units), dim3(kWave), p.lds_bytes, st, args);
    return launch_status();
}

// users whose lists come back in pieces (rows of the workspace)
static inline int64_t topk_rest_users(const TopkPlan &p, int64_t batch) {
    if (p.p_max <= 1) return 0;
    const int64_t first = p.n_whole * p.units * 32 * p.ng;
    return batch > first ? batch - first : 0;
}
// the partial lists (fp32 score + int32 id per entry), rounded up so that what follows stays aligned
static inline int64_t topk_merge_bytes(const TopkPlan &p, int64_t batch, int32_t k) {
    return (topk_rest_users(p, batch) * p.p_max * k * 8 + 255) / 256 * 256;
}

}  // namespace igcn

using namespace igcn;

extern "C" int64_t igcn_score_topk_workspace_bytes(int64_t batch, int64_t n_items, int32_t d, int32_t k)
{
    TopkPlan p, pb;                                                       // (the bounded sweep may plan narrower groups: the larger of the two)
    if (topk_make_plan(batch, n_items, d, k, &p) != IGCN_OK || topk_make_plan(batch, n_items, d, k, &pb, -1) != IGCN_OK) return -1;
    const int64_t mb = topk_merge_bytes(p, batch, k), mbb = topk_merge_bytes(pb, batch, k);
    return (mb > mbb ? mb : mbb) + (int64_t)p.n_tiles * 4;               // + the banned items as one word per tile
}

// One sweep: MODE 0 the exact fp32 one, MODE 1 the bf16 candidate sweep (d = 64, `packed` = the item planes).
static int topk_run(int mode, const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                    const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                    const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                    int32_t k, int64_t *out_idx, float *out_val, void *workspace, const float4 *packed,
                    const unsigned int *stats, hipStream_t st, const int32_t *perm = nullptr, const float *init_thr = nullptr,
                    const float *tile_bound = nullptr, const float *unorm2 = nullptr, unsigned int *exit_count = nullptr,
                    uint8_t *unfinished = nullptr, unsigned int *shared_thr = nullptr,
                    const int32_t *rows = nullptr, const int32_t *count_dev = nullptr, unsigned int *piece_best = nullptr,
                    float *warm_thr = nullptr)
{
    if ((rows || count_dev) && !(mode == 0 && init_thr && (d == 64 || d == 128))) return IGCN_E_SHAPE;   // the bounded variants only
    if (!user_rows || !item_rows || !out_idx || !out_val) return IGCN_E_NULL;
    if ((excl_rowptr == nullptr) != (excl_col == nullptr)) return IGCN_E_NULL;
    TopkPlan p;
    int rc = topk_make_plan(batch, n_items, d, k, &p, mode == 0 && init_thr ? -1 : mode);
    if (rc != IGCN_OK) return rc;
    if (ldu < d || ldi < d || ldu % 4 || ldi % 4 || n_items >= ((int64_t)1 << 31) - 64 || ldi > (1 << 20)) return IGCN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(user_rows) | reinterpret_cast<uintptr_t>(item_rows)) % 16) return IGCN_E_ALIGN;
    const int64_t rest_users = topk_rest_users(p, batch);
    if ((rest_users > 0 || banned) && !workspace) return IGCN_E_NULL;
    if (workspace && reinterpret_cast<uintptr_t>(workspace) % 8) return IGCN_E_ALIGN;
    float *ws_val = static_cast<float *>(workspace);
    int32_t *ws_idx = reinterpret_cast<int32_t *>(ws_val ? ws_val + rest_users * p.p_max * k : nullptr);
    uint32_t *banned_bits = nullptr;
    if (banned) {
        banned_bits = reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + topk_merge_bytes(p, batch, k));
        const int64_t pairs = ((int64_t)p.n_tiles + 1) / 2;
        hipLaunchKernelGGL(topk_pack_banned_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(kBlock), 0, st, banned, n_items,
                           p.n_tiles, perm, banned_bits);
        rc = launch_status();
        if (rc != IGCN_OK) return rc;
    }
    TopkArgs a{};
    a.user_rows = user_rows; a.ldu = ldu; a.user_ids = user_ids; a.batch = batch;
    a.item_rows = item_rows; a.ldi = ldi; a.n_items = n_items; a.d = (int)d;
    a.excl_rowptr = excl_rowptr; a.excl_col = excl_col; a.banned_bits = banned_bits;
    a.k = (int)k; a.cap = p.cap; a.n_tiles = p.n_tiles; a.p_max = p.p_max;
    const int stagger = tuning_get(IGCN_TUNE_TOPK_STAGGER);
    a.stagger = stagger < 0 ? 3 : stagger;                        // bit 0: base priority by wave slot, bit 1: boost in the slow path
    a.n_whole = p.n_whole; a.rest_tiles = p.rest_tiles; a.run = p.run;
    a.out_idx = out_idx; a.out_val = out_val; a.ws_val = ws_val; a.ws_idx = ws_idx;
    a.packed = packed;
    a.stats = stats;
    a.init_thr = init_thr;
    a.tile_bound = tile_bound; a.unorm2 = unorm2;
    a.exit_count = exit_count; a.unfinished = unfinished;
    a.shared_thr = p.p_max > 1 ? shared_thr : nullptr;          // (only pieces have anything to share)
    a.rows = rows; a.count_dev = count_dev; a.piece_best = p.p_max > 1 && p.ng == 1 ? piece_best : nullptr;
    { const int wt = tuning_get(IGCN_TUNE_TOPK_FAST_WARM); a.warm_tiles = warm_thr ? (wt < 0 ? kWarmTiles : wt) : 0; a.warm_thr = a.warm_tiles > 0 ? warm_thr : nullptr; }
    { const int ec = tuning_get(IGCN_TUNE_TOPK_FAST_EARLY_CHECKS); a.early_checks = ec < 0 ? kEarlyCheckEvery : ec / 3 * 3; }

    if (mode != 0) {
        if ((d != 64 && !(mode >= 2 && d == 128)) || !packed || (mode >= 2 && !stats)) return IGCN_E_SHAPE;
        if (d == 128 && mode == 3) rc = launch_topk<128, 2, true, 3>(p, st, a);
        else if (d == 128) rc = p.ng == 2 ? launch_topk<128, 2, true, 2>(p, st, a) : launch_topk<128, 1, true, 2>(p, st, a);
        else rc = mode == 3 ? (p.ng == 1 ? launch_topk<64, 1, true, 3>(p, st, a) : launch_topk<64, 2, true, 3>(p, st, a)) : mode == 2 ? launch_topk<64, 2, true, 2>(p, st, a) : launch_topk<64, 2, true, 1>(p, st, a);
    } else {
        switch (p.d_pad) {
        case 16: rc = d == 16 ? launch_topk<16, 2, true>(p, st, a) : launch_topk<16, 2, false>(p, st, a); break;
        case 32: rc = d == 32 ? launch_topk<32, 2, true>(p, st, a) : launch_topk<32, 2, false>(p, st, a); break;
        case 64:
            // the bound is an optimisation: the variants without it (other widths) simply do not use it
            if (d == 64 && init_thr) rc = p.ng == 1 ? launch_topk<64, 1, true, 0, true>(p, st, a) : launch_topk<64, 2, true, 0, true>(p, st, a);
            else rc = d == 64 ? launch_topk<64, 2, true>(p, st, a) : launch_topk<64, 2, false>(p, st, a);
            break;
        case 128:
            if (d == 128 && init_thr) rc = launch_topk<128, 1, true, 0, true>(p, st, a);
            else rc = d == 128 ? launch_topk<128, 1, true>(p, st, a) : launch_topk<128, 1, false>(p, st, a);
            break;
        default: rc = d == 256 ? launch_topk<256, 1, true>(p, st, a) : launch_topk<256, 1, false>(p, st, a); break;
        }
    }
    if (rc != IGCN_OK) return rc;
    if (rest_users > 0) {
        const int64_t first = p.n_whole * p.units * 32 * p.ng;
        if (p.p_max <= 4)
            hipLaunchKernelGGL(topk_merge_lanes_kernel<4>, dim3((unsigned)((rest_users + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                               ws_val, ws_idx, first, batch, p.n_tiles, p.run, p.p_max, (int)k, 32 * p.ng, out_idx, out_val, rows, count_dev);
        else if (p.p_max <= 8)
            hipLaunchKernelGGL(topk_merge_lanes_kernel<8>, dim3((unsigned)((rest_users + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                               ws_val, ws_idx, first, batch, p.n_tiles, p.run, p.p_max, (int)k, 32 * p.ng, out_idx, out_val, rows, count_dev);
        else if (p.p_max <= kWave)
            hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((rest_users + 3) / 4)), dim3(kBlock), 0, st, ws_val, ws_idx,
                               first, batch, p.n_tiles, p.run, p.p_max, (int)k, 32 * p.ng, out_idx, out_val, rows, count_dev);
        else
            hipLaunchKernelGGL(topk_merge_wide_kernel<4>, dim3((unsigned)((rest_users + 3) / 4)), dim3(kBlock), 0, st, ws_val, ws_idx,
                               first, batch, p.n_tiles, p.run, p.p_max, (int)k, 32 * p.ng, out_idx, out_val, rows, count_dev);
        rc = launch_status();
    }
    return rc;
}

extern "C" int igcn_score_topk_f32(const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                                   const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                                   const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                                   int32_t k, int64_t *out_idx, float *out_val, void *workspace, void *stream)
{
    return topk_run(0, user_rows, ldu, user_ids, batch, item_rows, ldi, n_items, d, excl_rowptr, excl_col, banned, k,
                    out_idx, out_val, workspace, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int igcn_score_topk_bounded_f32(const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                                           const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                                           const int64_t *excl_rowptr, const int32_t *excl_col, const uint8_t *banned,
                                           int32_t k, const float *lower_bound, int64_t *out_idx, float *out_val,
                                           void *workspace, void *stream)
{
    if (!lower_bound) return IGCN_E_NULL;
    return topk_run(0, user_rows, ldu, user_ids, batch, item_rows, ldi, n_items, d, excl_rowptr, excl_col, banned, k,
                    out_idx, out_val, workspace, nullptr, nullptr, static_cast<hipStream_t>(stream), nullptr, lower_bound);
}

// ---- the two-stage evaluation: bf16 candidate sweep + exact fp32 re-scoring (d = 64, k <= 60) --------------------
// workspace: [sweep workspace for k + 4][item planes][candidate ids][candidate scores][max |item|^2], each 256-aligned
static inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }
// Users the two-stage call finishes by itself: the first kFastFallbackMax flagged ones go through the bounded fp32 sweep inside
// igcn_score_topk_fast_f32, planned for that many and run for as many as the device-side count says (ABI v7).  256 = 8 narrow
// groups x 58 pieces fill a quarter of the wave slots — the plan a host that knew the count (43 at random init, 20-40 on
// trained tables) would make for anything up to 256 users; beyond it the caller re-does the rest.
constexpr int64_t kFastFallbackMax = IGCN_FAST_FALLBACK_MAX;
struct FastLayout { int64_t sweep, packed, cand_idx, cand_val, norm, filter, filter_lists, tile_bound, unorm2, warm, exit_state, order, fallback, total; int kc; TopkOrderLayout ord; };
constexpr int64_t kFilterState = kFastFallbackMax * 4 + 128 + 128 + kFastFallbackMax;     // [entries per flagged user][users left to the bounded sweep][weak-bound marks]: zero on entry
static int topk_fast_layout(int64_t batch, int64_t n_items, int32_t d, int32_t k, int64_t excl_rows, int64_t excl_nnz, FastLayout *L) {
    if ((d != 64 && d != 128) || k < 1 || k + kFastExtra > kWave) return IGCN_E_RANGE;
    L->kc = k + topk_fast_extra(k, topk_fast_mode(d));
    TopkPlan p;
    const int64_t kc = n_items < L->kc ? n_items : L->kc;        // never more candidates than items
    L->kc = (int)kc;
    int rc = topk_make_plan(batch, n_items, d, L->kc, &p, topk_fast_mode(d));
    if (rc != IGCN_OK) return rc;
    rc = topk_order_layout(n_items, excl_rows, excl_nnz, &L->ord);
    if (rc != IGCN_OK) return rc;
    L->sweep = 0;
    L->packed = align256(topk_merge_bytes(p, batch, L->kc) + (int64_t)p.n_tiles * 4);
    L->cand_idx = L->packed + (int64_t)p.n_tiles * 8 * kWave * 16;
    L->cand_val = L->cand_idx + align256(batch * L->kc * 8);
    // the filter's lists (kFilterCap sortable keys per flagged user), then the rows and bounds of the users it leaves to the bounded sweep
    L->filter_lists = L->cand_val + align256(batch * L->kc * 4);
    L->tile_bound = L->filter_lists + kFastFallbackMax * kFilterCap * 8 + 2 * align256(kFastFallbackMax * 4);
    L->unorm2 = L->tile_bound + align256((int64_t)p.n_tiles * 4);
    L->warm = L->unorm2 + align256(batch * 4);                  // the warm-up pass's bounds, one float per user
    L->norm = L->warm + align256(batch * 4);                    // (the table maxima sit right before the exit state: one memset clears both)
    L->filter = L->norm + 256;                                  // (cleared with them)
    L->exit_state = L->filter + kFilterState;
    // [256 B: early leavers per job][the fall-back's shared thresholds][batch B: users given up on][batch x 4 B: the sweep's shared thresholds]
    L->order = L->exit_state + 256 + kFastFallbackMax * 4 + kFastFallbackMax * kWave * 4 + align256(batch) + align256(batch * 4);   // (+ the fall-back's piece_best: k <= 64 slots per user)
    L->fallback = align256(L->order + L->ord.total);
    const int64_t fb = igcn_score_topk_workspace_bytes(batch < kFastFallbackMax ? batch : kFastFallbackMax, n_items, d, k);
    if (fb < 0) return IGCN_E_RANGE;
    L->total = L->fallback + align256(fb);
    return IGCN_OK;
}

extern "C" int64_t igcn_score_topk_fast_workspace_bytes(int64_t batch, int64_t n_items, int32_t d, int32_t k,
                                                        int64_t excl_rows, int64_t excl_nnz)
{
    FastLayout L;
    return topk_fast_layout(batch, n_items, d, k, excl_rows, excl_nnz, &L) == IGCN_OK ? L.total : -1;
}

// Flagged users the next igcn_score_topk_fast_f32 call of this shape finishes itself (ABI v8): the caller's share of the flagged list
// starts behind them.  The library is the one place that knows (the limit, and the "topk_fast_fallback" knob that turns it off).
extern "C" int64_t igcn_score_topk_fast_finished_max(int64_t batch, int32_t with_lower_bound)
{
    if (batch < 0) return -1;
    if (!with_lower_bound || tuning_get(IGCN_TUNE_TOPK_FAST_FALLBACK) == 0) return 0;
    return batch < kFastFallbackMax ? batch : kFastFallbackMax;
}

extern "C" int igcn_score_topk_fast_f32(const float *user_rows, int64_t ldu, const int64_t *user_ids, int64_t batch,
                                        const float *item_rows, int64_t ldi, int64_t n_items, int32_t d,
                                        const int64_t *excl_rowptr, const int32_t *excl_col, int64_t excl_rows, int64_t excl_nnz,
                                        const uint8_t *banned, int32_t k, int64_t *out_idx, float *out_val,
                                        int32_t *flagged, float *flagged_lower_bound, void *workspace, void *stream)
{
    if (!workspace || !flagged) return IGCN_E_NULL;
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return IGCN_E_ALIGN;
    if (k > n_items) return IGCN_E_RANGE;
    if ((excl_rowptr == nullptr) != (excl_col == nullptr)) return IGCN_E_NULL;
    if (!excl_rowptr) excl_rows = excl_nnz = 0;
    if (excl_rowptr && (excl_rows < 1 || excl_nnz < 1)) return IGCN_E_SHAPE;
    FastLayout L;
    int rc = topk_fast_layout(batch, n_items, d, k, excl_rows, excl_nnz, &L);
    if (rc != IGCN_OK) return rc;
    if (!item_rows || ldi < d || ldi % 4 || reinterpret_cast<uintptr_t>(item_rows) % 16) return IGCN_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    float4 *packed = reinterpret_cast<float4 *>(ws + L.packed);
    int64_t *cand_idx = reinterpret_cast<int64_t *>(ws + L.cand_idx);
    float *cand_val = reinterpret_cast<float *>(ws + L.cand_val);
    unsigned int *norm_bits = reinterpret_cast<unsigned int *>(ws + L.norm);
    char *ows = ws + L.order;
    // 2: one fp16 item plane (default; the only one at d = 128), 1: two bf16 planes
    const int mode = topk_fast_mode(d);
    const int ks = d / 16, lg = d == 128 ? 5 : 4;
    const bool by_norm = tuning_get(IGCN_TUNE_TOPK_FAST_ORDER) != 0;       // developer knob: 0 = sweep in id order
    // one memset: the table maxima, and behind them the exit counters, the fall-back's shared state, the give-up marks and the
    // sweep's shared thresholds (the last two are used by some plans only; clearing 0.6 MB costs what clearing 16 bytes costs)
    const int64_t fb_state = 256 + kFastFallbackMax * 4 + kFastFallbackMax * kWave * 4;
    // ... and, right behind them, the counters of the order build (the first kOrderBinsBytes of its workspace)
    static_assert(kOrderBinsBytes % 256 == 0, "");
    if (L.order != L.exit_state + fb_state + align256(batch) + align256(batch * 4) || L.ord.bins != 0) return IGCN_E_RANGE;   // (the layout this memset relies on)
    // (the library's own zeroing kernel, not hipMemsetAsync: a memset NODE of a captured graph is not ordered against the kernels behind
    // it on ROCm 7.2 — see zero_async in common.h)
    // (test-only knob "topk_fast_poison": the bins are left as the caller's workspace held them — the order build has to notice)
    const bool clear_bins = by_norm && tuning_get(IGCN_TUNE_TOPK_FAST_POISON) <= 0;
    rc = zero_async(norm_bits, (size_t)(256 + kFilterState + fb_state + align256(batch) + align256(batch * 4) + (clear_bins ? kOrderBinsBytes : 0)), st);
    if (rc != IGCN_OK) return rc;
    const int n_tiles = (int)((n_items + 31) / 32);
    const int64_t pack_threads = (int64_t)n_tiles * ks * kWave;
    int64_t stat_blocks = ((n_items << lg) + kBlock - 1) / kBlock;
    // (measured, round 4: eight workgroups per CU took 47 us against 23 for one here — two contended words and the norm table to
    // write; the users' pass below, one word and no table, takes 16 against 33 with eight)
    // (in the fused launch, late round 4: two per CU 22 us, one 25-31, four 41)
    if (stat_blocks > 2 * (int64_t)cu_count()) stat_blocks = 2 * (int64_t)cu_count();
    const int32_t *perm = nullptr, *excl_pos = nullptr;
    float *tile_bound = reinterpret_cast<float *>(ws + L.tile_bound), *unorm2 = reinterpret_cast<float *>(ws + L.unorm2);
    const bool early_exit = mode >= 2 && by_norm && tuning_get(IGCN_TUNE_TOPK_FAST_EXIT) != 0;     // developer knob: 0 = always sweep to the end
    int64_t ub = 0;
    if (mode >= 2) {
        if (!user_rows || ldu < d || ldu % 4 || reinterpret_cast<uintptr_t>(user_rows) % 16) return IGCN_E_SHAPE;
        ub = ((batch << lg) + kBlock - 1) / kBlock;
        if (ub > 8 * (int64_t)cu_count()) ub = 8 * (int64_t)cu_count();
    }
    hipLaunchKernelGGL(topk_row_stats_both_kernel, dim3((unsigned)(stat_blocks + ub)), dim3(kBlock), 0, st, item_rows, ldi, n_items,
                       by_norm ? reinterpret_cast<float *>(ows + L.ord.norm2) : (float *)nullptr, (unsigned)stat_blocks,
                       user_rows, ldu, batch, user_ids, early_exit ? unorm2 : (float *)nullptr, lg, norm_bits, flagged);
    if (by_norm) {
        rc = topk_order_build(L.ord, ows, n_items, excl_rowptr, excl_col, excl_rows, excl_nnz, user_ids, batch, st, &perm, &excl_pos);
        if (rc != IGCN_OK) return rc;
    }
    if (mode >= 2) {
        hipLaunchKernelGGL(topk_pack_items_f16_kernel, dim3((unsigned)((pack_threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           item_rows, ldi, n_items, n_tiles, ks, norm_bits, perm, packed,
                           early_exit ? reinterpret_cast<const float *>(ows + L.ord.norm2) : (const float *)nullptr,
                           early_exit ? tile_bound : (float *)nullptr,
                           by_norm ? reinterpret_cast<const uint32_t *>(ows + L.ord.status) : (const uint32_t *)nullptr);
    } else {
        hipLaunchKernelGGL(topk_pack_items_kernel, dim3((unsigned)((pack_threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           item_rows, ldi, n_items, n_tiles, perm, packed);
    }
    rc = launch_status();
    if (rc != IGCN_OK) return rc;
    // waves that outlast three quarters of the others hand their remaining users to the fp32 sweep (developer knob
    // "topk_fast_give_up" 0: every wave runs until it can leave or its sweep ends)
    // — only where a wave sweeps the whole table: the pieces a small batch is cut into end within a few hundred tiles anyway, and
    // handing users over costs ~0.3 ms (Gowalla-size trained tables: 1.1 ms with, 0.8 ms without)
    TopkPlan sweep_plan;
    rc = topk_make_plan(batch, n_items, d, L.kc, &sweep_plan, mode);
    if (rc != IGCN_OK) return rc;
    // (whole sweeps only — `n_whole` of them per wave, counted job by job; the pieces of a rest never give up)
    const bool give_up = early_exit && sweep_plan.n_whole >= 1 && tuning_get(IGCN_TUNE_TOPK_FAST_GIVE_UP) != 0;
    unsigned int *exit_count = reinterpret_cast<unsigned int *>(ws + L.exit_state);
    unsigned int *fb_shared_thr = reinterpret_cast<unsigned int *>(ws + L.exit_state + 256);
    unsigned int *fb_piece_best = reinterpret_cast<unsigned int *>(ws + L.exit_state + 256 + kFastFallbackMax * 4);
    uint8_t *unfinished = reinterpret_cast<uint8_t *>(ws + L.exit_state + fb_state);
    unsigned int *shared_thr = reinterpret_cast<unsigned int *>(ws + L.exit_state + fb_state + align256(batch));
    const bool share = sweep_plan.p_max > 1 && mode == 3 && d == 64 && tuning_get(IGCN_TUNE_TOPK_FAST_SHARE) != 0;
    // the sweep runs in position space: its exclusion lists and banned bits are those of the positions, and the
    // candidate ids it returns are positions (mapped back by the re-scoring kernel)
    uint8_t *flagged_weak = reinterpret_cast<uint8_t *>(ws + L.filter + kFastFallbackMax * 4 + 128);   // (in the cleared filter state, behind the count of users left over)
    rc = topk_run(mode, user_rows, ldu, user_ids, batch, item_rows, ldi, n_items, d, excl_rowptr, by_norm && excl_rowptr ? excl_pos : excl_col,
                  banned, L.kc, cand_idx, cand_val, ws + L.sweep, packed, norm_bits, st, perm, nullptr,
                  early_exit ? tile_bound : nullptr, early_exit ? unorm2 : nullptr, give_up ? exit_count : nullptr,
                  give_up ? unfinished : nullptr, share ? shared_thr : nullptr, nullptr, nullptr, nullptr,
                  mode == 3 && d == 64 ? reinterpret_cast<float *>(ws + L.warm) : nullptr);
    if (rc != IGCN_OK) return rc;
    if (L.kc <= 32)
        hipLaunchKernelGGL(topk_rescore_kernel<32>, dim3((unsigned)((batch + 7) / 8)), dim3(kBlock), 0, st, user_rows, ldu, user_ids, batch,
                           item_rows, ldi, cand_idx, cand_val, L.kc, (int)k, (int)d, norm_bits, mode, perm, out_idx, out_val, flagged,
                           flagged_lower_bound, give_up ? (const uint8_t *)unfinished : (const uint8_t *)nullptr, flagged_weak);
    else
        hipLaunchKernelGGL(topk_rescore_kernel<64>, dim3((unsigned)((batch + 3) / 4)), dim3(kBlock), 0, st, user_rows, ldu, user_ids, batch,
                           item_rows, ldi, cand_idx, cand_val, L.kc, (int)k, (int)d, norm_bits, mode, perm, out_idx, out_val, flagged,
                           flagged_lower_bound, give_up ? (const uint8_t *)unfinished : (const uint8_t *)nullptr, flagged_weak);
    rc = launch_status();
    if (rc != IGCN_OK || !flagged_lower_bound || tuning_get(IGCN_TUNE_TOPK_FAST_FALLBACK) == 0) return rc;
    // The flagged users' fp32 sweep, planned HERE for up to kFastFallbackMax of them and run for as many as flagged[0] says when
    // the kernel starts — no host read between the stages (round 3 read the count back first: ~0.11 ms of an idle GPU and five
    // small torch launches per call).  Each starts from the k-th exact score of its candidates.  In id space: the caller's
    // exclusion lists and banned items as they came.
    const int64_t fb_users = batch < kFastFallbackMax ? batch : kFastFallbackMax;
    const int32_t *slow_rows = flagged + 1, *slow_count = flagged;
    const float *slow_bound = flagged_lower_bound;
    if (tuning_get(IGCN_TUNE_TOPK_FAST_FILTER) != 0) {
        // the streaming filter first (see topk_filter_kernel); what it cannot finish arrives at the bounded sweep through a list of its own
        int32_t *cnt = reinterpret_cast<int32_t *>(ws + L.filter), *left = reinterpret_cast<int32_t *>(ws + L.filter + kFastFallbackMax * 4);
        unsigned long long *lists = reinterpret_cast<unsigned long long *>(ws + L.filter_lists);
        int32_t *left_rows = reinterpret_cast<int32_t *>(ws + L.filter_lists + kFastFallbackMax * kFilterCap * 8);
        float *left_bound = reinterpret_cast<float *>(ws + L.filter_lists + kFastFallbackMax * kFilterCap * 8 + align256(kFastFallbackMax * 4));
        const dim3 fgrid((unsigned)((n_items + kBlock - 1) / kBlock), (unsigned)((fb_users + kFilterUsers - 1) / kFilterUsers));
        if (d == 64)
            hipLaunchKernelGGL(topk_filter_kernel<64>, fgrid, dim3(kBlock), 0, st, user_rows, ldu, user_ids, item_rows, ldi, n_items, excl_rowptr,
                               excl_col, banned, (const int32_t *)flagged, (const float *)flagged_lower_bound, (const uint8_t *)flagged_weak, (int)fb_users, cnt, lists);
        else
            hipLaunchKernelGGL(topk_filter_kernel<128>, fgrid, dim3(kBlock), 0, st, user_rows, ldu, user_ids, item_rows, ldi, n_items, excl_rowptr,
                               excl_col, banned, (const int32_t *)flagged, (const float *)flagged_lower_bound, (const uint8_t *)flagged_weak, (int)fb_users, cnt, lists);
        hipLaunchKernelGGL(topk_filter_select_kernel, dim3((unsigned)((fb_users + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0, st,
                           (const int32_t *)flagged, (const float *)flagged_lower_bound, (int)fb_users, (const int32_t *)cnt,
                           (const unsigned long long *)lists, (int)k, out_idx, out_val, left, left_rows, left_bound);
        rc = launch_status();
        if (rc != IGCN_OK) return rc;
        slow_rows = left_rows; slow_count = left; slow_bound = left_bound;
    }
    return topk_run(0, user_rows, ldu, user_ids, fb_users, item_rows, ldi, n_items, d, excl_rowptr, excl_col, banned, k, out_idx, out_val,
                    ws + L.fallback, nullptr, nullptr, st, nullptr, slow_bound, nullptr, nullptr, nullptr, nullptr,
                    tuning_get(IGCN_TUNE_TOPK_FAST_SHARE) != 0 ? fb_shared_thr : nullptr, slow_rows, slow_count,
                    tuning_get(IGCN_TUNE_TOPK_FAST_SHARE) != 0 ? fb_piece_best : nullptr);
}

#ifdef IGCN_TOPK_STATS
extern "C" int igcn_debug_topk_wave_times(unsigned long long *host, int n_waves)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(host, HIP_SYMBOL(igcn::g_topk_wave_times), (size_t)n_waves * 24);
    return (int)e;
}
extern "C" int igcn_debug_topk_stats(unsigned long long *host8, int reset)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && host8) e = hipMemcpyFromSymbol(host8, HIP_SYMBOL(igcn::g_topk_stats), 128);
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(igcn::g_topk_stats), z, 128);
    }
    return (int)e;
}
#endif

extern "C" int64_t igcn_eval_metrics_workspace_bytes(int64_t n_users)
{
    if (n_users < 0) return -1;
    return ((n_users + kBlock - 1) / kBlock + 1) * (3 * IGCN_MAX_METRIC_CUTS + 1) * 8;
}

extern "C" int igcn_eval_metrics(const int64_t *rec, int64_t n_users, int32_t k_rec, const int64_t *eval_rowptr, const int32_t *eval_col,
                                 const int32_t *topks_host, int32_t n_topks, double *out, void *workspace, void *stream)
{
    if (!rec || !eval_rowptr || !topks_host || !out || !workspace) return IGCN_E_NULL;
    if (n_users < 1 || k_rec < 1 || n_topks < 1 || n_topks > IGCN_MAX_METRIC_CUTS) return IGCN_E_SHAPE;
    MetricCuts cuts{};
    cuts.n = n_topks;
    for (int t = 0; t < n_topks; ++t) {
        if (topks_host[t] < 1 || topks_host[t] > k_rec) return IGCN_E_RANGE;
        cuts.k[t] = topks_host[t];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_blocks = (n_users + kBlock - 1) / kBlock;
    if (n_blocks >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    double *partial = static_cast<double *>(workspace);
    hipLaunchKernelGGL(eval_metrics_kernel, dim3((unsigned)n_blocks), dim3(kBlock), 0, st, rec, n_users, (int)k_rec, eval_rowptr, eval_col, cuts, partial);
    hipLaunchKernelGGL(eval_metrics_finish_kernel, dim3(1), dim3(kWave), 0, st, (const double *)partial, n_blocks, out);
    return launch_status();
}

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
