// CSR SpMM for embedding propagation on MI355X (gfx950):  Y = epilogue(M @ X).
//
// Replaces dgl.ops.gspmm(g,'mul','sum',X,w) of the reference (model.py:102,
// :430, :442) together with the per-call graph rebuild (model.py:99-100), the
// layer mean (model.py:104-105), the sparse dropout (model.py:263-275 via :435)
// and the row-constant feature values (model.py:374-377).
//
// Mapping to the hardware (HBM / Infinity-Cache bound, no MFMA: there is no
// dense contraction here):
//   * one 64-lane wave owns one output row at a time; the wave is cut into
//     G = 64/LPR lane groups, LPR = d/4 lanes each, and every lane moves one
//     float4, so ONE gather instruction fetches G whole source rows as G fully
//     coalesced d*4-byte requests (d=64: 4 rows x 256 B = 1 KiB per instruction,
//     the widest access the memory pipeline has);
//   * col/val of up to 64 nonzeros are read with one coalesced load per array
//     and handed to the groups through the cross-lane network (ds_bpermute), so
//     the index stream is read once, as whole lines;
//   * 4 gather instructions (4*G source rows) are kept in flight per wave before
//     the first FMA; with 8 waves per SIMD that is >64 KiB in flight per CU;
//   * rows longer than `long_threshold` nonzeros (power-law item rows) are not
//     processed as rows: the host plan cuts them into row segments that are
//     scheduled like ordinary rows ("virtual rows" after the real ones), write
//     partial sums, and a second tiny kernel adds the partials in a fixed order
//     — results are bitwise reproducible, no float atomics;
//   * rows are dealt round-robin to the waves of the grid, so neighbouring waves
//     stream neighbouring col/val lines.  The grid is sized for about 21 KB of
//     gathered cache lines per wave (4 rows per wave on the Amazon-like graph at
//     d = 64, one at d = 128): enough workgroups that the dispatcher evens out a
//     power-law load, few enough to amortise the per-wave set-up — and never just
//     above what is resident at once (see resident_blocks_per_cu below).
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace igcn {

struct SpmmEpilogue {
    const float *add[IGCN_MAX_ADDS];
    int n_adds;
    float out_scale;
    float add_scale;
    const float *row_scale;
    const float *col_scale;
    const uint32_t *col_mask;  // one bit per source row (26 KB for 206 k rows: stays in L1): rows known to be zero are not gathered
};

struct SpmmDropout {
    const int32_t *edge_id;
    const uint32_t *seed_words;   // the seed in device memory (low word, high word) instead of s0 / s1: a launch captured
                                  // in a HIP graph then drops different edges at every replay
    uint32_t s0, s1;
    uint32_t keep_below;   // keep iff hash < keep_below
    float keep_prob;
};

#ifdef IGCN_SPMM_TRACE
// Developer build only (scripts/dev_spmm_trace.py): [begin, end] of every wave in s_memrealtime ticks (100 MHz)
__device__ unsigned long long g_spmm_wave_times[6 * 131072];   // begin, end, HW_ID, rows, nonzeros, 64-entry chunks
__device__ __forceinline__ unsigned long long spmm_realtime() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#endif

// The part of the dealing order a wave walks: entries first, first + stride, ... < end, `per_wave` consecutive entries
// at a time.  Without xcd_off: the whole order, waves of the grid interleaved.  With xcd_off (int64 [9], device): the
// order is 8 lists back to back, list x = [xcd_off[x], xcd_off[x + 1]), and a workgroup walks list blockIdx.x % 8 only,
// interleaved with the other workgroups of that residue.  Workgroups b and b + 8 share an XCD under the hardware's
// round-robin placement, so the work of one list goes through ONE 4 MiB L2: the host puts into a list the rows and row
// segments that gather from the same slice of the operand (graph.py: xcd plan).  Placement is a speed assumption only —
// every entry is computed exactly once wherever its workgroup runs.
struct DealRange { int64_t first, end, stride; };
__device__ __forceinline__ DealRange deal_range(const int64_t *__restrict__ xcd_off, int64_t n_virtual, int per_wave)
{
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    DealRange r;
    if (xcd_off) {
        const int x = blockIdx.x & 7;
        const int64_t blocks_x = ((int64_t)gridDim.x - x + 7) >> 3;          // workgroups with this residue
        const int64_t wave_x = (int64_t)(blockIdx.x >> 3) * (kBlock / kWave) + wave_in_block;
        r.first = xcd_off[x] + wave_x * per_wave;
        r.end = xcd_off[x + 1];
        r.stride = blocks_x * (kBlock / kWave) * per_wave;
    } else {
        r.first = ((int64_t)blockIdx.x * (kBlock / kWave) + wave_in_block) * per_wave;
        r.end = n_virtual;
        r.stride = (int64_t)gridDim.x * (kBlock / kWave) * per_wave;
    }
    return r;
}

// The entries a wave visits.  SKIP = false: every entry of its deal range, `first`, `first + stride`, ...  SKIP = true (launches
// whose row mask leaves most rows out and that need not zero them — the last forward launch of a training step computes the ~6 000
// batch rows of 206 151): a masked-out entry used to cost the wave a dependent chain — row_order[vv] -> row_mask / rowptr -> nothing —
// of a few hundred ns, 59 times per wave: 8 of that launch's 38 us (the rest is the wanted rows' own nonzeros).  `bits` holds the mask in DEALING order (bit vv = the row of entry
// vv is wanted; igcn_pack_mask_bits_ordered), so a wave reads the bits of its next 64 visits in one load (lane j those of visit j:
// the PER entries vv .. vv + PER - 1, which may straddle two words — the array is padded), ballots, and walks the set bits only.
// A set bit whose row turns out masked costs one visit (the body checks row_mask as before); a missing bit would lose a row.
template <bool SKIP, int PER>
struct DealCursor {
    DealRange r;
    const uint32_t *bits;
    int lane;
    int64_t batch;
    unsigned long long todo;
    __device__ __forceinline__ int64_t begin()
    {
        if constexpr (!SKIP) return r.first;
        batch = r.first - 64 * r.stride;
        todo = 0ull;
        return next();
    }
    __device__ __forceinline__ int64_t after(int64_t vb)
    {
        if constexpr (!SKIP) return vb + r.stride;
        return next();
    }
    __device__ __forceinline__ int64_t next()
    {
        while (!todo) {
            batch += 64 * r.stride;
            if (batch >= r.end) return r.end;
            const int64_t mine = batch + lane * r.stride;
            bool some = false;
            if (mine < r.end) {
                const int64_t wi = mine >> 5;
                unsigned long long win = bits[wi];
                if (PER > 1) win |= (unsigned long long)bits[wi + 1] << 32;
                some = ((win >> (mine & 31)) & ((1ull << PER) - 1ull)) != 0ull;
            }
            todo = __ballot(some);
        }
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        return batch + k * r.stride;
    }
};

// Epilogue of one finished output row: r = (out_scale * acc + add_scale * sum(adds[dst])) * row_scale[dst] -> y[dst].
__device__ __forceinline__ void finish_row(const float4 &acc, int64_t dst, int t, const SpmmEpilogue &ep, float *__restrict__ y, int64_t ldy)
{
    float4 r = make_float4(acc.x * ep.out_scale, acc.y * ep.out_scale, acc.z * ep.out_scale, acc.w * ep.out_scale);
    if (ep.n_adds > 0) {
        float4 s = f4_zero();
        for (int i = 0; i < ep.n_adds; ++i)
            f4_add(s, *reinterpret_cast<const float4 *>(ep.add[i] + dst * ldy + 4 * t));
        f4_fma(r, ep.add_scale, s);
    }
    if (ep.row_scale) {
        const float rs = ep.row_scale[dst];
        r.x *= rs; r.y *= rs; r.z *= rs; r.w *= rs;
    }
    *reinterpret_cast<float4 *>(y + dst * ldy + 4 * t) = r;
}

// ---- cut rows folded inside the launch (round 5) -------------------------------------------------------------------------------
// A row cut into segments used to be finished by a second kernel (spmm_long_rows_reduce_kernel: 5 us, launch-bound — 15 of the
// 337 us of the headline pass, 29 us of a training step).  First attempt: the wave that delivers a row's LAST partial sum adds the
// row up, found by an arrival atomic WITH RETURN per segment — bit-equal, and 18 % slower with the XCD plan's 35 141 segments: every
// segment wave sat on a memory round trip (profiles/r05c_spmm_fold_in_launch_negative.jsonl: the wait for the stores costs 0.8 us
// per launch, the atomics with return 25).  What is built instead — nobody waits for an atomic (and it is still an opt-in knob:
// +5 % on the headline graph, see launch_rows):
//   * a segment's partial sum is stored with AGENT scope (sc1: written through to the memory side — the row's other segments ran on
//     other XCDs, whose L2s are private), the wave waits for its stores (vmcnt(0)) and counts itself in with a fire-and-forget
//     agent-scope atomic on the row's arrival counter (behind the partial sums, 128 bytes apart: see kCounterStride);
//   * ONE segment of every cut row is its CLOSING segment (bit 31 of igcn_row_segment.long_index; the plan's choice: the row's
//     last).  The dealing order hands it out well after the row's other segments (a quarter into the rows of the same phase:
//     graph.CLOSING_AT), so that when its wave has stored its own partial sum the counter already reads n_slots - 1; it polls until
//     it does (agent-scope loads, s_sleep in between), loads the row's n_slots partial sums with agent scope (never from its own
//     L2), adds them in the fixed four-chain order of sum_partials_four_chains — the reduce kernel's order: the bits do not
//     depend on the grid, on the kernel variant or on who polls —, applies the epilogue and puts the counter back to zero for the
//     next launch.
// Forward progress: a closing segment waits only for segments that come EARLIER in the dealing order of their lists; workgroups are
// dispatched in order, so those are resident or done — waves that are resident always finish.  The poll is bounded all the same
// (kClosePolls ~ seconds): a row whose count never completes — a broken plan — comes back as NaN instead of hanging the GPU.
constexpr int kClosePolls = 1 << 22;
constexpr int kClosePoison = -(1 << 30);   // what a timed-out row's arrival counter is left at (segment_done)
constexpr int kClosingBit = (int)0x80000000u;
// (one 16-byte store with the agent-scope bit, written by hand: four __hip_atomic_store of a float each cost the kernel four
// registers and a wave per SIMD — 67 instead of 63 VGPRs at d = 64)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_partial_agent(float *p, const float4 &v)
{
#ifdef IGCN_X_PLAINSTORE
    *reinterpret_cast<float4 *>(p) = v; return;
#endif
    const f32x4_t q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ float4 load_partial_agent(const float *p)
{
    float4 v;
    v.x = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
// The n_slots partial sums of a cut row, added in a FIXED order that does not depend on who adds: four chains per column — chain c
// adds slots c, c + 4, c + 8, ... in that order — and total = (chain 0 + chain 1) + (chain 2 + chain 3).  (Round 5 first summed all
// slots in one chain, one lane group at work: the reduce kernel's longest row — a popular template of INMO's transposed feature
// matrix, dozens of slots — then took 59 us instead of 16 and an IGCN training step 0.686 instead of 0.637 ms.)  G lane groups of
// LPR lanes take part (G >= 4: groups 0..3 a chain each; 2: two chains each; 1: all four), IF loads in flight per chain; `base`
// already points at this lane's four columns of slot 0, `on` = the lane holds real columns.  Every taking-part lane gets the total.
template <int G, int LPR, int IF>
__device__ __forceinline__ float4 sum_partials_four_chains(const float *base, int n_slots, int d, int g, bool on, bool agent)
{
    constexpr int CPG = G >= 4 ? 1 : 4 / G;                      // chains per group
    float4 ch[CPG];
#pragma unroll
    for (int k = 0; k < CPG; ++k) {
        ch[k] = f4_zero();
        const int c = g + k * G;                                 // (G >= 4: groups 4.. hold no chain)
        if (c < 4 && on) {
            int s = c;
            for (; s + 4 * (IF - 1) < n_slots; s += 4 * IF) {
                float4 p[IF];
#pragma unroll
                for (int i = 0; i < IF; ++i)
                    p[i] = agent ? load_partial_agent(base + (int64_t)(s + 4 * i) * d) : *reinterpret_cast<const float4 *>(base + (int64_t)(s + 4 * i) * d);
#pragma unroll
                for (int i = 0; i < IF; ++i) f4_add(ch[k], p[i]);
            }
            for (; s < n_slots; s += 4)
                f4_add(ch[k], agent ? load_partial_agent(base + (int64_t)s * d) : *reinterpret_cast<const float4 *>(base + (int64_t)s * d));
        }
    }
    if constexpr (G >= 4) {
        float4 t = ch[0];
        f4_add(t, f4_shfl_xor(ch[0], LPR));                      // groups 0, 1: chain 0 + chain 1; groups 2, 3: chain 2 + chain 3 (a + b == b + a)
        float4 r = t;
        f4_add(r, f4_shfl_xor(t, 2 * LPR));
        return r;
    } else if constexpr (G == 2) {
        float4 t01 = ch[0], t23 = ch[1];
        f4_add(t01, f4_shfl_xor(ch[0], LPR));                    // group 0 holds chains 0, 2; group 1 chains 1, 3
        f4_add(t23, f4_shfl_xor(ch[1], LPR));
        f4_add(t01, t23);
        return t01;
    } else {
        float4 t01 = ch[0], t23 = ch[2];
        f4_add(t01, ch[1]);
        f4_add(t23, ch[3]);
        f4_add(t01, t23);
        return t01;
    }
}

// The arrival counters: one per cut row, kCounterStride bytes apart, BEHIND the partial sums in the caller's `partial` buffer
// (zero when handed over, zero again after every launch).  Not the 16-byte igcn_long_row entries: eight counters to a 128-byte line
// took ~70 agent-scope atomics per line, and those are serialised where they execute (measured: +11 % per launch against +8 %).
constexpr int kCounterStride = 128;
template <int G, int LPR>
__device__ __forceinline__ void segment_done(const igcn_long_row *long_rows, int li, bool leader, int g, int t, int d,
                                             float *partial, int64_t n_segments, const SpmmEpilogue &ep, float *__restrict__ y, int64_t ldy)
{
    const bool on = 4 * t < d, writer = g == 0 && on;
    const int row_i = li & ~kClosingBit;
    int *arrived = reinterpret_cast<int *>(reinterpret_cast<char *>(partial + n_segments * d) + (int64_t)row_i * kCounterStride);
    // (developer ablations, never shipped — wrong results, only timed: IGCN_X_FOLD_NOCOUNT / _NOPOLL / _NOCLOSE)
    if (!(li & kClosingBit)) {
#ifndef IGCN_X_FOLD_NOCOUNT
        if (leader) (void)__hip_atomic_fetch_add(arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (no return: nobody waits)
#endif
        return;
    }
#ifdef IGCN_X_FOLD_NOCLOSE
    return;
#endif
    const igcn_long_row *lr = long_rows + row_i;
    const int n_slots = lr->n_slots;
    bool complete = true;
#ifndef IGCN_X_FOLD_NOPOLL
    for (int polls = 0; __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != n_slots - 1; ++polls) {
        if (polls >= kClosePolls) { complete = false; break; }
        __builtin_amdgcn_s_sleep(8);
    }
#endif
    {
        constexpr int kInFlight = G >= 2 ? 2 : 1;                                   // (a cold path of the hot kernel: few registers before speed)
#ifdef IGCN_X_FOLD_PLAINLOAD
        float4 acc = sum_partials_four_chains<G, LPR, kInFlight>(partial + (int64_t)lr->first_slot * d + 4 * t, n_slots, d, g, on, false);
#else
        float4 acc = sum_partials_four_chains<G, LPR, kInFlight>(partial + (int64_t)lr->first_slot * d + 4 * t, n_slots, d, g, on, true);
#endif
        if (!complete) { const float nan = __int_as_float(0x7fc00000); acc = make_float4(nan, nan, nan, nan); }
#ifndef IGCN_X_FOLD_NOFINISH
        if (writer) finish_row(acc, lr->row, t, ep, y, ldy);
#else
        if (writer && acc.x == 12345.f) finish_row(acc, lr->row, t, ep, y, ldy);
#endif
    }
    // A row that timed out keeps a POISONED counter (late arrivals only add to it): every later launch on this matrix times out on
    // the row again and writes NaN again, instead of adding up a mixture of two launches' segments behind a counter that was put
    // back to zero too early.  (A plan whose closing segments sit next to their siblings is not folded at all: graph.py.)
    if (leader) __hip_atomic_store(arrived, complete ? 0 : kClosePoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 80 scalar registers at most: a wave is charged its SGPRs + 16 (rounded up to 16) out of 800 per SIMD, so 80 is
// the most that still lets 8 waves share a SIMD; without the cap the d = 32 variant took 100 and ran 6 (-15 %).
// FOLD: the cut rows are added up inside the launch (opt-in, see segment_done); an instantiation of its own, so that the default
// kernels carry nothing of it (its mere presence cost the headline launch 0.4 %).
template <int LPR, bool DROPOUT, bool FOLD = false, bool SKIP = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void spmm_csr_rows_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
    int64_t n_rows, int d, SpmmEpilogue ep, SpmmDropout dr,
    const igcn_row_segment *__restrict__ segments, int64_t n_segments,
    float *__restrict__ partial, int long_threshold,
    const uint8_t *__restrict__ row_mask, int masked_rows_zero, const int32_t *__restrict__ row_order,
    const int64_t *__restrict__ xcd_off, const igcn_long_row *__restrict__ long_rows, const uint32_t *__restrict__ order_bits)
{
    constexpr int G = kWave / LPR;           // source rows per gather instruction
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / LPR;
    const int t = lane % LPR;
    const bool lane_on = (4 * t) < d;
    const int64_t n_virtual = n_rows + n_segments;
    const DealRange deal = deal_range(xcd_off, n_virtual, 1);
    const uint32_t seed0 = DROPOUT && dr.seed_words ? dr.seed_words[0] : dr.s0;
    const uint32_t seed1 = DROPOUT && dr.seed_words ? dr.seed_words[1] : dr.s1;
#ifdef IGCN_SPMM_TRACE
    const unsigned long long tr_begin = spmm_realtime();
    unsigned long long tr_rows = 0, tr_nnz = 0, tr_chunks = 0;
#endif

    DealCursor<SKIP, 1> cursor{deal, order_bits, lane, 0, 0ull};
    for (int64_t vv = cursor.begin(); vv < deal.end; vv = cursor.after(vv)) {
        int64_t start, end, dst;
        bool to_partial;
        const int64_t v = row_order ? (int64_t)row_order[vv] : vv;
        if (v < n_rows) {
            start = rowptr[v];
            end = rowptr[v + 1];
            if (!xcd_off && n_segments > 0 && end - start > long_threshold) continue;   // handled as segments (a plan's lists hold no cut row)
            dst = v;
            to_partial = false;
            if (row_mask && !row_mask[v]) {                                   // output not needed
                if (masked_rows_zero && g == 0 && lane_on)
                    *reinterpret_cast<float4 *>(y + dst * ldy + 4 * t) = f4_zero_here();
                continue;
            }
        } else {
            const igcn_row_segment s = segments[v - n_rows];
            if (row_mask && !row_mask[s.row]) {
                // (folded launches have no reduce kernel to zero a masked cut row: its closing segment does)
                if (FOLD && masked_rows_zero && (s.long_index & kClosingBit) && g == 0 && lane_on)
                    *reinterpret_cast<float4 *>(y + (int64_t)s.row * ldy + 4 * t) = f4_zero_here();
                continue;
            }
            start = s.start;
            end = s.start + s.len;
            dst = s.slot;
            to_partial = true;
        }

        float4 acc = f4_zero();
#ifdef IGCN_SPMM_TRACE
        ++tr_rows; tr_nnz += end - start; tr_chunks += (end - start + kWave - 1) / kWave;
#endif
        for (int64_t base = start; base < end; base += kWave) {
            const int64_t rem = end - base;
            const int cnt = rem < kWave ? (int)rem : kWave;     // wave-uniform
            int c = 0;
            float w = 0.f;
            if (lane < cnt) {
                const int64_t p = base + lane;
                c = col[p];
                w = val ? val[p] : 1.f;
                if (ep.col_scale) w *= ep.col_scale[c];
                if (ep.col_mask && !((ep.col_mask[c >> 5] >> (c & 31)) & 1u)) w = 0.f;
                if (DROPOUT) {
                    const uint64_t e = dr.edge_id ? (uint64_t)(uint32_t)dr.edge_id[p] : (uint64_t)p;
                    w = hash_counter(e, seed0, seed1) < dr.keep_below ? w / dr.keep_prob : 0.f;
                }
            }
            int k = 0;
            // main: 4 gather instructions in flight
            for (; k + 4 * G <= cnt; k += 4 * G) {
                int c0 = __shfl(c, k + g), c1 = __shfl(c, k + G + g), c2 = __shfl(c, k + 2 * G + g), c3 = __shfl(c, k + 3 * G + g);
                float w0 = __shfl(w, k + g), w1 = __shfl(w, k + G + g), w2 = __shfl(w, k + 2 * G + g), w3 = __shfl(w, k + 3 * G + g);
                float4 x0 = f4_zero(), x1 = f4_zero(), x2 = f4_zero(), x3 = f4_zero();
                // an edge of weight zero (dropped out, or its source row masked as zero) is not gathered
                if (lane_on && w0 != 0.f) x0 = *reinterpret_cast<const float4 *>(x + (int64_t)c0 * ldx + 4 * t);
                if (lane_on && w1 != 0.f) x1 = *reinterpret_cast<const float4 *>(x + (int64_t)c1 * ldx + 4 * t);
                if (lane_on && w2 != 0.f) x2 = *reinterpret_cast<const float4 *>(x + (int64_t)c2 * ldx + 4 * t);
                if (lane_on && w3 != 0.f) x3 = *reinterpret_cast<const float4 *>(x + (int64_t)c3 * ldx + 4 * t);
                f4_fma(acc, w0, x0); f4_fma(acc, w1, x1); f4_fma(acc, w2, x2); f4_fma(acc, w3, x3);
            }
            // tail: one gather instruction per step, groups past the end masked off
            for (; k < cnt; k += G) {
                const int src = k + g;
                const int cc = __shfl(c, src);
                const float ww = __shfl(w, src);
                if (lane_on && src < cnt && ww != 0.f) {
                    const float4 xv = *reinterpret_cast<const float4 *>(x + (int64_t)cc * ldx + 4 * t);
                    f4_fma(acc, ww, xv);
                }
            }
        }
        // fold the G groups (fixed order -> deterministic)
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1) f4_add(acc, f4_shfl_xor(acc, off));

        if (g == 0 && lane_on) {
            if (to_partial) {
                if constexpr (FOLD) store_partial_agent(partial + dst * (int64_t)d + 4 * t, acc);
                else *reinterpret_cast<float4 *>(partial + dst * (int64_t)d + 4 * t) = acc;
            } else {
                finish_row(acc, dst, t, ep, y, ldy);
            }
        }
        if (FOLD && to_partial) {                                             // (wave-uniform)
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            const int lane_c = cold(lane);                                   // (the fold's addresses are made here, not before the gather loop)
            segment_done<G, LPR>(long_rows, segments[dst].long_index, lane_c == 0, lane_c / LPR, lane_c % LPR, d, partial, n_segments, ep, y, ldy);   // (slot == index in `segments`)
        }
    }
#ifdef IGCN_SPMM_TRACE
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)));
    if (lane == 0 && wave0 < 131072) {
        g_spmm_wave_times[6 * wave0] = tr_begin;
        g_spmm_wave_times[6 * wave0 + 1] = spmm_realtime();
        g_spmm_wave_times[6 * wave0 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        g_spmm_wave_times[6 * wave0 + 3] = tr_rows; g_spmm_wave_times[6 * wave0 + 4] = tr_nnz; g_spmm_wave_times[6 * wave0 + 5] = tr_chunks;
    }
#endif
}

// Embeddings of up to 64 columns: R rows at a time per wave, one per sub-wave of S = 64/R lanes.
// With one row per wave a 32-byte-wide row keeps only a few of the 64 lanes busy per dependent step
// (row pointers -> col/val chunk -> gathers), and at these widths the kernel is bound by that latency
// chain, not by bandwidth; R independent chains per wave give R times the requests in flight.  Same
// arithmetic per row (each lane group sums its neighbours in storage order, groups folded in a fixed
// order); every control decision is per sub-wave, so the loops run while ANY sub-wave has work.
template <int LPR, int R, bool DROPOUT, bool FOLD = false, bool SKIP = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void spmm_csr_multirow_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
    int64_t n_rows, int d, SpmmEpilogue ep, SpmmDropout dr,
    const igcn_row_segment *__restrict__ segments, int64_t n_segments,
    float *__restrict__ partial, int long_threshold,
    const uint8_t *__restrict__ row_mask, int masked_rows_zero, const int32_t *__restrict__ row_order,
    const int64_t *__restrict__ xcd_off, const igcn_long_row *__restrict__ long_rows, const uint32_t *__restrict__ order_bits)
{
    constexpr int S = kWave / R;             // lanes of a sub-wave
    constexpr int G = S / LPR;               // source rows per gather instruction and sub-wave
    static_assert(G >= 1 && S * R == kWave && G * LPR == S, "bad sub-wave shape");
    const int lane = threadIdx.x & (kWave - 1);
    const int sub_base = lane / S * S;
    const int sl = lane % S;
    const int g = sl / LPR;
    const int t = sl % LPR;
    const bool lane_on = (4 * t) < d;
    const int64_t n_virtual = n_rows + n_segments;
    const DealRange deal = deal_range(xcd_off, n_virtual, R);
    const uint32_t seed0 = DROPOUT && dr.seed_words ? dr.seed_words[0] : dr.s0;
    const uint32_t seed1 = DROPOUT && dr.seed_words ? dr.seed_words[1] : dr.s1;
#ifdef IGCN_SPMM_TRACE
    const unsigned long long tr_begin = spmm_realtime();
    unsigned long long tr_rows = 0, tr_nnz = 0;
#endif

    DealCursor<SKIP, R> cursor{deal, order_bits, lane, 0, 0ull};
    for (int64_t vb = cursor.begin(); vb < deal.end; vb = cursor.after(vb)) {
        const int64_t vv = vb + lane / S;                         // this sub-wave's entry of the dealing order
        const int64_t v = vv >= deal.end ? n_virtual : row_order ? (int64_t)row_order[vv] : vv;
        // kind: 0 nothing, 1 ordinary row -> y, 2 row segment -> partial, 3 masked row that must read as zero
        int64_t start = 0;
        int dst = 0;                                             // row or slot (< 2^31): widened where an address is formed
        int len = 0, kind = 0;
        if (v < n_rows) {
            const int64_t s0 = rowptr[v], e0 = rowptr[v + 1];
            const bool is_long = !xcd_off && n_segments > 0 && e0 - s0 > long_threshold;    // (a plan's lists hold no cut row)
            const bool masked = row_mask && !row_mask[v];
            start = s0; len = (int)(e0 - s0); dst = (int)v;
            kind = is_long ? 0 : masked ? (masked_rows_zero ? 3 : 0) : 1;
        } else if (v < n_virtual) {
            const igcn_row_segment sg = segments[v - n_rows];
            const bool masked = row_mask && !row_mask[sg.row];
            start = sg.start; len = sg.len; dst = sg.slot;
            kind = masked ? 0 : 2;
            // (folded launches have no reduce kernel to zero a masked cut row: its closing segment does)
            if (FOLD && masked && masked_rows_zero && (sg.long_index & kClosingBit)) { kind = 3; dst = sg.row; }
        }
        if (kind != 1 && kind != 2) len = 0;
#ifdef IGCN_SPMM_TRACE
        if (sl == 0) { tr_rows += kind != 0; tr_nnz += len; }
#endif

        float4 acc = f4_zero();
        for (int off = 0; __any(off < len); off += S) {
            const int rem = len - off;
            const int cnt = rem < 0 ? 0 : rem < S ? rem : S;       // per sub-wave
            int c = 0;
            float w = 0.f;
            if (sl < cnt) {
                const int64_t p = start + off + sl;
                c = col[p];
                w = val ? val[p] : 1.f;
                if (ep.col_scale) w *= ep.col_scale[c];
                if (ep.col_mask && !((ep.col_mask[c >> 5] >> (c & 31)) & 1u)) w = 0.f;
                if (DROPOUT) {
                    const uint64_t e = dr.edge_id ? (uint64_t)(uint32_t)dr.edge_id[p] : (uint64_t)p;
                    w = hash_counter(e, seed0, seed1) < dr.keep_below ? w / dr.keep_prob : 0.f;
                }
            }
            // 4 gather instructions in flight; a group past the end of its row takes no part
            for (int k = 0; __any(k < cnt); k += 4 * G) {
                const int s0 = k + g, s1 = k + G + g, s2 = k + 2 * G + g, s3 = k + 3 * G + g;
                const int c0 = __shfl(c, sub_base + (s0 & (S - 1))), c1 = __shfl(c, sub_base + (s1 & (S - 1)));
                const int c2 = __shfl(c, sub_base + (s2 & (S - 1))), c3 = __shfl(c, sub_base + (s3 & (S - 1)));
                const float w0 = __shfl(w, sub_base + (s0 & (S - 1))), w1 = __shfl(w, sub_base + (s1 & (S - 1)));
                const float w2 = __shfl(w, sub_base + (s2 & (S - 1))), w3 = __shfl(w, sub_base + (s3 & (S - 1)));
                float4 x0 = f4_zero(), x1 = f4_zero(), x2 = f4_zero(), x3 = f4_zero();
                // an edge of weight zero (dropped out, or its source row masked as zero) is not gathered
                if (lane_on && s0 < cnt && w0 != 0.f) x0 = *reinterpret_cast<const float4 *>(x + (int64_t)c0 * ldx + 4 * t);
                if (lane_on && s1 < cnt && w1 != 0.f) x1 = *reinterpret_cast<const float4 *>(x + (int64_t)c1 * ldx + 4 * t);
                if (lane_on && s2 < cnt && w2 != 0.f) x2 = *reinterpret_cast<const float4 *>(x + (int64_t)c2 * ldx + 4 * t);
                if (lane_on && s3 < cnt && w3 != 0.f) x3 = *reinterpret_cast<const float4 *>(x + (int64_t)c3 * ldx + 4 * t);
                if (s0 < cnt) f4_fma(acc, w0, x0);
                if (s1 < cnt) f4_fma(acc, w1, x1);
                if (s2 < cnt) f4_fma(acc, w2, x2);
                if (s3 < cnt) f4_fma(acc, w3, x3);
            }
        }
        // fold the G groups of the sub-wave (fixed order -> deterministic)
#pragma unroll
        for (int o = LPR; o < S; o <<= 1) f4_add(acc, f4_shfl_xor(acc, o));

        if (g == 0 && lane_on) {
            if (kind == 3) {
                *reinterpret_cast<float4 *>(y + (int64_t)dst * ldy + 4 * t) = f4_zero_here();
            } else if (kind == 2) {
                if constexpr (FOLD) store_partial_agent(partial + (int64_t)dst * d + 4 * t, acc);
                else *reinterpret_cast<float4 *>(partial + (int64_t)dst * d + 4 * t) = acc;
            } else if (kind == 1) {
                finish_row(acc, dst, t, ep, y, ldy);
            }
        }
        if (FOLD && __any(kind == 2)) {
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            // (a segment's slot IS its index in `segments`: its row is read back here instead of being carried through the loop)
            if (kind == 2) {
                const int sl_c = cold(sl);                       // (the fold's addresses are made here, not before the gather loop)
                segment_done<G, LPR>(long_rows, segments[dst].long_index, sl_c == 0, sl_c / LPR, sl_c % LPR, d, partial, n_segments, ep, y, ldy);
            }
        }
    }
#ifdef IGCN_SPMM_TRACE
    {
        const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)));
        unsigned long long rows_w = tr_rows, nnz_w = tr_nnz;
        for (int o = S; o < kWave; o <<= 1) { rows_w += __shfl_xor(rows_w, o); nnz_w += __shfl_xor(nnz_w, o); }
        if (lane == 0 && wave0 < 131072) {
            g_spmm_wave_times[6 * wave0] = tr_begin;
            g_spmm_wave_times[6 * wave0 + 1] = spmm_realtime();
            g_spmm_wave_times[6 * wave0 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |
                                               ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);
            g_spmm_wave_times[6 * wave0 + 3] = rows_w; g_spmm_wave_times[6 * wave0 + 4] = nnz_w; g_spmm_wave_times[6 * wave0 + 5] = 0;
        }
    }
#endif
}

// The two-launch form (the default; "spmm_fold" 1 adds the rows up inside the launch instead): adds the partial sums of each cut row
// in the four-chain order of sum_partials_four_chains — the same bits as the in-launch form — and applies the epilogue.  One wave
// per long row, all its lane groups at work.
template <int LPR>
__global__ __launch_bounds__(kBlock) void spmm_long_rows_reduce_kernel(
    const igcn_long_row *__restrict__ long_rows, int64_t n_long, const float *__restrict__ partial,
    float *__restrict__ y, int64_t ldy, int d, SpmmEpilogue ep, const uint8_t *__restrict__ row_mask, int masked_rows_zero)
{
    constexpr int G = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / LPR, t = lane % LPR;
    const bool lane_on = (4 * t) < d;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (wave >= n_long) return;
    const igcn_long_row lr = long_rows[wave];
    if (row_mask && !row_mask[lr.row]) {
        if (masked_rows_zero && g == 0 && lane_on) *reinterpret_cast<float4 *>(y + (int64_t)lr.row * ldy + 4 * t) = f4_zero();
        return;
    }
    const float4 acc = sum_partials_four_chains<G, LPR, 4>(partial + (int64_t)lr.first_slot * d + 4 * t, lr.n_slots, d, g, lane_on, false);
    if (g == 0 && lane_on) finish_row(acc, lr.row, t, ep, y, ldy);
}

// Any d (not a multiple of 4, or misaligned leading dimensions): one wave per
// row, lane j owns columns j, j+64, ...; slow path, used for odd shapes only.
__global__ __launch_bounds__(kBlock) void spmm_csr_scalar_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
    int64_t n_rows, int d, SpmmEpilogue ep, SpmmDropout dr, bool dropout)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave);
    const uint32_t seed0 = dropout && dr.seed_words ? dr.seed_words[0] : dr.s0;
    const uint32_t seed1 = dropout && dr.seed_words ? dr.seed_words[1] : dr.s1;
    for (int64_t r = wave0; r < n_rows; r += n_waves) {
        const int64_t start = rowptr[r], end = rowptr[r + 1];
        for (int j0 = 0; j0 < d; j0 += kWave) {
            const int j = j0 + lane;
            float acc = 0.f;
            for (int64_t p = start; p < end; ++p) {
                float w = val ? val[p] : 1.f;
                if (ep.col_scale) w *= ep.col_scale[col[p]];
                if (ep.col_mask && !((ep.col_mask[col[p] >> 5] >> (col[p] & 31)) & 1u)) w = 0.f;
                if (dropout) {
                    const uint64_t e = dr.edge_id ? (uint64_t)(uint32_t)dr.edge_id[p] : (uint64_t)p;
                    w = hash_counter(e, seed0, seed1) < dr.keep_below ? w / dr.keep_prob : 0.f;
                }
                // as in the vector kernels: a zero-weight edge (masked source row, dropped edge) is not read —
                // a masked source row may hold uninitialised memory (0 * NaN would poison the sum)
                if (j < d && w != 0.f) acc = fmaf(w, x[(int64_t)col[p] * ldx + j], acc);
            }
            if (j < d) {
                float rr = acc * ep.out_scale;
                float s = 0.f;
                for (int i = 0; i < ep.n_adds; ++i) s += ep.add[i][r * ldy + j];
                if (ep.n_adds > 0) rr = fmaf(ep.add_scale, s, rr);
                if (ep.row_scale) rr *= ep.row_scale[r];
                y[r * ldy + j] = rr;
            }
        }
    }
}

__global__ void csr_row_pow_kernel(const int64_t *__restrict__ rowptr, const float *__restrict__ row_sum,
                                   float exponent, float *__restrict__ val_out, float *__restrict__ row_scale_out,
                                   int64_t n_rows)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t r = wave0; r < n_rows; r += n_waves) {
        const float v = powf(row_sum[r], exponent);
        if (row_scale_out && lane == 0) row_scale_out[r] = v;
        if (val_out)
            for (int64_t p = rowptr[r] + lane; p < rowptr[r + 1]; p += kWave) val_out[p] = v;
    }
}

int g_tuning[IGCN_TUNE_COUNT];          // value + 1; 0 (the static initialiser) = library default

// Developer tuning knobs (igcn_set_tuning): spmm_blocks_per_cu, spmm_multirow.
// A launch reads them ONCE, here, into a value of its own: `over` = the per-call knobs of igcn_spmm_args (value + 1, 0 = not given),
// which win over the process-wide ones — concurrent callers with different launch shapes use those, not igcn_set_tuning.
struct SpmmTuning { int blocks_per_cu; int multirow; int fold; };
struct SpmmTuneOverride { int blocks_per_cu = 0, multirow = 0, fold = 0; };
static SpmmTuning tuning(const SpmmTuneOverride &over) {
    SpmmTuning v{0, 1, 0};                                    // 0 = sized by rows per wave; two rows per wave; the reduce kernel
    const int b = over.blocks_per_cu > 0 ? over.blocks_per_cu - 1 : tuning_get(IGCN_TUNE_SPMM_BLOCKS_PER_CU);
    const int m = over.multirow > 0 ? over.multirow - 1 : tuning_get(IGCN_TUNE_SPMM_MULTIROW);
    const int f = over.fold > 0 ? over.fold - 1 : tuning_get(IGCN_TUNE_SPMM_FOLD);
    if (b >= 1 && b <= 4096) v.blocks_per_cu = b;
    if (m >= 0) v.multirow = m != 0;
    if (f >= 0) v.fold = f != 0;
    return v;
}

// mask1[id] = 1 for the listed rows; mask2[id] = 1 and mask2[c] = 1 for every column c of a listed row.
__global__ void mark_rows_kernel(const int64_t *__restrict__ ids, int64_t n, const int64_t *__restrict__ rowptr,
                                 const int32_t *__restrict__ col, uint8_t *__restrict__ mask1, uint8_t *__restrict__ mask2)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t r = ids[i];
    if (lane == 0) { mask1[r] = 1; if (mask2) mask2[r] = 1; }
    if (mask2 && rowptr)
        for (int64_t p = rowptr[r] + lane; p < rowptr[r + 1]; p += kWave) mask2[col[p]] = 1;
}

// uint8 masks [n_masks][stride] -> one bit per entry, ceil(n / 32) words per mask; a wave packs 64 entries
__global__ __launch_bounds__(kBlock) void pack_mask_bits_kernel(const uint8_t *__restrict__ masks, int64_t n, int64_t stride,
                                                                int64_t words, int64_t pairs, int n_masks, uint32_t *__restrict__ bits)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wv = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (wv >= pairs * n_masks) return;
    const int64_t m = wv / pairs, pair = wv % pairs;
    const int64_t e = pair * kWave + lane;
    const bool on = e < n && masks[m * stride + e] != 0;
    const unsigned long long b = __ballot(on);
    if (lane == 0) {
        bits[m * words + 2 * pair] = (uint32_t)b;
        if (2 * pair + 1 < words) bits[m * words + 2 * pair + 1] = (uint32_t)(b >> 32);
    }
}

// pack_mask_bits_kernel + the FIRST mask once more in the dealing order of a matrix (DealCursor): bit vv of order_bits = the row of
// entry vv — row_order[vv], or the row of the segment it names — is set in masks[0].  One launch (a captured training step pays
// ~5 us per kernel node); the waves behind the packing ones do the ordered bits; the last of them zeroes the two padding words.
__global__ __launch_bounds__(kBlock) void pack_mask_bits_ordered_kernel(const uint8_t *__restrict__ masks, int64_t n, int64_t stride,
                                                                        int64_t words, int64_t pairs, int n_masks, uint32_t *__restrict__ bits,
                                                                        const int32_t *__restrict__ row_order,
                                                                        const igcn_row_segment *__restrict__ segments, int64_t n_virtual,
                                                                        int64_t pairs_o, uint32_t *__restrict__ order_bits)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wv = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (wv < pairs * n_masks) {
        const int64_t m = wv / pairs, pair = wv % pairs;
        const int64_t e = pair * kWave + lane;
        const bool on = e < n && masks[m * stride + e] != 0;
        const unsigned long long b = __ballot(on);
        if (lane == 0) {
            bits[m * words + 2 * pair] = (uint32_t)b;
            if (2 * pair + 1 < words) bits[m * words + 2 * pair + 1] = (uint32_t)(b >> 32);
        }
        return;
    }
    const int64_t pair = wv - pairs * n_masks;
    if (pair >= pairs_o) return;
    const int64_t vv = pair * kWave + lane;
    bool on = false;
    if (vv < n_virtual) {
        const int64_t v = row_order ? (int64_t)row_order[vv] : vv;
        const int64_t row = v < n ? v : (int64_t)segments[v - n].row;
        on = masks[row] != 0;
    }
    const unsigned long long b = __ballot(on);
    if (lane == 0) {
        order_bits[2 * pair] = (uint32_t)b;
        order_bits[2 * pair + 1] = (uint32_t)(b >> 32);
        if (pair == pairs_o - 1) { order_bits[2 * pairs_o] = 0u; order_bits[2 * pairs_o + 1] = 0u; }
    }
}

// Workgroups of this kernel variant that a CU really holds at once (a floor for the grid).  Rows are
// dealt to the waves round-robin, so a grid only slightly larger than what is resident is the worst
// case: measured on MI355X (scripts/dev_spmm_trace.py), 8 workgroups per CU of the d = 64 variant left 1
// in 8 starting 77 us late in a 156 us kernel, although hipOccupancyMaxActiveBlocksPerMultiprocessor
// answers 8 (7 per CU: 17 % faster; the query was one too high for every variant measured,
// scripts/dev_spmm_blocks_ab.py, scripts/probes/residency_probe.hip).  One less than it says is safe.
template <int LPR, bool DROPOUT>
static int resident_blocks_per_cu() {
    static const int n = [] {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spmm_csr_rows_kernel<LPR, DROPOUT>, kBlock, 0) != hipSuccess) nb = 0;
        (void)hipGetLastError();
        return nb > 1 ? nb - 1 : (nb == 1 ? 1 : 6);
    }();
    return n;
}

template <int LPR>
static int launch_rows(bool dropout, int64_t blocks, hipStream_t st,
                       const int64_t *rowptr, const int32_t *col, const float *val, const float *x, int64_t ldx,
                       float *y, int64_t ldy, int64_t n_rows, int d, const SpmmEpilogue &ep, const SpmmDropout &dr,
                       const igcn_row_segment *segments, int64_t n_segments, float *partial, int long_threshold,
                       const igcn_long_row *long_rows, int64_t n_long, const uint8_t *row_mask, int masked_rows_zero,
                       int64_t nnz, const int32_t *row_order, const int64_t *xcd_off, bool closing, const uint32_t *order_bits,
                       const SpmmTuning &tune)
{
    // Grid: `blocks` on entry = one wave per row.  Fewer, longer-lived waves amortise the per-wave set-up;
    // many short ones let the hardware dispatcher even out the load (a power-law graph deals very
    // different amounts of work to equal row counts: per-wave time follows the nonzeros, 43 ns each,
    // scripts/dev_spmm_trace.py).  Measured on MI355X, two boxes (profiles/r01f_*): for the Amazon-like
    // graph (21 nonzeros per row) the best grids have 1 row per wave at d = 128, 1-4 at d = 64, 4-8 at
    // d <= 32, all within 5 % of each other and 10-20 % ahead of a grid of just the resident waves;
    // heavier rows get proportionally fewer per wave.  Never fewer workgroups than are resident at once.
    const int resident = dropout ? resident_blocks_per_cu<LPR, true>() : resident_blocks_per_cu<LPR, false>();
    const int64_t n_virtual = n_rows + n_segments;
    int64_t want;
    if (tune.blocks_per_cu > 0) {
        want = (int64_t)cu_count() * tune.blocks_per_cu;
    } else {
        const int64_t base_rows = d >= 128 ? 1 : d >= 64 ? 2 : 6;           // at 21 nonzeros per row
        const int64_t mean_deg = nnz > 0 ? (nnz + n_virtual - 1) / n_virtual : 21;
        int64_t rows_per_wave = (base_rows * 21 + mean_deg / 2) / (mean_deg > 0 ? mean_deg : 1);
        if (rows_per_wave < 1) rows_per_wave = 1;
        if (rows_per_wave > 8) rows_per_wave = 8;
        want = (n_virtual + rows_per_wave * (kBlock / kWave) - 1) / (rows_per_wave * (kBlock / kWave));
        const int64_t fill = (int64_t)cu_count() * resident;
        if (want < fill) want = fill;
    }
    if (blocks > want) blocks = want;
    if (xcd_off) blocks = (blocks + 7) / 8 * 8;                       // the same number of workgroups for every list
    const dim3 grid((unsigned)blocks);
    // rows a wave works on at once: 2 at d = 64 and 32 (-7 % / -23 %), 4 below (-17...-22 %; 8 was measured too: no
    // different — at d <= 16 the kernel then moves ~8 TB/s of 128-byte lines, the gather granularity, and is bound
    // by that); at d >= 128 one row already fills the wave's loads (2: +1...6 %)
    constexpr int R = LPR >= 32 ? 1 : LPR >= 8 ? 2 : 4;
    const bool multirow = R > 1 && tune.multirow;
    // Cut rows added up inside the launch by their closing segments: OPT-IN ("spmm_fold" 1) — measured in round 5
    // (profiles/r05c_spmm_fold_in_launch_negative.jsonl): bit-equal to the two-launch form everywhere; -1.2 ... -3.6 % per pass with
    // the plain plan (3 761 segments), -1.8 % Gowalla-like, +-0 Yelp-like, but +5 % on the headline graph with the XCD plan (35 141
    // segments of 4 133 rows): the closing segments' agent-scope reads of the partial sums and the epilogue behind them cost more
    // inside the launch (+8 us) than the 5 us kernel they replace.  Default: the second kernel.
    const igcn_long_row *fold = n_long > 0 && closing && tune.fold ? long_rows : nullptr;
#define IGCN_SPMM_LAUNCH(...)                                                                                                          \
    hipLaunchKernelGGL((__VA_ARGS__), grid, dim3(kBlock), 0, st, rowptr, col, val, x, ldx, y, ldy, n_rows, d, ep, dr, segments, n_segments, \
                       partial, long_threshold, row_mask, masked_rows_zero, row_order, xcd_off, fold, skip)
    // (skipping by the mask in dealing order: only where masked rows need no zeros, nothing is dropped out and no row is folded)
    const uint32_t *skip = row_mask && !masked_rows_zero && !dropout && !fold ? order_bits : nullptr;
    if (skip) {
        if (multirow) { if constexpr (R > 1) IGCN_SPMM_LAUNCH(spmm_csr_multirow_kernel<LPR, R, false, false, true>); }
        else IGCN_SPMM_LAUNCH(spmm_csr_rows_kernel<LPR, false, false, true>);
    } else if (multirow) {
        if constexpr (R > 1) {
            if (fold) { if (dropout) IGCN_SPMM_LAUNCH(spmm_csr_multirow_kernel<LPR, R, true, true>); else IGCN_SPMM_LAUNCH(spmm_csr_multirow_kernel<LPR, R, false, true>); }
            else if (dropout) IGCN_SPMM_LAUNCH(spmm_csr_multirow_kernel<LPR, R, true, false>);
            else IGCN_SPMM_LAUNCH(spmm_csr_multirow_kernel<LPR, R, false, false>);
        }
    } else if (fold) { if (dropout) IGCN_SPMM_LAUNCH(spmm_csr_rows_kernel<LPR, true, true>); else IGCN_SPMM_LAUNCH(spmm_csr_rows_kernel<LPR, false, true>); }
    else if (dropout) IGCN_SPMM_LAUNCH(spmm_csr_rows_kernel<LPR, true, false>);
    else IGCN_SPMM_LAUNCH(spmm_csr_rows_kernel<LPR, false, false>);
#undef IGCN_SPMM_LAUNCH
    int rc = launch_status();
    if (rc != IGCN_OK) return rc;
    if (n_long > 0 && !fold) {
        const int64_t blocks = (n_long + (kBlock / kWave) - 1) / (kBlock / kWave);
        hipLaunchKernelGGL((spmm_long_rows_reduce_kernel<LPR>), dim3((unsigned)blocks), dim3(kBlock), 0, st,
                           long_rows, n_long, partial, y, ldy, d, ep, row_mask, masked_rows_zero);
        rc = launch_status();
    }
    return rc;
}

}  // namespace igcn

using namespace igcn;

extern "C" int igcn_spmm_plan_count_host(const int64_t *rowptr_host, int64_t n_rows, int32_t long_threshold,
                                         int32_t segment_len, int64_t *n_long_rows, int64_t *n_segments)
{
    if (!rowptr_host || !n_long_rows || !n_segments) return IGCN_E_NULL;
    if (n_rows < 0) return IGCN_E_SHAPE;
    if (long_threshold < 1 || segment_len < 1 || segment_len > long_threshold) return IGCN_E_RANGE;
    int64_t nl = 0, ns = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t deg = rowptr_host[r + 1] - rowptr_host[r];
        if (deg < 0) return IGCN_E_SHAPE;
        if (deg > long_threshold) { ++nl; ns += (deg + segment_len - 1) / segment_len; }
    }
    *n_long_rows = nl;
    *n_segments = ns;
    return IGCN_OK;
}

extern "C" int igcn_spmm_plan_fill_host(const int64_t *rowptr_host, int64_t n_rows, int32_t long_threshold,
                                        int32_t segment_len, igcn_long_row *long_rows_host, int64_t n_long_rows,
                                        igcn_row_segment *segments_host, int64_t n_segments)
{
    if (!rowptr_host) return IGCN_E_NULL;
    if ((n_long_rows > 0 && !long_rows_host) || (n_segments > 0 && !segments_host)) return IGCN_E_NULL;
    if (long_threshold < 1 || segment_len < 1 || segment_len > long_threshold) return IGCN_E_RANGE;
    if (n_rows >= (int64_t)1 << 31) return IGCN_E_SHAPE;
    int64_t il = 0, is = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t s = rowptr_host[r], e = rowptr_host[r + 1];
        if (e - s <= long_threshold) continue;
        const int64_t n = (e - s + segment_len - 1) / segment_len;
        if (il >= n_long_rows || is + n > n_segments || is + n >= (int64_t)1 << 31) return IGCN_E_SHAPE;
        long_rows_host[il].row = (int32_t)r;
        long_rows_host[il].first_slot = (int32_t)is;
        long_rows_host[il].n_slots = (int32_t)n;
        long_rows_host[il].reserved = 0;
        const int32_t this_long = (int32_t)il;
        ++il;
        for (int64_t p = s; p < e; p += segment_len, ++is) {
            segments_host[is].start = p;
            segments_host[is].len = (int32_t)((e - p) < segment_len ? (e - p) : segment_len);
            segments_host[is].slot = (int32_t)is;
            segments_host[is].row = (int32_t)r;
            segments_host[is].long_index = this_long | (p + segment_len >= e ? (int32_t)0x80000000u : 0);   // the row's last segment closes it
        }
    }
    return (il == n_long_rows && is == n_segments) ? IGCN_OK : IGCN_E_SHAPE;
}

static int spmm_run(const int64_t *rowptr, const int32_t *col, const float *val,
                    const float *x, int64_t ldx, float *y, int64_t ldy,
                    int64_t n_rows, int64_t n_cols, int32_t d,
                    float out_scale, const float *const *adds_host, int32_t n_adds,
                    float add_scale, const float *row_scale, const float *col_scale,
                    const igcn_long_row *long_rows, int64_t n_long_rows,
                    const igcn_row_segment *segments, int64_t n_segments,
                    float *partial, int32_t long_threshold,
                    const int32_t *edge_id, uint64_t seed, float keep_prob,
                    const uint8_t *row_mask, int32_t flags,
                    int64_t nnz, const int32_t *row_order, const uint32_t *col_mask,
                    const uint64_t *seed_dev, const int64_t *xcd_off, const uint32_t *order_bits, void *stream,
                    const SpmmTuneOverride &over)
{
    const SpmmTuning tune = tuning(over);
    const int32_t masked_rows_zero = flags & IGCN_SPMM_MASKED_ROWS_ZERO;
    // cut rows added up inside the launch: only with a plan that marks closing segments and deals them late (the caller says so)
    const bool closing = (flags & IGCN_SPMM_CLOSING_SEGMENTS) != 0 && row_order != nullptr;
    if (!rowptr || !x || !y) return IGCN_E_NULL;
    if (n_rows < 0 || n_cols < 0 || d < 1 || d > 256 || ldx < d || ldy < d) return IGCN_E_SHAPE;
    if (n_rows >= ((int64_t)1 << 31) || n_cols >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    if (n_adds < 0 || n_adds > IGCN_MAX_ADDS || (n_adds > 0 && !adds_host)) return IGCN_E_RANGE;
    if (!(keep_prob > 0.f)) return IGCN_E_RANGE;
    if ((n_long_rows > 0 || n_segments > 0) && (!long_rows || !segments || !partial || n_long_rows < 1 || n_segments < 1))
        return IGCN_E_NULL;
    if (n_segments > 0 && long_threshold < 1) return IGCN_E_RANGE;
    if (xcd_off && !row_order) return IGCN_E_NULL;
    if (order_bits && !row_mask) return IGCN_E_NULL;       // the mask in dealing order comes with the mask itself
    if (x == y) return IGCN_E_RANGE;     // in-place propagation would read rows being written
    if (n_rows == 0) return IGCN_OK;
    // col may be NULL only for a matrix without stored entries (rowptr all zero): it is never read then

    SpmmEpilogue ep{};
    ep.n_adds = n_adds;
    for (int i = 0; i < n_adds; ++i) {
        if (!adds_host[i]) return IGCN_E_NULL;
        ep.add[i] = adds_host[i];
    }
    ep.out_scale = out_scale;
    ep.add_scale = add_scale;
    ep.row_scale = row_scale;
    ep.col_scale = col_scale;
    ep.col_mask = col_mask;

    const bool dropout = keep_prob < 1.f;
    SpmmDropout dr{};
    dr.edge_id = edge_id;
    dr.s0 = (uint32_t)seed;
    dr.s1 = (uint32_t)(seed >> 32);
    dr.seed_words = reinterpret_cast<const uint32_t *>(seed_dev);
    double kb = (double)keep_prob * 4294967296.0;
    dr.keep_below = kb >= 4294967295.0 ? 4294967295u : (uint32_t)kb;
    dr.keep_prob = keep_prob;

    hipStream_t st = static_cast<hipStream_t>(stream);
    const int waves_per_block = kBlock / kWave;
    const int64_t n_virtual = n_rows + n_segments;
    int64_t blocks = (n_virtual + waves_per_block - 1) / waves_per_block;
    const int64_t scalar_cap = (int64_t)cu_count() * 8;
    const dim3 scalar_grid((unsigned)(blocks < scalar_cap ? blocks : scalar_cap));

    bool vec = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) &&
               ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0);
    for (int i = 0; i < n_adds && vec; ++i) vec = reinterpret_cast<uintptr_t>(ep.add[i]) % 16 == 0;
    if (n_segments > 0 && (reinterpret_cast<uintptr_t>(partial) % 16 != 0)) return IGCN_E_ALIGN;
    if (!vec) {
        if (n_segments > 0 || row_mask || row_order || xcd_off) return IGCN_E_ALIGN;   // plan / row masks / row order need the vector path
        hipLaunchKernelGGL(spmm_csr_scalar_kernel, scalar_grid, dim3(kBlock), 0, st, rowptr, col, val, x, ldx, y, ldy,
                           n_rows, (int)d, ep, dr, dropout);
        return launch_status();
    }
#define IGCN_SPMM_CASE(L)                                                                                         \
    return launch_rows<L>(dropout, blocks, st, rowptr, col, val, x, ldx, y, ldy, n_rows, (int)d, ep, dr, segments,  \
                          n_segments, partial, (int)long_threshold, long_rows, n_long_rows, row_mask, (int)masked_rows_zero, \
                          nnz, row_order, xcd_off, closing, order_bits, tune)
    const int q = d / 4;
    if (q <= 1) IGCN_SPMM_CASE(1);
    if (q <= 2) IGCN_SPMM_CASE(2);
    if (q <= 4) IGCN_SPMM_CASE(4);
    if (q <= 8) IGCN_SPMM_CASE(8);
    if (q <= 16) IGCN_SPMM_CASE(16);
    if (q <= 32) IGCN_SPMM_CASE(32);
    IGCN_SPMM_CASE(64);
#undef IGCN_SPMM_CASE
}

extern "C" int igcn_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                                 const float *x, int64_t ldx, float *y, int64_t ldy,
                                 int64_t n_rows, int64_t n_cols, int32_t d,
                                 float out_scale, const float *const *adds_host, int32_t n_adds,
                                 float add_scale, const float *row_scale, const float *col_scale,
                                 const igcn_long_row *long_rows, int64_t n_long_rows,
                                 const igcn_row_segment *segments, int64_t n_segments,
                                 float *partial, int32_t long_threshold,
                                 const int32_t *edge_id, uint64_t seed, float keep_prob,
                                 const uint8_t *row_mask, int32_t flags,
                                 int64_t nnz, const int32_t *row_order, const uint32_t *col_mask,
                                 const uint64_t *seed_dev, const int64_t *xcd_off, const uint32_t *order_bits, void *stream)
{
    return spmm_run(rowptr, col, val, x, ldx, y, ldy, n_rows, n_cols, d, out_scale, adds_host, n_adds, add_scale, row_scale, col_scale,
                    long_rows, n_long_rows, segments, n_segments, partial, long_threshold, edge_id, seed, keep_prob, row_mask, flags,
                    nnz, row_order, col_mask, seed_dev, xcd_off, order_bits, stream, SpmmTuneOverride{});
}

// The same launch through ONE struct (ABI v10).  Only the first `struct_size` bytes are read — a caller compiled against a shorter
// (older) struct keeps working, what it does not know is zero —, and zero means "not used / library default" in every optional
// field, so a binding of the reference call site (model.py:99-102: graph, X, vals) fills eight fields of a zeroed struct.
extern "C" int igcn_spmm_csr_f32_args(const igcn_spmm_args *args, void *stream)
{
    if (!args) return IGCN_E_NULL;
    const size_t need = offsetof(igcn_spmm_args, d) + sizeof(int32_t);       // the required prefix
    if (args->struct_size < need) return IGCN_E_SHAPE;
    igcn_spmm_args a;
    memset(&a, 0, sizeof a);
    memcpy(&a, args, args->struct_size < sizeof a ? (size_t)args->struct_size : sizeof a);
    if (a.flags & ~(uint32_t)(IGCN_SPMM_MASKED_ROWS_ZERO | IGCN_SPMM_CLOSING_SEGMENTS)) return IGCN_E_RANGE;
    if (a.n_adds < 0 || a.n_adds > IGCN_MAX_ADDS) return IGCN_E_RANGE;
    if (a.out_scale != a.out_scale || a.add_scale != a.add_scale || a.keep_prob != a.keep_prob) return IGCN_E_RANGE;     // NaN
    if (a.tune_blocks_per_cu < 0 || a.tune_multirow < 0 || a.tune_fold < 0) return IGCN_E_RANGE;
    SpmmTuneOverride over;
    over.blocks_per_cu = a.tune_blocks_per_cu;
    over.multirow = a.tune_multirow;
    over.fold = a.tune_fold;
    return spmm_run(a.rowptr, a.col, a.val, a.x, a.ldx > 0 ? a.ldx : a.d, a.y, a.ldy > 0 ? a.ldy : a.d, a.n_rows, a.n_cols, a.d,
                    a.out_scale == 0.f ? 1.f : a.out_scale, a.adds, a.n_adds, a.add_scale == 0.f ? 1.f : a.add_scale, a.row_scale,
                    a.col_scale, a.long_rows, a.n_long_rows, a.segments, a.n_segments, a.partial, a.long_threshold, a.edge_id, a.seed,
                    a.keep_prob == 0.f ? 1.f : a.keep_prob, a.row_mask, (int32_t)a.flags, a.nnz, a.row_order, a.col_mask, a.seed_dev,
                    a.xcd_off, a.order_bits, stream, over);
}

extern "C" int igcn_mark_rows(const int64_t *ids, int64_t n, const int64_t *rowptr, const int32_t *col,
                              uint8_t *mask1, uint8_t *mask2, int64_t n_rows, void *stream)
{
    if (!ids || !mask1) return IGCN_E_NULL;
    if (mask2 && (!rowptr || !col)) return IGCN_E_NULL;
    if (n < 0 || n_rows < 0) return IGCN_E_SHAPE;
    if (n == 0) return IGCN_OK;
    hipLaunchKernelGGL(mark_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       ids, n, rowptr, col, mask1, mask2);
    return launch_status();
}

extern "C" int igcn_pack_mask_bits(const uint8_t *masks, int64_t n, int64_t stride, int32_t n_masks, uint32_t *bits, void *stream)
{
    if (!masks || !bits) return IGCN_E_NULL;
    if (n < 0 || n_masks < 1 || stride < n) return IGCN_E_SHAPE;
    if (n == 0) return IGCN_OK;
    const int64_t words = (n + 31) / 32, pairs = (n + kWave - 1) / kWave;
    const int64_t waves = pairs * n_masks;
    hipLaunchKernelGGL(pack_mask_bits_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       masks, n, stride, words, pairs, (int)n_masks, bits);
    return launch_status();
}

extern "C" int igcn_pack_mask_bits_ordered(const uint8_t *masks, int64_t n, int64_t stride, int32_t n_masks, uint32_t *bits,
                                           const int32_t *row_order, int64_t n_order, const igcn_row_segment *segments,
                                           int64_t n_segments, uint32_t *order_bits, void *stream)
{
    if (!masks || !bits || !order_bits) return IGCN_E_NULL;
    if (n < 0 || n_masks < 1 || stride < n || n_segments < 0) return IGCN_E_SHAPE;
    if (n_segments > 0 && !segments) return IGCN_E_NULL;
    if (n + n_segments >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    // entries of the dealing order: all rows and segments without one; with one, what it lists (a plan's lists hold no cut row)
    if (!row_order) n_order = n + n_segments;
    if (n_order < 0 || n_order > n + n_segments) return IGCN_E_SHAPE;
    if (n == 0) return IGCN_OK;
    const int64_t words = (n + 31) / 32, pairs = (n + kWave - 1) / kWave;
    const int64_t n_virtual = n_order, pairs_o = (n_virtual + kWave - 1) / kWave;
    const int64_t waves = pairs * n_masks + pairs_o;
    hipLaunchKernelGGL(pack_mask_bits_ordered_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       masks, n, stride, words, pairs, (int)n_masks, bits, row_order, segments, n_virtual, pairs_o, order_bits);
    return launch_status();
}

extern "C" int igcn_csr_row_pow_f32(const int64_t *rowptr, const float *row_sum, float exponent,
                                    float *val_out, float *row_scale_out, int64_t n_rows, void *stream)
{
    if (!rowptr || !row_sum || (!val_out && !row_scale_out)) return IGCN_E_NULL;
    if (n_rows < 0) return IGCN_E_SHAPE;
    if (n_rows == 0) return IGCN_OK;
    int64_t blocks = (n_rows + 3) / 4;
    const int64_t max_blocks = (int64_t)cu_count() * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(csr_row_pow_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       rowptr, row_sum, exponent, val_out, row_scale_out, n_rows);
    return launch_status();
}

#ifdef IGCN_SPMM_TRACE
extern "C" int igcn_debug_spmm_wave_times(unsigned long long *host, int n_waves)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(host, HIP_SYMBOL(igcn::g_spmm_wave_times), (size_t)n_waves * 48);
    return (int)e;
}
#endif

extern "C" int igcn_abi_version(void) { return IGCN_ABI_VERSION; }

extern "C" int igcn_set_tuning(const char *name, int32_t value)
{
    static const char *const names[IGCN_TUNE_COUNT] = {"spmm_blocks_per_cu", "spmm_multirow", "topk_slots",
                                                       "topk_waves_per_cu", "topk_cap", "topk_stagger", "topk_fast_mode",
                                                       "topk_fast_order", "topk_fast_exit", "topk_fast_wide", "topk_fast_extra", "topk_fast_give_up", "topk_fast_narrow", "topk_fast_share",
                                                       "topk_fast_fallback", "topk_fast_early_checks", "topk_fast_warm", "topk_fast_filter", "topk_fast_pieces", "spmm_fold",
                                                       "topk_fast_poison"};
    if (!name) return IGCN_E_NULL;
    for (int i = 0; i < IGCN_TUNE_COUNT; ++i)
        if (strcmp(name, names[i]) == 0) { g_tuning[i] = value < 0 ? 0 : value + 1; return IGCN_OK; }
    return IGCN_E_RANGE;
}

extern "C" const char *igcn_error_string(int code)
{
    switch (code) {
    case IGCN_OK: return "ok";
    case IGCN_E_NULL: return "a required pointer is NULL";
    case IGCN_E_SHAPE: return "invalid size or leading dimension";
    case IGCN_E_ALIGN: return "pointer or stride not 16-byte aligned";
    case IGCN_E_RANGE: return "scalar argument out of range";
    case IGCN_E_NO_DEVICE: return "no HIP device";
    case IGCN_E_CAPTURE: return "stream is capturing and a kernel of this call uses scratch memory (rebuild: the kernels are meant to have none)";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown igcn error";
    }
}
