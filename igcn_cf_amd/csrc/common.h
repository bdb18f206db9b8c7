// Shared device/host helpers for libigcn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "igcn_hip.h"

namespace igcn {

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kBlock = 256;    // 4 waves per workgroup

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? IGCN_OK : static_cast<int>(e);
}

// No kernel of this library carries a private segment (scratch: register spills, dynamic indexing of a local array) —
// tests/test_host_cpu.py reads the private segment sizes out of the built code object, and build() fails on a non-zero one.
// Round 4 blamed a GPU fault at the first replay of a captured igcn_score_topk_fast_f32 on the 32-76 bytes a lane its sweep kernels
// spilled then (scratch is per-queue state the replaying queue may not have been given).  Round 5 removed the spills and the fault
// stayed; under rocgdb it turned out to be the call's memset NODE (see zero_async below).  The rule is kept as what it is, a
// property a drop-in should have (nothing of a call depends on per-queue state), and this is the guard behind it: a launch site
// calls it once per kernel; should a later compiler bring scratch back, the call is REFUSED while the stream is capturing
// (IGCN_E_CAPTURE) rather than left to chance at replay time.  Eager launches are unaffected.
inline int capture_guard(const void *kernel, hipStream_t st, int *cached_scratch_bytes) {
    if (*cached_scratch_bytes < 0) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) { (void)hipGetLastError(); return IGCN_OK; }
        *cached_scratch_bytes = (int)fa.localSizeBytes;
    }
    if (*cached_scratch_bytes == 0) return IGCN_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return IGCN_OK; }
    return cs == hipStreamCaptureStatusNone ? IGCN_OK : IGCN_E_CAPTURE;
}

// Zeroing device memory from inside the library: a KERNEL of its own, never hipMemsetAsync.  Round 5, under rocgdb
// (profiles/r05b_capture_fault_rocgdb.txt): a captured igcn_score_topk_fast_f32 replayed by PyTorch's standard recipe (warm-up on a
// side stream, capture, replay on the current stream, fresh process) faulted in order_place_kernel, "write access to a read-only
// page" — its counting bins, which "arrive zeroed" by the call's hipMemsetAsync, had not: on ROCm 7.2 the memset NODE of a captured
// graph is not ordered against the kernel nodes behind it the way an eager hipMemsetAsync is ordered against the kernels behind it
// on its stream (the process has a DMA queue beside its compute queues; the fault round 4 blamed on scratch).  With the zeroing
// done by a kernel node the same recipe replays cleanly (tests/capture_child.py).  Kernel nodes are ordered against each other —
// everything else in the library relies on that, and only on that.  p: 4-byte aligned; any byte count.
static __global__ __launch_bounds__(256) void zero_bytes_kernel(uint32_t *__restrict__ p, int64_t n_words, int64_t n_bytes)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n_quads = n_words >> 2;
    uint4 *q = reinterpret_cast<uint4 *>(p);
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0)
        for (; i < n_quads; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
    else
        for (; i < n_quads; i += stride) { p[4 * i] = 0u; p[4 * i + 1] = 0u; p[4 * i + 2] = 0u; p[4 * i + 3] = 0u; }
    if (blockIdx.x == 0) {
        for (int64_t w = 4 * n_quads + threadIdx.x; w < n_words; w += 256) p[w] = 0u;
        uint8_t *b = reinterpret_cast<uint8_t *>(p);
        for (int64_t k = 4 * n_words + threadIdx.x; k < n_bytes; k += 256) b[k] = 0;
    }
}
inline int zero_async(void *p, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return IGCN_OK;
    if (!p) return IGCN_E_NULL;
    if (reinterpret_cast<uintptr_t>(p) & 3) return IGCN_E_ALIGN;
    const int64_t n_words = (int64_t)(bytes / 4);
    int64_t blocks = (n_words / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<uint32_t *>(p), n_words, (int64_t)bytes);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? IGCN_OK : static_cast<int>(e);
}

// Number of CUs of the current device (256 on MI355X); cached.
inline int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Developer / test knobs, set through igcn_set_tuning() (never read from the environment: a getenv() per launch is
// host time on a launch-bound path).  -1 = library default.
enum {
    IGCN_TUNE_SPMM_BLOCKS_PER_CU = 0,   // grid = CUs x this many workgroups (default: sized by rows per wave)
    IGCN_TUNE_SPMM_MULTIROW,            // 0: one row per wave at every width
    IGCN_TUNE_TOPK_SLOTS,               // resident-wave count the top-k plan assumes (tests: cut sweeps at small sizes)
    IGCN_TUNE_TOPK_WAVES_PER_CU,        // 8, 4, 2 or 1
    IGCN_TUNE_TOPK_CAP,                 // staging slots per lane and group
    IGCN_TUNE_TOPK_STAGGER,             // 0: no static wave priorities
    IGCN_TUNE_TOPK_FAST_MODE,           // candidate sweep of the two-stage path: 3 = one fp16 plane each side (default), 2 = two user planes, 1 = two bf16 planes each
    IGCN_TUNE_TOPK_FAST_ORDER,          // 0: the candidate sweep meets the items in id order (default: by descending norm)
    IGCN_TUNE_TOPK_FAST_EXIT,           // 0: the candidate sweep never stops early (default: Cauchy-Schwarz exit in norm order)
    IGCN_TUNE_TOPK_FAST_WIDE,           // d = 128 candidate sweep: 1 = two user groups per wave, one wave per SIMD; 0 = one group, two waves
    IGCN_TUNE_TOPK_FAST_EXTRA,          // candidates the sweep keeps beyond k (default by mode; k + extra <= 64)
    IGCN_TUNE_TOPK_FAST_GIVE_UP,        // 0: no wave of the candidate sweep gives up on its users (default: stragglers hand them to the fp32 sweep)
    IGCN_TUNE_TOPK_FAST_NARROW,         // 0: small batches keep 64-user wave-groups in the candidate sweep (default: 32-user groups)
    IGCN_TUNE_TOPK_FAST_SHARE,          // 0: the pieces of a cut candidate sweep do not share their users' thresholds
    IGCN_TUNE_TOPK_FAST_FALLBACK,       // 0: igcn_score_topk_fast_f32 leaves every flagged user to the caller (ABI v6 behaviour)
    IGCN_TUNE_TOPK_FAST_EARLY_CHECKS,   // 0: the candidate sweep checks for its exit every 24 tiles and gives up from tile 48 on (round 3); default: every 6 tiles up to tile 48, gives up from tile 12
    IGCN_TUNE_TOPK_FAST_WARM,           // whole candidate sweeps: tiles of the warm-up pass that bounds every user's k-th best before the sweep stages anything (0: none)
    IGCN_TUNE_TOPK_FAST_FILTER,         // 0: the flagged users of the two-stage path all take the bounded fp32 sweep (default: a streaming filter first, the sweep for what overflows it)
    IGCN_TUNE_TOPK_FAST_PIECES,         // 0: the narrow bounded sweep is cut into at most 58 pieces per group (default: up to 232, four lists per lane of its merge)
    IGCN_TUNE_SPMM_FOLD,                // 1: cut rows are added up inside the launch by their closing segments (default, unset or 0: by a second kernel, spmm_long_rows_reduce_kernel — faster on the headline graph)
    IGCN_TUNE_TOPK_FAST_POISON,         // TEST ONLY, 1: igcn_score_topk_fast_f32 does NOT clear the order build's counting bins (they keep what the caller's workspace held): the build must notice and fall back to the id order
    IGCN_TUNE_COUNT
};
extern int g_tuning[IGCN_TUNE_COUNT];   // defined in spmm.hip; holds value + 1, 0 = unset
inline int tuning_get(int key) { return g_tuning[key] - 1; }

// 32-bit finaliser (lowbias32-style); used as a counter-based RNG: the value
// depends only on (seed, counter), so a matrix and its transposed view agree.
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
__host__ __device__ __forceinline__ uint32_t hash_counter(uint64_t counter, uint32_t s0, uint32_t s1) {
    uint32_t lo = static_cast<uint32_t>(counter), hi = static_cast<uint32_t>(counter >> 32);
    uint32_t h = mix32(lo ^ s0);
    h = mix32(h + (hi ^ s1) * 0x9e3779b9u + 0x85ebca6bu);
    return h;
}

// An integer the optimiser cannot see through.  Address arithmetic that starts from it stays where it is written: without it the
// per-lane addresses of a kernel's COLD paths (bounds read back, give-up flags, the emit of the scoring sweeps; the fold of a cut row
// in the SpMM) are hoisted out of the outer loop as loop invariants and stay live across the hot loop — in the scoring sweeps they
// ended up in SCRATCH (and a kernel with a private segment is something this library does not ship, see capture_guard), in the SpMM
// they cost a wave per SIMD.  Costs no instruction.
__device__ __forceinline__ int cold(int x) { asm volatile("" : "+v"(x)); return x; }

// a zero row made where it is stored (a cold path: masked rows that must read as zero) — f4_zero() there is hoisted out of the row
// loop as four registers of zeros carried across the gather loop
__device__ __forceinline__ float4 f4_zero_here() { const float z = __int_as_float(cold(0)); return make_float4(z, z, z, z); }
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_fma(float4 &a, float w, const float4 &x) {
    a.x = fmaf(w, x.x, a.x); a.y = fmaf(w, x.y, a.y); a.z = fmaf(w, x.z, a.z); a.w = fmaf(w, x.w, a.w);
}
__device__ __forceinline__ void f4_add(float4 &a, const float4 &x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; }
__device__ __forceinline__ float4 f4_shfl_xor(const float4 &a, int m) {
    return make_float4(__shfl_xor(a.x, m), __shfl_xor(a.y, m), __shfl_xor(a.z, m), __shfl_xor(a.w, m));
}

}  // namespace igcn
