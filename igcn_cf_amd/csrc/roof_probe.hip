// Measurement kernels for bench.py (libigcn_roof.so) — NOT part of the product library.
//
// igcn_roof_gather_f32: the roof of a CSR SpMM whose operand X is cache resident, measured on the
// box the bench runs on.  It does everything an SpMM must do per nonzero — read a column id and a
// value (8 B, streamed), gather the d*4-byte source row X[col], one FMA per element — and writes
// one output row per `nnz_per_row` nonzeros, but it knows nothing of rows: the index stream is cut
// into 64-entry chunks dealt round-robin to the waves, so every wave does the same work, there is
// no row-pointer chain, no short-row tail, no long-row pass.  A real SpMM over the same index
// stream cannot be faster than this; how close it comes is roofline.frac (bench.py).
// Called with uniformly random indices it measures what the guide calls the rate of random-row
// gathers from a cache-resident table (MI355X_MICROARCH.md, "Indexed rows").
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace roof {

constexpr int kWave = 64;
constexpr int kBlock = 256;

template <int LPR>
__global__ __launch_bounds__(kBlock) void gather_roof_kernel(
    const int32_t *__restrict__ idx, const float *__restrict__ val, int64_t n_idx,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy, int64_t n_out, double rows_per_idx)
{
    constexpr int G = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / LPR, t = lane % LPR;
    const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave);
    const int64_t n_chunks = (n_idx + kWave - 1) / kWave;
    for (int64_t ch = wave0; ch < n_chunks; ch += n_waves) {
        const int64_t base = ch * kWave;
        const int64_t rem = n_idx - base;
        const int cnt = rem < kWave ? (int)rem : kWave;
        int c = 0;
        float w = 0.f;
        if (lane < cnt) { c = idx[base + lane]; w = val[base + lane]; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 4 * G <= cnt; k += 4 * G) {
            const int c0 = __shfl(c, k + g), c1 = __shfl(c, k + G + g), c2 = __shfl(c, k + 2 * G + g), c3 = __shfl(c, k + 3 * G + g);
            const float w0 = __shfl(w, k + g), w1 = __shfl(w, k + G + g), w2 = __shfl(w, k + 2 * G + g), w3 = __shfl(w, k + 3 * G + g);
            const float4 x0 = *reinterpret_cast<const float4 *>(x + (int64_t)c0 * ldx + 4 * t);
            const float4 x1 = *reinterpret_cast<const float4 *>(x + (int64_t)c1 * ldx + 4 * t);
            const float4 x2 = *reinterpret_cast<const float4 *>(x + (int64_t)c2 * ldx + 4 * t);
            const float4 x3 = *reinterpret_cast<const float4 *>(x + (int64_t)c3 * ldx + 4 * t);
            acc.x = fmaf(w0, x0.x, acc.x); acc.y = fmaf(w0, x0.y, acc.y); acc.z = fmaf(w0, x0.z, acc.z); acc.w = fmaf(w0, x0.w, acc.w);
            acc.x = fmaf(w1, x1.x, acc.x); acc.y = fmaf(w1, x1.y, acc.y); acc.z = fmaf(w1, x1.z, acc.z); acc.w = fmaf(w1, x1.w, acc.w);
            acc.x = fmaf(w2, x2.x, acc.x); acc.y = fmaf(w2, x2.y, acc.y); acc.z = fmaf(w2, x2.z, acc.z); acc.w = fmaf(w2, x2.w, acc.w);
            acc.x = fmaf(w3, x3.x, acc.x); acc.y = fmaf(w3, x3.y, acc.y); acc.z = fmaf(w3, x3.z, acc.z); acc.w = fmaf(w3, x3.w, acc.w);
        }
        for (; k < cnt; k += G) {
            const int src = k + g;
            const int cc = __shfl(c, src);
            const float ww = __shfl(w, src);
            if (src < cnt) {
                const float4 xv = *reinterpret_cast<const float4 *>(x + (int64_t)cc * ldx + 4 * t);
                acc.x = fmaf(ww, xv.x, acc.x); acc.y = fmaf(ww, xv.y, acc.y); acc.z = fmaf(ww, xv.z, acc.z); acc.w = fmaf(ww, xv.w, acc.w);
            }
        }
        // the output rows this chunk's share of the index stream stands for (n_out / n_idx rows per index);
        // groups take them in turn (no fold: any sum will do)
        const int64_t r0 = (int64_t)((double)base * rows_per_idx);
        int64_t r1 = (int64_t)((double)(base + cnt) * rows_per_idx);
        if (r1 > n_out) r1 = n_out;
        for (int64_t r = r0 + g; r < r1; r += G)
            *reinterpret_cast<float4 *>(y + r * ldy + 4 * t) = acc;
    }
}

// What the memory system of THIS box streams (the ceiling every HBM-bound figure of bench.py is set against, next to the
// 8 TB/s spec): every lane moves 16 bytes per instruction, four independent loads in flight per lane before the first use,
// a grid-stride sweep over a buffer far larger than the 256 MiB Infinity Cache.
//   MODE 0  read only  — the loads are folded into one float per lane that is stored only if it equals a magic value
//                        (never), so the compiler must keep the loads and nothing is written;
//   MODE 1  copy       — dst[i] = src[i]  (read + write bytes).
typedef float f32x4 __attribute__((ext_vector_type(4)));      // the nontemporal builtins take native vectors, not HIP_vector_type
template <int MODE, int U, bool NT>
__global__ __launch_bounds__(kBlock) void stream_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, int64_t n4, float magic)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float acc = 0.f;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 1) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
            else acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        }
    }
    for (; i < n4; i += stride) {
        const f32x4 a = src[i];
        if (MODE == 1) dst[i] = a;
        else acc += a.x + a.y + a.z + a.w;
    }
    if (MODE == 0 && acc == magic) dst[0] = f32x4{acc, acc, acc, acc};
}

}  // namespace roof
using namespace roof;

// src, dst: n4 float4 each (16-byte aligned); mode 0 = read only (dst: any valid float4, never written unless the sum of
// the buffer equals 1.2345e33), 1 = copy.  variant: bit 0 = eight loads in flight per lane instead of four, bit 1 =
// non-temporal loads / stores.  blocks: workgroups of 256 threads.  Returns 0 or a hipError_t / -1.
extern "C" int igcn_roof_stream_f32(const void *src, void *dst, int64_t n4, int32_t mode, int32_t variant, int64_t blocks, void *stream)
{
    if (!src || !dst || n4 < 1 || blocks < 1 || blocks >= ((int64_t)1 << 31) || (mode != 0 && mode != 1) || variant < 0 || variant > 3) return -1;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)blocks), block(kBlock);
    const f32x4 *s4 = static_cast<const f32x4 *>(src);
    f32x4 *d4 = static_cast<f32x4 *>(dst);
    const float magic = mode == 0 ? 1.2345e33f : 0.f;
#define IGCN_STREAM(M, U, NT) hipLaunchKernelGGL((stream_kernel<M, U, NT>), grid, block, 0, st, s4, d4, n4, magic)
    switch (mode * 4 + variant) {
    case 0: IGCN_STREAM(0, 4, false); break;
    case 1: IGCN_STREAM(0, 8, false); break;
    case 2: IGCN_STREAM(0, 4, true); break;
    case 3: IGCN_STREAM(0, 8, true); break;
    case 4: IGCN_STREAM(1, 4, false); break;
    case 5: IGCN_STREAM(1, 8, false); break;
    case 6: IGCN_STREAM(1, 4, true); break;
    default: IGCN_STREAM(1, 8, true); break;
    }
#undef IGCN_STREAM
    return (int)hipGetLastError();
}
// idx [n_idx] int32 in [0, n_x_rows), val [n_idx], x [n_x_rows, d] (ldx), y [n_out, d] (ldy); d in {16, 32, 64, 128, 256}.
// blocks: workgroups of 256 threads to launch.  Returns 0 or a hipError_t / -1 on a bad argument.
extern "C" int igcn_roof_gather_f32(const int32_t *idx, const float *val, int64_t n_idx, const float *x, int64_t ldx,
                                    float *y, int64_t ldy, int64_t n_out, int32_t d, int64_t blocks, void *stream)
{
    if (!idx || !val || !x || !y || n_idx < 1 || n_out < 1 || blocks < 1 || blocks >= ((int64_t)1 << 31)) return -1;
    if (ldx < d || ldy < d || ldx % 4 || ldy % 4) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)blocks), block(kBlock);
    const double rpi = (double)n_out / (double)n_idx;
    switch (d) {
    case 16: hipLaunchKernelGGL(gather_roof_kernel<4>, grid, block, 0, st, idx, val, n_idx, x, ldx, y, ldy, n_out, rpi); break;
    case 32: hipLaunchKernelGGL(gather_roof_kernel<8>, grid, block, 0, st, idx, val, n_idx, x, ldx, y, ldy, n_out, rpi); break;
    case 64: hipLaunchKernelGGL(gather_roof_kernel<16>, grid, block, 0, st, idx, val, n_idx, x, ldx, y, ldy, n_out, rpi); break;
    case 128: hipLaunchKernelGGL(gather_roof_kernel<32>, grid, block, 0, st, idx, val, n_idx, x, ldx, y, ldy, n_out, rpi); break;
    case 256: hipLaunchKernelGGL(gather_roof_kernel<64>, grid, block, 0, st, idx, val, n_idx, x, ldx, y, ldy, n_out, rpi); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}

