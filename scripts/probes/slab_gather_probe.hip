// Probe (developer only, not shipped): would a COLUMN-SLAB layout of the operand make every gather an L2 hit?
//   X as S = 8 slabs [slab][row][8 floats]; workgroup b works for slab b % 8 (= its XCD), so an XCD's L2 only ever sees
//   1/8 of the table (3.5 MB of the 28 MB user table) — but every XCD reads the whole index stream (8 x 8 B per
//   nonzero) and a gather instruction touches 32 pieces of 32 B instead of 4 rows of 256 B.
// Rowless like igcn_roof_gather_f32: 64-entry chunks of the index stream, dealt to the waves of a slab.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kWave = 64, kBlock = 256;

// PIECE floats per slab row (8: 32 B, 16: 64 B); S = d / PIECE slabs; workgroup b -> slab b % S
template <int PIECE>
__global__ __launch_bounds__(kBlock) void slab_gather_kernel(const int32_t *__restrict__ idx, const float *__restrict__ val, int64_t n_idx,
                                                             const float *__restrict__ x, int64_t n_rows, int S,
                                                             float *__restrict__ y, int64_t n_out, double rows_per_idx)
{
    constexpr int LPR = PIECE / 4, G = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / LPR, t = lane % LPR;
    const int slab = blockIdx.x % S;
    const float *xs = x + (int64_t)slab * n_rows * PIECE;
    float *ys = y + (int64_t)slab * n_out * PIECE;
    const int64_t wave0 = (int64_t)(blockIdx.x / S) * (kBlock / kWave) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)(gridDim.x / S) * (kBlock / kWave);
    const int64_t n_chunks = (n_idx + kWave - 1) / kWave;
    for (int64_t ch = wave0; ch < n_chunks; ch += n_waves) {
        const int64_t base = ch * kWave;
        const int64_t rem = n_idx - base;
        const int cnt = rem < kWave ? (int)rem : kWave;
        int c = 0;
        float w = 0.f;
        if (lane < cnt) { c = idx[base + lane]; w = val[base + lane]; }   // (entries past the end: weight 0, row 0)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < kWave; k += G) {
            const int cc = __shfl(c, k + g);
            const float ww = __shfl(w, k + g);
            const float4 xv = *reinterpret_cast<const float4 *>(xs + (int64_t)cc * PIECE + 4 * t);
            acc.x = fmaf(ww, xv.x, acc.x); acc.y = fmaf(ww, xv.y, acc.y); acc.z = fmaf(ww, xv.z, acc.z); acc.w = fmaf(ww, xv.w, acc.w);
        }
        if (rows_per_idx < 0.) {
            // checking mode: the chunk's sum over ALL groups -> row `ch` of the slab's output (y has n_chunks rows)
#pragma unroll
            for (int m = LPR; m < kWave; m <<= 1) {
                acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m); acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
            }
            if (g == 0) *reinterpret_cast<float4 *>(ys + ch * PIECE + 4 * t) = acc;
            continue;
        }
        const int64_t r0 = (int64_t)((double)base * rows_per_idx);
        int64_t r1 = (int64_t)((double)(base + cnt) * rows_per_idx);
        if (r1 > n_out) r1 = n_out;
        for (int64_t r = r0 + g; r < r1; r += G)
            *reinterpret_cast<float4 *>(ys + r * PIECE + 4 * t) = acc;
    }
}

extern "C" int slab_gather(const int32_t *idx, const float *val, int64_t n_idx, const float *x, int64_t n_rows, int d, int piece,
                           float *y, int64_t n_out, int64_t blocks, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int S = d / piece;
    blocks = blocks / S * S;
    const dim3 grid((unsigned)blocks), block(kBlock);
    const double rpi = n_out < 0 ? -1. : (double)n_out / (double)n_idx;
    if (n_out < 0) n_out = (n_idx + kWave - 1) / kWave;
    if (piece == 8) hipLaunchKernelGGL(slab_gather_kernel<8>, grid, block, 0, st, idx, val, n_idx, x, n_rows, S, y, n_out, rpi);
    else if (piece == 16) hipLaunchKernelGGL(slab_gather_kernel<16>, grid, block, 0, st, idx, val, n_idx, x, n_rows, S, y, n_out, rpi);
    else if (piece == 32) hipLaunchKernelGGL(slab_gather_kernel<32>, grid, block, 0, st, idx, val, n_idx, x, n_rows, S, y, n_out, rpi);
    else if (piece == 64) hipLaunchKernelGGL(slab_gather_kernel<64>, grid, block, 0, st, idx, val, n_idx, x, n_rows, S, y, n_out, rpi);
    else return -1;
    return (int)hipGetLastError();
}
