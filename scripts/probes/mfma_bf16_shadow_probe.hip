// Probe: rate of v_mfma_f32_32x32x16_bf16 (dependent chain, one wave per SIMD) and whether a wave's own
// independent v_fma_f32 instructions issue in ITS shadow (they do not in the shadow of the fp32 MFMA).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, int CHAINS>
__global__ __launch_bounds__(64) void probe(float *out, int iters, float a0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + threadIdx.x * 1e-3f + i); b[i] = (__bf16)(a0 + i); }
    float x0 = a0, x1 = a0 + 1, x2 = a0 + 2, x3 = a0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 32 / CHAINS; ++q) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < N; v += 4)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = x0 + x1 + x2 + x3;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

template <int N, int CHAINS>
void run(float *d, int waves_per_simd) {
    const int iters = 1000, blocks = 1024 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<N, CHAINS>), dim3(blocks), dim3(64), 0, 0, d, 10, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<N, CHAINS>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)iters * 32 * waves_per_simd;
    printf("{\"chains\": %d, \"waves_per_simd\": %d, \"valu_after_each_mfma_group\": %d, \"ms\": %.3f, \"cycles_per_mfma_per_simd_at_2p4GHz\": %.1f, \"tflops\": %.0f}\n",
           CHAINS, waves_per_simd, N, ms, ms * 1e6 / mfmas_per_simd * 2.4, mfmas_per_simd * 1024 * 32.0 * 32 * 16 * 2 / ms / 1e9);
}

int main() {
    float *d; (void)hipMalloc(&d, 4);
    run<0, 1>(d, 1); run<0, 2>(d, 1); run<0, 4>(d, 1); run<0, 1>(d, 2); run<0, 1>(d, 4);
    run<4, 1>(d, 1); run<8, 1>(d, 1); run<4, 2>(d, 1); run<8, 2>(d, 1); run<8, 4>(d, 1); run<16, 4>(d, 1);
    return 0;
}
