// Probe: do a wave's OWN independent VALU instructions issue in the shadow of its dependent
// v_mfma_f32_32x32x2_f32 chain?  One wave per SIMD; between consecutive (dependent) MFMAs the wave issues
// N independent v_fma_f32.  If the shadow is free, cycles per MFMA stay at 64 until N x ~4 cycles fill it.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, int KIND>
__global__ __launch_bounds__(64) void probe(float *out, int iters, float a0) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = a0;
    float x0 = a0, x1 = a0 + 1, x2 = a0 + 2, x3 = a0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < N; v += 4) {
                if (KIND == 0)
                    asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                else if (KIND == 1)      // integer adds
                    asm volatile("v_add_u32 %0, %0, %0\n\tv_add_u32 %1, %1, %1\n\tv_add_u32 %2, %2, %2\n\tv_add_u32 %3, %3, %3"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                else if (KIND == 2)      // the staging row's mix: fp compare, add with carry, min, shift-add
                    asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\ts_nop 1\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_min_i32 %3, %2, %3\n\tv_lshl_add_u32 %0, %3, 9, %0"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "vcc");
                else                     // scalar ALU
                    asm volatile("s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\ts_add_u32 s22, s22, 1\n\ts_add_u32 s23, s23, 1" : : : "s20", "s21", "s22", "s23");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = x0 + x1 + x2 + x3;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[0] = s;
}

template <int N, int KIND>
void run(float *d) {
    const int iters = 1000, blocks = 1024;          // one single-wave workgroup per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<N, KIND>), dim3(blocks), dim3(64), 0, 0, d, 10, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<N, KIND>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("{\"kind\": \"%s\", \"instructions_between_mfmas\": %d, \"ms\": %.3f, \"ns_per_mfma\": %.2f, \"cycles_per_mfma_at_2p4GHz\": %.1f}\n",
           KIND == 0 ? "v_fma_f32" : KIND == 1 ? "v_add_u32" : KIND == 2 ? "staging row mix (4 VALU + s_nop)" : "s_add_u32", N, ms, ms * 1e6 / (iters * 32.0), ms * 1e6 / (iters * 32.0) * 2.4);
}

int main() {
    float *d; (void)hipMalloc(&d, 4);
    run<0, 0>(d); run<4, 0>(d); run<8, 0>(d); run<12, 0>(d); run<16, 0>(d); run<24, 0>(d);
    run<4, 1>(d); run<8, 1>(d); run<16, 1>(d);
    run<4, 2>(d); run<8, 2>(d); run<16, 2>(d);
    run<4, 3>(d); run<8, 3>(d); run<16, 3>(d);
    return 0;
}
