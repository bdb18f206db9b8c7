// Probe: does a wave's dependent v_mfma_f32_32x32x2_f32 chain starve the VALU instructions of
// the OTHER waves on the same SIMD?  512-thread blocks (8 waves, 2 per SIMD), one block per CU:
// waves 0-3 run MFMA chains, waves 4-7 run a plain VALU stream; each records its own s_memtime span.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned long long clk() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

// mode bit0: MFMA waves active; bit1: VALU waves active; chains: independent accumulators per MFMA wave
template <int CHAINS>
__global__ __launch_bounds__(1024) void probe(unsigned long long *out, int mfma_iters, int valu_iters, int mode, float a0, int mfma_waves = 4) {
    extern __shared__ char hog[];
    const int wave = threadIdx.x >> 6;
    const unsigned hw_simd = (__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) >> 4) & 3u;
    const bool is_mfma = wave < mfma_waves;          // waves are dealt to the 4 SIMDs round-robin: mfma_waves / 4 chains per SIMD
    unsigned long long t0 = clk();
    if (is_mfma) {
        if (!(mode & 1)) return;
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        float a = a0 + threadIdx.x * 1e-6f, b = a0;
        for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
            for (int q = 0; q < 32 / CHAINS; ++q)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
        unsigned long long t1 = clk();
        if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + wave] = (((t1 - t0) + (s == 12345.678f)) << 2) | hw_simd;
    } else {
        if (!(mode & 2)) return;
        float x0 = a0, x1 = a0 + 1, x2 = a0 + 2, x3 = a0 + 3, x4 = a0 + 4, x5 = a0 + 5, x6 = a0 + 6, x7 = a0 + 7;
        for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                             "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
        unsigned long long t1 = clk();
        float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + wave] = (((t1 - t0) + (s == 12345.678f)) << 2) | hw_simd;
    }
}

template <int CHAINS>
void run(const char *name, int mode, int mfma_iters, int valu_iters, unsigned long long *d) {
    const int blocks = 256;
    hipMemset(d, 0, blocks * 16 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)probe<CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(probe<CHAINS>, dim3(blocks), dim3(512), 100 * 1024, 0, d, 10, 10, mode, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<CHAINS>, dim3(blocks), dim3(512), 100 * 1024, 0, d, mfma_iters, valu_iters, mode, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 16);
    hipMemcpy(h.data(), d, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double tm = 0, tv = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? tm : tv) += (double)(h[b * 16 + w] >> 2);
    tm /= blocks * 4; tv /= blocks * 4;
    printf("{\"case\": \"%s\", \"chains\": %d, \"ms\": %.3f, \"mfma_ticks_per_mfma\": %.1f, \"valu_ticks_per_inst\": %.2f}\n",
           name, CHAINS, ms, tm / ((double)mfma_iters * 32), tv / ((double)valu_iters * 64));
}

void run_multi(int mfma_per_simd, int mfma_iters, int valu_iters, unsigned long long *d) {
    const int blocks = 256, waves = 4 * (mfma_per_simd + 1);
    (void)hipMemset(d, 0, blocks * 16 * 8);
    (void)hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64 * waves), 100 * 1024, 0, d, mfma_iters, valu_iters, 3, 1.f, 4 * mfma_per_simd);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 16);
    (void)hipMemcpy(h.data(), d, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double tm = 0, tv = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) (w < 4 * mfma_per_simd ? tm : tv) += (double)(h[b * 16 + w] >> 2);
    tm /= blocks * 4 * mfma_per_simd; tv /= blocks * 4;
    printf("{\"simd_of_waves_block0\": [");
    for (int w = 0; w < waves; ++w) printf("%d%s", (int)(h[w] & 3), w + 1 < waves ? ", " : "]}\n");
    printf("{\"case\": \"%d mfma-chain waves + 1 valu wave per SIMD\", \"ms\": %.3f, \"mfma_ticks_per_mfma_per_wave\": %.1f, \"valu_ticks_per_inst\": %.2f}\n",
           mfma_per_simd, ms, tm / ((double)mfma_iters * 32), tv / ((double)valu_iters * 64));
}

int main() {
    unsigned long long *d; hipMalloc(&d, 256 * 16 * 8);
    run<1>("mfma_only", 1, 4000, 0, d);
    run<1>("valu_only", 2, 0, 20000, d);
    run<1>("both_dependent_chain", 3, 4000, 20000, d);     // VALU stream ends first (if unhindered)
    run<2>("both_two_chains", 3, 4000, 20000, d);
    run<4>("both_four_chains", 3, 4000, 20000, d);
    run<1>("both_long_valu", 3, 2000, 200000, d);           // VALU stream outlasts the MFMA waves
    run_multi(1, 4000, 10000, d);
    run_multi(2, 4000, 10000, d);
    run_multi(3, 4000, 10000, d);
    return 0;
}
