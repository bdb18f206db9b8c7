// Probe: sustained rate of v_mfma_f32_32x32x2_f32 with ONE dependent accumulator chain per
// wave, for 1..4 waves per SIMD (and 2 chains per wave for comparison).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(64) void probe(float *out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
void run(int waves_per_simd, float *d) {
    int iters = 2000;
    int blocks = 256 * 4 * waves_per_simd;          // single-wave blocks
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)blocks * iters * 32 * CHAINS;
    double tflops = mfmas * 32 * 32 * 2 * 2 / ms / 1e9;
    printf("{\"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"ns_per_mfma_per_simd\": %.2f}\n",
           CHAINS, waves_per_simd, ms, tflops, ms * 1e6 / (mfmas / 1024));
}

int main() {
    float *d; hipMalloc(&d, 4);
    for (int w = 1; w <= 4; ++w) run<1>(w, d);
    for (int w = 1; w <= 2; ++w) run<2>(w, d);
    run<4>(1, d);
    return 0;
}
