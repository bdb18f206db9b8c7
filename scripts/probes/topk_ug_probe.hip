// Probe: user groups per wave (UG = 1 or 2: 32 or 64 users share one item-tile load) x waves per SIMD,
// fragment-shaped global loads of the item tile, selection pass on every tile (worst case).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <int UG, int WPS, int EPI>
__global__ __launch_bounds__(64, WPS) void sweep(const float *__restrict__ users, const float *__restrict__ items, int n_items,
                                                 float *out, float thr0) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    float b[UG][32];
#pragma unroll
    for (int g = 0; g < UG; ++g)
        for (int q = 0; q < 8; ++q) {
            float4 v = *reinterpret_cast<const float4 *>(users + (((size_t)blockIdx.x * UG + g) * 32 + j) * 64 + 8 * q + 4 * h);
            b[g][4 * q] = v.x; b[g][4 * q + 1] = v.y; b[g][4 * q + 2] = v.z; b[g][4 * q + 3] = v.w;
        }
    float thr = thr0, best = 0.f;
    int cnt = 0;
    for (int t = 0; t < n_items; t += 32) {
        float4 a[8];
        const float *p = items + (size_t)(t + j) * 64 + 4 * h;
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const float4 *>(p + 8 * q);
        f32x16 acc[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int g = 0; g < UG; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[g][4 * q], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[g][4 * q + 1], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[g][4 * q + 2], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[g][4 * q + 3], acc[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            float m = acc[g][0];
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[g][r]);
            bool go = __any(m >= thr);
            if (EPI == 2) go = true;
            if (go) {
                float cs = 0.f; int cr = -1;
                for (int r = 0; r < 16; ++r) {
                    const float s = acc[g][r];
                    const bool take = cr < 0 && s >= thr;
                    cs = take ? s : cs; cr = take ? r : cr;
                }
                if (cr >= 0) { best = fmaxf(best, cs); ++cnt; }
            }
        }
    }
    if (best == 12345.678f || cnt == 123456789) out[0] = best;
}

template <int UG, int WPS, int EPI>
int run(const float *u, const float *it, int n_items, float *out, int lds) {
    const int n_blocks = 1024 * WPS;                      // all resident
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((sweep<UG, WPS, EPI>), dim3(n_blocks), dim3(64), lds, 0, u, it, 3200, out, 1e30f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((sweep<UG, WPS, EPI>), dim3(n_blocks), dim3(64), lds, 0, u, it, n_items, out, 1e30f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double fl = 2.0 * n_blocks * UG * 32 * (double)n_items * 64;
    printf("{\"ug\": %d, \"waves_per_simd\": %d, \"epi\": %d, \"ms\": %.3f, \"tflops\": %.1f}\n", UG, WPS, EPI, ms, fl / ms / 1e9);
    return 0;
}

int main() {
    const int n_items = 96416, n_users = 8192 * 32;
    float *u, *it, *out;
    CK(hipMalloc(&u, (size_t)n_users * 64 * 4)); CK(hipMalloc(&it, (size_t)(n_items + 64) * 64 * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemset(u, 0, (size_t)n_users * 64 * 4)); CK(hipMemset(it, 0, (size_t)(n_items + 64) * 64 * 4));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<1, 4, 1>(u, it, n_items, out, 10240)) return 1;
        if (run<1, 4, 2>(u, it, n_items, out, 10240)) return 1;
        if (run<2, 2, 1>(u, it, n_items, out, 20480)) return 1;
        if (run<2, 2, 2>(u, it, n_items, out, 20480)) return 1;
        if (run<1, 2, 2>(u, it, n_items, out, 20480)) return 1;
        if (run<2, 1, 2>(u, it, n_items, out, 40960)) return 1;
    }
    return 0;
}
