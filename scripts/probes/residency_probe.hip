// Probe: how many 256-thread workgroups does a CU of this chip really hold at once?
// Every workgroup records when it starts, then spins ~40 us; the count of early starters per launch
// is the true residency (the HIP occupancy query is printed next to it).  Variants differ in the
// registers they pin, to find which resource caps the SpMM kernel at 7 workgroups where the query says 8.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long rt() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

template <int NV, int NS>
__global__ __launch_bounds__(256) void spin(unsigned long long *begin, float *sink, int us) {
    const unsigned long long t0 = rt();
    float v[NV];
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.5f + i;
    int sacc[NS];
    for (int i = 0; i < NS; ++i) sacc[i] = __builtin_amdgcn_readfirstlane(blockIdx.x + i);
    while (rt() - t0 < (unsigned long long)us * 100) {
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + 0.5f;
        for (int i = 0; i < NS; ++i) sacc[i] = sacc[i] * 3 + 1;
    }
    float s = 0.f;
    for (int i = 0; i < NV; ++i) s += v[i];
    int ss = 0;
    for (int i = 0; i < NS; ++i) ss ^= sacc[i];
    if (s == 1234.5f || ss == 12345) sink[0] = s;
    if (threadIdx.x == 0) begin[blockIdx.x] = t0;
}

template <int NV, int NS>
void run(const char *name, unsigned long long *d, float *sink, int cus) {
    const int blocks = cus * 10;
    int api = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, spin<NV, NS>, 256, 0);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)spin<NV, NS>);
    hipLaunchKernelGGL((spin<NV, NS>), dim3(blocks), dim3(256), 0, 0, d, sink, 40);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int early = 0;
    for (auto t : h) early += (t - t0) < 1000;     // started within 10 us
    printf("{\"variant\": \"%s\", \"regs_per_thread\": %d, \"occupancy_api_blocks_per_cu\": %d, \"measured_resident_blocks_per_cu\": %.2f}\n",
           name, fa.numRegs, api, (double)early / cus);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, cus * 10 * 8); (void)hipMalloc(&sink, 4);
    run<4, 4>("tiny", d, sink, cus);
    run<40, 4>("40 vgpr", d, sink, cus);
    run<52, 4>("52 vgpr", d, sink, cus);
    run<56, 40>("56 vgpr + 40 sgpr", d, sink, cus);
    run<56, 70>("56 vgpr + 70 sgpr", d, sink, cus);
    run<72, 4>("72 vgpr", d, sink, cus);
    run<100, 4>("100 vgpr", d, sink, cus);
    return 0;
}
