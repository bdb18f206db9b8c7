// Device-side BPR negative sampler for MI355X (gfx950).
//
// Replaces the six DataLoader worker processes running BasicDataset.__getitem__
// (dataset.py:119-131, trainer.py:226-227): per draw, a uniform user with a
// non-empty train list, a uniform positive from that list and a uniform item
// rejected while it is in the list.  One thread per draw; the membership test
// is a binary search in the user's sorted CSR row; randomness is a
// counter-based hash of (seed, draw, attempt), so a batch is reproducible from
// its seed alone.  Parity with the reference is statistical (the reference's
// own draws come from unseeded worker processes).
#include "common.h"

namespace igcn {

__device__ __forceinline__ int64_t uniform_below(uint64_t counter, uint32_t s0, uint32_t s1, int64_t n) {
    const uint64_t r = ((uint64_t)hash_counter(counter, s0, s1) << 32) | hash_counter(counter ^ 0x9e3779b97f4a7c15ull, s1, s0);
    return (int64_t)__umul64hi(r, (uint64_t)n);       // floor(r * n / 2^64): unbiased to 2^-64 * n
}

__global__ void bpr_sample_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                  const int32_t *__restrict__ nonempty, int64_t n_nonempty, int64_t n_items,
                                  int64_t batch, uint32_t s0, uint32_t s1, int64_t *__restrict__ out, int64_t soa_item_offset)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const uint64_t base = (uint64_t)i << 12;          // 4096 counters per draw
    const int64_t user = nonempty[uniform_below(base, s0, s1, n_nonempty)];
    const int64_t lo0 = rowptr[user], hi0 = rowptr[user + 1];
    const int64_t pos = col[lo0 + uniform_below(base + 1, s0, s1, hi0 - lo0)];
    int64_t neg = 0;
    for (int attempt = 0; attempt < 4000; ++attempt) {  // bounded: every thread terminates
        neg = uniform_below(base + 2 + attempt, s0, s1, n_items);
        int64_t lo = lo0, hi = hi0;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (col[mid] < neg) lo = mid + 1; else hi = mid;
        }
        if (!(lo < hi0 && col[lo] == neg)) break;
    }
    if (soa_item_offset < 0) {                         // [batch, 3] triplets (what a DataLoader batch looks like)
        out[3 * i + 0] = user;
        out[3 * i + 1] = pos;
        out[3 * i + 2] = neg;
    } else {                                           // [3, batch] node ids: users, offset + positives, offset + negatives
        out[i] = user;
        out[batch + i] = soa_item_offset + pos;
        out[2 * batch + i] = soa_item_offset + neg;
    }
}

}  // namespace igcn

using namespace igcn;

extern "C" int igcn_bpr_sample(const int64_t *train_rowptr, const int32_t *train_col,
                               const int32_t *nonempty_users, int64_t n_nonempty, int64_t n_items,
                               int64_t batch, uint64_t seed, int64_t *out, void *stream)
{
    if (!train_rowptr || !train_col || !nonempty_users || !out) return IGCN_E_NULL;
    if (n_nonempty < 1 || n_items < 1 || batch < 0) return IGCN_E_SHAPE;
    if (batch == 0) return IGCN_OK;
    hipLaunchKernelGGL(bpr_sample_kernel, dim3((unsigned)((batch + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), train_rowptr, train_col, nonempty_users, n_nonempty, n_items,
                       batch, (uint32_t)seed, (uint32_t)(seed >> 32), out, (int64_t)-1);
    return launch_status();
}

extern "C" int igcn_bpr_sample_nodes(const int64_t *train_rowptr, const int32_t *train_col,
                                     const int32_t *nonempty_users, int64_t n_nonempty, int64_t n_items,
                                     int64_t batch, uint64_t seed, int64_t item_offset, int64_t *out, void *stream)
{
    if (!train_rowptr || !train_col || !nonempty_users || !out) return IGCN_E_NULL;
    if (n_nonempty < 1 || n_items < 1 || batch < 0 || item_offset < 0) return IGCN_E_SHAPE;
    if (batch == 0) return IGCN_OK;
    hipLaunchKernelGGL(bpr_sample_kernel, dim3((unsigned)((batch + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), train_rowptr, train_col, nonempty_users, n_nonempty, n_items,
                       batch, (uint32_t)seed, (uint32_t)(seed >> 32), out, item_offset);
    return launch_status();
}
