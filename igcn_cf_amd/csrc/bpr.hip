// Fused BPR triplet scoring for MI355X (gfx950): gather + dot + softplus + L2,
// forward and backward.  Replaces trainer.py:238-243 / :306-311 and the row
// gathers + squared norms of model.py:110-116, :295-299, :62-67.
//
// This is 3*B row gathers of d floats (B = 2048, d = 64: 1.5 MB) — latency
// bound, nowhere near a GEMM, so no MFMA: one wave per triplet, each row read
// as one coalesced d*4-byte request, wave-level shuffles for the dots, a fixed
// order second stage for the two batch means (bitwise reproducible), and in
// the backward pass one 256-byte float-atomic row segment per gradient row
// (the shape the memory-side atomic units run at full rate).
#include "common.h"

namespace igcn {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(kBlock) void bpr_fwd_kernel(
    const float *__restrict__ u_tab, const float *__restrict__ p_tab, const float *__restrict__ n_tab, int64_t ld,
    const float *__restrict__ l2u, const float *__restrict__ l2p, const float *__restrict__ l2n, int64_t ld2,
    const int64_t *__restrict__ users, const int64_t *__restrict__ pos, const int64_t *__restrict__ neg,
    int64_t batch, int d, const float *__restrict__ w, float *__restrict__ work)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t b = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (b >= batch) return;
    const int64_t iu = users[b], ip = pos[b], in = neg[b];
    float ps = 0.f, ns = 0.f, l2 = 0.f;
    for (int j = lane; j < d; j += kWave) {
        const float u = u_tab[iu * ld + j], p = p_tab[ip * ld + j], n = n_tab[in * ld + j];
        const float uw = w ? u * w[j] : u;
        ps = fmaf(uw, p, ps);
        ns = fmaf(uw, n, ns);
        if (l2u) {
            const float a = l2u[iu * ld2 + j], c = l2p[ip * ld2 + j], e = l2n[in * ld2 + j];
            l2 = fmaf(a, a, l2); l2 = fmaf(c, c, l2); l2 = fmaf(e, e, l2);
        }
    }
    ps = wave_sum(ps); ns = wave_sum(ns); l2 = wave_sum(l2);
    if (lane == 0) {
        const float x = ns - ps;
        // softplus(x), beta=1, threshold=20 (torch.nn.functional.softplus)
        const float sp = x > 20.f ? x : log1pf(expf(x));
        work[b] = 1.f / (1.f + expf(-x));
        work[batch + b] = sp;
        work[2 * batch + b] = l2;
    }
}

// Partial dots of one embedding-column slice (column-sharded multi-GPU path): work = [pos | neg | l2],
// to be summed over the slices (all-reduce) and finished by bpr_finish_kernel.
__global__ __launch_bounds__(kBlock) void bpr_dots_kernel(
    const float *__restrict__ u_tab, const float *__restrict__ p_tab, const float *__restrict__ n_tab, int64_t ld,
    const float *__restrict__ l2u, const float *__restrict__ l2p, const float *__restrict__ l2n, int64_t ld2,
    const int64_t *__restrict__ users, const int64_t *__restrict__ pos, const int64_t *__restrict__ neg,
    int64_t batch, int d, const float *__restrict__ w, float *__restrict__ work)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t b = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (b >= batch) return;
    const int64_t iu = users[b], ip = pos[b], in = neg[b];
    float ps = 0.f, ns = 0.f, l2 = 0.f;
    for (int j = lane; j < d; j += kWave) {
        const float u = u_tab[iu * ld + j], p = p_tab[ip * ld + j], n = n_tab[in * ld + j];
        const float uw = w ? u * w[j] : u;
        ps = fmaf(uw, p, ps);
        ns = fmaf(uw, n, ns);
        if (l2u) {
            const float a = l2u[iu * ld2 + j], c = l2p[ip * ld2 + j], e = l2n[in * ld2 + j];
            l2 = fmaf(a, a, l2); l2 = fmaf(c, c, l2); l2 = fmaf(e, e, l2);
        }
    }
    ps = wave_sum(ps); ns = wave_sum(ns); l2 = wave_sum(l2);
    if (lane == 0) { work[b] = ps; work[batch + b] = ns; work[2 * batch + b] = l2; }
}

// dots [pos | neg | l2] (complete sums) -> work [sigmoid | softplus | l2], the layout of bpr_fwd_kernel
__global__ void bpr_finish_kernel(const float *__restrict__ dots, int64_t batch, float *__restrict__ work)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const float x = dots[batch + b] - dots[b];
    work[b] = 1.f / (1.f + expf(-x));
    work[batch + b] = x > 20.f ? x : log1pf(expf(x));
    work[2 * batch + b] = dots[2 * batch + b];
}

// loss_out[0] = mean(work[B..2B)), loss_out[1] = mean(work[2B..3B)); one block, fixed order.
__global__ __launch_bounds__(kBlock) void bpr_reduce_kernel(const float *__restrict__ work, int64_t batch,
                                                            float *__restrict__ loss_out, float l2_weight = 0.f, int write3 = 0)
{
    __shared__ float sm[2][kBlock];
    float a = 0.f, c = 0.f;
    for (int64_t i = threadIdx.x; i < batch; i += kBlock) { a += work[batch + i]; c += work[2 * batch + i]; }
    sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = c;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sm[0][threadIdx.x] += sm[0][threadIdx.x + s]; sm[1][threadIdx.x] += sm[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float a0 = sm[0][0] / (float)batch, a1 = sm[1][0] / (float)batch;
        loss_out[0] = a0; loss_out[1] = a1;
        if (write3) loss_out[2] = a0 + l2_weight * a1;       // the training loss of trainer.py:242 in the same launch
    }
}

__global__ __launch_bounds__(kBlock) void bpr_bwd_kernel(
    const float *__restrict__ u_tab, const float *__restrict__ p_tab, const float *__restrict__ n_tab, int64_t ld,
    const float *__restrict__ l2u, const float *__restrict__ l2p, const float *__restrict__ l2n, int64_t ld2,
    const int64_t *__restrict__ users, const int64_t *__restrict__ pos, const int64_t *__restrict__ neg,
    int64_t batch, int d, const float *__restrict__ w, const float *__restrict__ work, const float *__restrict__ g_out,
    float *__restrict__ gu, float *__restrict__ gp, float *__restrict__ gn,
    float *__restrict__ g2u, float *__restrict__ g2p, float *__restrict__ g2n, float *__restrict__ gw,
    int g_len = 2, float s0 = 1.f, float s1 = 1.f)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave);
    const float inv_b = 1.f / (float)batch;
    // d loss / d w: every triplet adds to the same d floats — kept in registers across the wave's triplets (the launch
    // uses a small grid then) and added once per wave (columns lane, lane + 64, ... up to 256; beyond: straight atomics)
    float gw_acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t b = wave; b < batch; b += n_waves) {
        const int64_t iu = users[b], ip = pos[b], in = neg[b];
        const float c = g_out[0] * s0 * work[b] * inv_b;  // d loss / d (neg_b - pos_b)
        const float c2 = 2.f * g_out[g_len > 1 ? 1 : 0] * s1 * inv_b;
        for (int j = lane, q = 0; j < d; j += kWave, ++q) {
            const float u = u_tab[iu * ld + j], p = p_tab[ip * ld + j], n = n_tab[in * ld + j];
            const float wj = w ? w[j] : 1.f;
            atomicAdd(gu + iu * ld + j, c * (n - p) * wj);
            atomicAdd(gp + ip * ld + j, -c * u * wj);
            atomicAdd(gn + in * ld + j, c * u * wj);
            if (gw) {
                if (q < 4) gw_acc[q] += c * u * (n - p);
                else atomicAdd(gw + j, c * u * (n - p));
            }
            if (l2u && g2u) {
                atomicAdd(g2u + iu * ld2 + j, c2 * l2u[iu * ld2 + j]);
                atomicAdd(g2p + ip * ld2 + j, c2 * l2p[ip * ld2 + j]);
                atomicAdd(g2n + in * ld2 + j, c2 * l2n[in * ld2 + j]);
            }
        }
    }
    if (gw)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lane + q * kWave < d) atomicAdd(gw + lane + q * kWave, gw_acc[q]);
}

// The row-sparse tail of a training step's backward pass, one wave per id: dst[ids[i]] += scale * src[ids[i]]
// (float atomics: an id may occur several times) and / or zero_tab[ids[i]] = 0.
__global__ __launch_bounds__(kBlock) void rows_finish_kernel(float *__restrict__ dst, int64_t ldd, const float *__restrict__ src,
                                                             int64_t lds, float *__restrict__ zero_tab, int64_t ldz,
                                                             const int64_t *__restrict__ ids, int64_t n, int d,
                                                             const float *__restrict__ scale_dev, float scale_host)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t r = ids[i];
    if (dst) {
        const float sc = scale_dev ? scale_host * scale_dev[0] : scale_host;
        for (int j = lane; j < d; j += kWave) atomicAdd(dst + r * ldd + j, sc * src[r * lds + j]);
    }
    if (zero_tab)
        for (int j = lane; j < d; j += kWave) zero_tab[r * ldz + j] = 0.f;
}


// Row-sharded training (igcn_cf_amd/dist.py): of the batch's node ids, a rank owns the users in [ulo, uhi) and the
// items (id - n_users) in [ilo, ihi).  One wave per id.
__global__ __launch_bounds__(kBlock) void owned_rows_gather_kernel(
    const int64_t *__restrict__ ids, int64_t n, int64_t n_users, int64_t ulo, int64_t uhi, int64_t ilo, int64_t ihi,
    const float *__restrict__ tab_u, int64_t ld_u, const float *__restrict__ tab_i, int64_t ld_i, int d,
    float *__restrict__ out, int64_t ld_out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t id = ids[i];
    const float *src = nullptr;
    if (id < n_users) { if (id >= ulo && id < uhi) src = tab_u + (id - ulo) * ld_u; }
    else { const int64_t it = id - n_users; if (it >= ilo && it < ihi) src = tab_i + (it - ilo) * ld_i; }
    for (int j = lane; j < d; j += kWave) out[i * ld_out + j] = src ? src[j] : 0.f;      // rows of other ranks: zero (x + 0 is exact)
}

__global__ __launch_bounds__(kBlock) void owned_rows_scatter_add_kernel(
    const int64_t *__restrict__ ids, int64_t n, int64_t n_users, int64_t ulo, int64_t uhi, int64_t ilo, int64_t ihi,
    const float *__restrict__ g, int64_t ld_g, int d,
    float *__restrict__ gu, int64_t ld_gu, float *__restrict__ gi, int64_t ld_gi)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t id = ids[i];
    float *dst = nullptr;
    if (id < n_users) { if (id >= ulo && id < uhi) dst = gu + (id - ulo) * ld_gu; }
    else { const int64_t it = id - n_users; if (it >= ilo && it < ihi) dst = gi + (it - ilo) * ld_gi; }
    if (!dst) return;
    for (int j = lane; j < d; j += kWave) atomicAdd(dst + j, g[i * ld_g + j]);            // an id may occur several times
}

}  // namespace igcn

using namespace igcn;

extern "C" int igcn_rows_finish_f32(float *dst, int64_t ldd, const float *src, int64_t lds, float *zero_tab, int64_t ldz,
                                    const int64_t *ids, int64_t n, int32_t d, const float *scale_dev, float scale_host,
                                    void *stream)
{
    if (!ids || (!dst && !zero_tab) || (dst && !src)) return IGCN_E_NULL;
    if (n < 0 || d < 1 || (dst && (ldd < d || lds < d)) || (zero_tab && ldz < d)) return IGCN_E_SHAPE;
    if (dst && zero_tab == dst) return IGCN_E_RANGE;
    if (n == 0) return IGCN_OK;
    hipLaunchKernelGGL(rows_finish_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       dst, ldd, src, lds, zero_tab, ldz, ids, n, (int)d, scale_dev, scale_host);
    return launch_status();
}

static int bpr_check(const float *u, const float *p, const float *n, int64_t ld,
                     const float *a, const float *b, const float *c, int64_t ld2,
                     const int64_t *users, const int64_t *pos, const int64_t *neg, int64_t batch, int32_t d)
{
    if (!u || !p || !n || !users || !pos || !neg) return IGCN_E_NULL;
    const int n_l2 = (a != nullptr) + (b != nullptr) + (c != nullptr);
    if (n_l2 != 0 && n_l2 != 3) return IGCN_E_NULL;
    if (batch < 0 || d < 1 || ld < d || (n_l2 == 3 && ld2 < d)) return IGCN_E_SHAPE;
    return IGCN_OK;
}

extern "C" int igcn_bpr_fwd_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                                const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                                const int64_t *users, const int64_t *pos, const int64_t *neg,
                                int64_t batch, int32_t d, const float *w,
                                float *loss_out, float *work, void *stream)
{
    int rc = bpr_check(u_tab, p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, d);
    if (rc != IGCN_OK) return rc;
    if (!loss_out || !work) return IGCN_E_NULL;
    if (batch == 0) return IGCN_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t blocks = (batch + 3) / 4;
    hipLaunchKernelGGL(bpr_fwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, u_tab, p_tab, n_tab, ld,
                       l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, (int)d, w, work);
    rc = launch_status();
    if (rc != IGCN_OK) return rc;
    hipLaunchKernelGGL(bpr_reduce_kernel, dim3(1), dim3(kBlock), 0, st, work, batch, loss_out);
    return launch_status();
}

extern "C" int igcn_bpr_loss_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                                 const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                                 const int64_t *users, const int64_t *pos, const int64_t *neg,
                                 int64_t batch, int32_t d, const float *w, float l2_weight,
                                 float *loss_out3, float *work, void *stream)
{
    int rc = bpr_check(u_tab, p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, d);
    if (rc != IGCN_OK) return rc;
    if (!loss_out3 || !work) return IGCN_E_NULL;
    if (batch == 0) return IGCN_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bpr_fwd_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(kBlock), 0, st, u_tab, p_tab, n_tab, ld,
                       l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, (int)d, w, work);
    rc = launch_status();
    if (rc != IGCN_OK) return rc;
    hipLaunchKernelGGL(bpr_reduce_kernel, dim3(1), dim3(kBlock), 0, st, work, batch, loss_out3, l2_weight, 1);
    return launch_status();
}

extern "C" int igcn_bpr_dots_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                                 const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                                 const int64_t *users, const int64_t *pos, const int64_t *neg,
                                 int64_t batch, int32_t d, const float *w, float *dots, void *stream)
{
    int rc = bpr_check(u_tab, p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, d);
    if (rc != IGCN_OK) return rc;
    if (!dots) return IGCN_E_NULL;
    if (batch == 0) return IGCN_E_SHAPE;
    const int64_t blocks = (batch + 3) / 4;
    hipLaunchKernelGGL(bpr_dots_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), u_tab,
                       p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, (int)d, w, dots);
    return launch_status();
}

extern "C" int igcn_bpr_finish_f32(const float *dots, int64_t batch, float *loss_out, float *work, void *stream)
{
    if (!dots || !loss_out || !work) return IGCN_E_NULL;
    if (batch < 1) return IGCN_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bpr_finish_kernel, dim3((unsigned)((batch + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, dots, batch, work);
    int rc = launch_status();
    if (rc != IGCN_OK) return rc;
    hipLaunchKernelGGL(bpr_reduce_kernel, dim3(1), dim3(kBlock), 0, st, work, batch, loss_out);
    return launch_status();
}

static int bpr_bwd_launch(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                          const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                          const int64_t *users, const int64_t *pos, const int64_t *neg,
                          int64_t batch, int32_t d, const float *w, const float *work, const float *g_out,
                          float *gu_tab, float *gp_tab, float *gn_tab,
                          float *gl2_u_tab, float *gl2_p_tab, float *gl2_n_tab,
                          float *gw_out, int g_len, float s0, float s1, void *stream)
{
    int rc = bpr_check(u_tab, p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, d);
    if (rc != IGCN_OK) return rc;
    if (!work || !g_out || !gu_tab || !gp_tab || !gn_tab) return IGCN_E_NULL;
    const int n_g2 = (gl2_u_tab != nullptr) + (gl2_p_tab != nullptr) + (gl2_n_tab != nullptr);
    if (n_g2 != 0 && n_g2 != 3) return IGCN_E_NULL;
    if (n_g2 == 3 && !l2_u_tab) return IGCN_E_NULL;
    if (gw_out && !w) return IGCN_E_NULL;
    if (batch == 0) return IGCN_E_SHAPE;
    int64_t blocks = (batch + 3) / 4;
    if (gw_out && blocks > 64) blocks = 64;      // a few triplets per wave: d loss / d w is added once per wave, not once per triplet
    hipLaunchKernelGGL(bpr_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       u_tab, p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, (int)d, w,
                       work, g_out, gu_tab, gp_tab, gn_tab, gl2_u_tab, gl2_p_tab, gl2_n_tab, gw_out, g_len, s0, s1);
    return launch_status();
}

extern "C" int igcn_bpr_bwd_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                                const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                                const int64_t *users, const int64_t *pos, const int64_t *neg,
                                int64_t batch, int32_t d, const float *w, const float *work, const float *g_out,
                                float *gu_tab, float *gp_tab, float *gn_tab,
                                float *gl2_u_tab, float *gl2_p_tab, float *gl2_n_tab,
                                float *gw_out, void *stream)
{
    return bpr_bwd_launch(u_tab, p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, d, w, work, g_out,
                          gu_tab, gp_tab, gn_tab, gl2_u_tab, gl2_p_tab, gl2_n_tab, gw_out, 2, 1.f, 1.f, stream);
}

extern "C" int igcn_bpr_loss_bwd_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                                     const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                                     const int64_t *users, const int64_t *pos, const int64_t *neg,
                                     int64_t batch, int32_t d, const float *w, const float *work, const float *g_loss,
                                     float l2_weight,
                                     float *gu_tab, float *gp_tab, float *gn_tab,
                                     float *gl2_u_tab, float *gl2_p_tab, float *gl2_n_tab,
                                     float *gw_out, void *stream)
{
    return bpr_bwd_launch(u_tab, p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, d, w, work, g_loss,
                          gu_tab, gp_tab, gn_tab, gl2_u_tab, gl2_p_tab, gl2_n_tab, gw_out, 1, 1.f, l2_weight, stream);
}

extern "C" int igcn_bpr_loss_bwd_scaled_f32(const float *u_tab, const float *p_tab, const float *n_tab, int64_t ld,
                                            const float *l2_u_tab, const float *l2_p_tab, const float *l2_n_tab, int64_t ld_l2,
                                            const int64_t *users, const int64_t *pos, const int64_t *neg,
                                            int64_t batch, int32_t d, const float *w, const float *work, const float *g_loss,
                                            float bpr_weight, float l2_weight,
                                            float *gu_tab, float *gp_tab, float *gn_tab,
                                            float *gl2_u_tab, float *gl2_p_tab, float *gl2_n_tab,
                                            float *gw_out, void *stream)
{
    return bpr_bwd_launch(u_tab, p_tab, n_tab, ld, l2_u_tab, l2_p_tab, l2_n_tab, ld_l2, users, pos, neg, batch, d, w, work, g_loss,
                          gu_tab, gp_tab, gn_tab, gl2_u_tab, gl2_p_tab, gl2_n_tab, gw_out, 1, bpr_weight, l2_weight, stream);
}

extern "C" int igcn_owned_rows_gather_f32(const int64_t *ids, int64_t n, int64_t n_users, int64_t ulo, int64_t uhi,
                                          int64_t ilo, int64_t ihi, const float *tab_u, int64_t ld_u,
                                          const float *tab_i, int64_t ld_i, int32_t d, float *out, int64_t ld_out, void *stream)
{
    if (!ids || !out || (uhi > ulo && !tab_u) || (ihi > ilo && !tab_i)) return IGCN_E_NULL;
    if (n < 0 || d < 1 || ld_out < d || (tab_u && ld_u < d) || (tab_i && ld_i < d) || uhi < ulo || ihi < ilo) return IGCN_E_SHAPE;
    if (n == 0) return IGCN_OK;
    hipLaunchKernelGGL(owned_rows_gather_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       ids, n, n_users, ulo, uhi, ilo, ihi, tab_u, ld_u, tab_i, ld_i, (int)d, out, ld_out);
    return launch_status();
}

extern "C" int igcn_owned_rows_scatter_add_f32(const int64_t *ids, int64_t n, int64_t n_users, int64_t ulo, int64_t uhi,
                                               int64_t ilo, int64_t ihi, const float *g, int64_t ld_g, int32_t d,
                                               float *gu, int64_t ld_gu, float *gi, int64_t ld_gi, void *stream)
{
    if (!ids || !g || (uhi > ulo && !gu) || (ihi > ilo && !gi)) return IGCN_E_NULL;
    if (n < 0 || d < 1 || ld_g < d || (gu && ld_gu < d) || (gi && ld_gi < d) || uhi < ulo || ihi < ilo) return IGCN_E_SHAPE;
    if (n == 0) return IGCN_OK;
    hipLaunchKernelGGL(owned_rows_scatter_add_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), ids, n, n_users, ulo, uhi, ilo, ihi, g, ld_g, (int)d, gu, ld_gu, gi, ld_gi);
    return launch_status();
}
