// vibo_trainer.hip -- the O(I) part of a VIBO train step as two kernels (see include/vibo_hip.h):
// prologue (item sample, item KL, 2-row encoder MLP forward) and epilogue (loss, MLP backward, item
// backward, Adam).  One workgroup (block 0) owns the MLP; the other workgroups own 256 item entries each.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_philox.hpp"

namespace vibo {

constexpr int kMaxHidden = 256;

struct MlpOffsets {
    int w0, b0, w1, b1, w2, b2, total;
};
__host__ __device__ inline MlpOffsets mlp_offsets(int H, int O) {
    MlpOffsets o;
    o.w0 = 0; o.b0 = H; o.w1 = 2 * H; o.b1 = 2 * H + H * H; o.w2 = o.b1 + H; o.b2 = o.w2 + O * H; o.total = o.b2 + O;
    return o;
}

__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : expm1f(x); }

__global__ __launch_bounds__(256) void train_prologue_kernel(int H, int O, int n_item_entries, const float* __restrict__ P,
                                                             const float* __restrict__ mu, const float* __restrict__ lv,
                                                             const float* __restrict__ eps, float* __restrict__ item_feat,
                                                             float* __restrict__ table, float* __restrict__ saved_h,
                                                             float* __restrict__ kl_parts, int32_t* step_count,
                                                             // noise (gen != 0): eps is written here instead of read, and the
                                                             // blocks past the item blocks fill eps_ab [n_ab]; both draws are
                                                             // the streams vibo_fill_normal gives for step_count[1]
                                                             int gen, uint32_t seed_lo, uint32_t seed_hi, float* __restrict__ eps_w,
                                                             float* __restrict__ eps_ab, long long n_ab, uint32_t ab_stream,
                                                             int n_item_blocks) {
    __shared__ float h1[2][kMaxHidden], h2[2][kMaxHidden];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        if (tid == 0) *step_count += 1;
        const MlpOffsets o = mlp_offsets(H, O);
        for (int t = tid; t < 2 * H; t += 256) {          // layer 0: input is the response value c in {0,1}
            const int r = t / H, j = t % H;
            h1[r][j] = elu(P[o.w0 + j] * (float)r + P[o.b0 + j]);
        }
        __syncthreads();
        for (int t = tid; t < 2 * H; t += 256) {
            const int r = t / H, j = t % H;
            float a = P[o.b1 + j];
#pragma unroll 16
            for (int k = 0; k < H; ++k) a = fmaf(P[o.w1 + j * H + k], h1[r][k], a);
            h2[r][j] = elu(a);
        }
        __syncthreads();
        for (int t = tid; t < 2 * O; t += 256) {
            const int r = t / O, q = t % O;
            float a = P[o.b2 + q];
#pragma unroll 16
            for (int k = 0; k < H; ++k) a = fmaf(P[o.w2 + q * H + k], h2[r][k], a);
            table[r * O + q] = a;
        }
        for (int t = tid; t < 2 * H; t += 256) {
            saved_h[t] = h1[t / H][t % H];
            saved_h[2 * H + t] = h2[t / H][t % H];
        }
        return;
    }
    if ((int)blockIdx.x > n_item_blocks) {          // ability noise (stream ab_stream), 4 normals per thread
        const long long g = (long long)(blockIdx.x - 1 - n_item_blocks) * 256 + tid;
        if (4 * g < n_ab) store_normal4(eps_ab, n_ab, g, philox_normal4(g, (uint32_t)step_count[1], ab_stream, seed_lo, seed_hi));
        return;
    }
    // item side: 256 entries of [I][D] per workgroup
    const int idx = (blockIdx.x - 1) * 256 + tid;
    float kl = 0.f;
    if (idx < n_item_entries) {
        const float m = mu[idx], l = lv[idx];
        float e;
        if (gen) {                                   // entry idx of stream 0 (its group of 4 is recomputed by 4 threads: O(I) work)
            const float4 z = philox_normal4(idx >> 2, (uint32_t)step_count[1], 0u, seed_lo, seed_hi);
            e = (idx & 3) == 0 ? z.x : (idx & 3) == 1 ? z.y : (idx & 3) == 2 ? z.z : z.w;
            eps_w[idx] = e;
        } else {
            e = eps[idx];
        }
        item_feat[idx] = fmaf(expf(0.5f * l), e, m);
        kl = -0.5f * (1.0f + l - m * m - expf(l));
    }
    kl = wave_total(kl);
    if ((tid & 63) == 0) red[tid >> 6] = kl;
    __syncthreads();
    if (tid == 0) kl_parts[blockIdx.x - 1] = red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ void adam_update(float& p, float& m, float& v, const float g, const float lr, const float bc1,
                                            const float bc2_sqrt) {
    m = 0.9f * m + 0.1f * g;                       // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = 0.999f * v + 0.001f * g * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / bc2_sqrt + 1e-8f;
    p -= (lr / bc1) * (m / denom);
}

constexpr int kEpiThreads = 1024;      // block 0's chain of small dependent stages is latency-bound: more lanes per stage, fewer passes
__global__ __launch_bounds__(kEpiThreads) void train_epilogue_kernel(int H, int O, int n_item_entries, int n_kl_parts,
                                                             const float* __restrict__ flat, const float* __restrict__ saved_h,
                                                             const float* __restrict__ kl_parts, const float* __restrict__ eps,
                                                             const float* __restrict__ beta_p, const float* __restrict__ lr_p,
                                                             const int32_t* __restrict__ step_count, float* P, float* M, float* V,
                                                             float* mu, float* lv, float* im, float* iv, float* loss_out) {
    __shared__ float h1[2][kMaxHidden], h2[2][kMaxHidden], gh2[2][kMaxHidden], gh1[2][kMaxHidden], gout[2][2 * VIBO_MAX_ABILITY_DIM];
    const int tid = threadIdx.x;
    const float beta = *beta_p, lr = *lr_p;
    const float t = (float)(*step_count);
    const float bc1 = 1.0f - powf(0.9f, t), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t));
    const int n_table = 2 * O;
    constexpr int BS = kEpiThreads;
    if (blockIdx.x == 0) {
        if (tid == 0) const_cast<int32_t*>(step_count)[1] += 1;      // completed steps: the noise counter of the NEXT step
        const MlpOffsets o = mlp_offsets(H, O);
        for (int k = tid; k < 2 * H; k += BS) {
            h1[k / H][k % H] = saved_h[k];
            h2[k / H][k % H] = saved_h[2 * H + k];
        }
        // d loss / d table = -dLL + beta dREG     (flat: [8 scalars | grad_table set 0 | set 1 | grad_item])
        for (int k = tid; k < n_table; k += BS) gout[k / O][k % O] = -flat[VIBO_NUM_SCALARS + k] + beta * flat[VIBO_NUM_SCALARS + n_table + k];
        if (tid < 64) {                  // item KL: the prologue's partial sums, fixed order (lane-strided, then the wave sum)
            float kl = 0.f;
            for (int k = tid; k < n_kl_parts; k += 64) kl += kl_parts[k];
            kl = wave_total(kl);
            if (tid == 0) *loss_out = -flat[VIBO_S_LL] + beta * (flat[VIBO_S_REG] + kl);
        }
        __syncthreads();
        // g_h2 = W2^T g_out * elu'(pre2),  elu'(x) = x > 0 ? 1 : elu(x) + 1
        for (int k = tid; k < 2 * H; k += BS) {
            const int r = k / H, j = k % H;
            float a = 0.f;
#pragma unroll 16
            for (int q = 0; q < O; ++q) a = fmaf(P[o.w2 + q * H + j], gout[r][q], a);      // 16 loads in flight
            const float h = h2[r][j];
            gh2[r][j] = a * (h > 0.f ? 1.0f : h + 1.0f);
        }
        __syncthreads();
        // g_h1 = W1^T g_h2 * elu'(pre1): each of the 2 H dot products over H is cut into 8 pieces (8 neighbouring lanes)
        {
            const int len = (H + 7) / 8;
            for (int k0 = 0; k0 < 2 * H * 8; k0 += BS) {
                const int k = k0 + tid;
                const int out = k >> 3, part = k & 7;
                const int r = out / H, j = out % H;
                float a = 0.f;
                if (out < 2 * H) {
                    const int q1 = min(H, (part + 1) * len);
                    for (int q = part * len; q < q1; ++q) a = fmaf(P[o.w1 + q * H + j], gh2[r][q], a);
                }
                a += __shfl_xor(a, 1);
                a += __shfl_xor(a, 2);
                a += __shfl_xor(a, 4);
                if (out < 2 * H && part == 0) {
                    const float h = h1[r][j];
                    gh1[r][j] = a * (h > 0.f ? 1.0f : h + 1.0f);
                }
            }
        }
        __syncthreads();      // all reads of the OLD weights are done: parameters may now be updated in place
        // Adam over the MLP parameters: 8 independent elements per thread and pass, all loads issued before the
        // first store (P, M, V are not restrict-qualified, so a store would otherwise fence the next loads)
        constexpr int U = 8;
        for (int k0 = tid; k0 < o.total; k0 += BS * U) {
            float pv[U], mv[U], vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + BS * u;
                const bool ok = k < o.total;
                pv[u] = ok ? P[k] : 0.f;
                mv[u] = ok ? M[k] : 0.f;
                vv[u] = ok ? V[k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + BS * u;
                if (k >= o.total) continue;
                float g;
                if (k < o.b0) {                         // W0[j]: input of row r is r
                    g = gh1[1][k - o.w0];
                } else if (k < o.w1) {
                    const int j = k - o.b0;
                    g = gh1[0][j] + gh1[1][j];
                } else if (k < o.b1) {
                    const int j = (k - o.w1) / H, q = (k - o.w1) % H;
                    g = gh2[0][j] * h1[0][q] + gh2[1][j] * h1[1][q];
                } else if (k < o.w2) {
                    const int j = k - o.b1;
                    g = gh2[0][j] + gh2[1][j];
                } else if (k < o.b2) {
                    const int q = (k - o.w2) / H, j = (k - o.w2) % H;
                    g = gout[0][q] * h2[0][j] + gout[1][q] * h2[1][j];
                } else {
                    const int q = k - o.b2;
                    g = gout[0][q] + gout[1][q];
                }
                adam_update(pv[u], mv[u], vv[u], g, lr, bc1, bc2_sqrt);
                P[k] = pv[u];
                M[k] = mv[u];
                V[k] = vv[u];
            }
        }
        return;
    }
    const int idx = (blockIdx.x - 1) * BS + tid;
    if (idx < n_item_entries) {
        const float m = mu[idx], l = lv[idx];
        const float gf = -flat[VIBO_NUM_SCALARS + 2 * n_table + idx];          // d loss / d item_feat = -dLL/ditem
        const float g_mu = gf + beta * m;
        const float g_lv = gf * 0.5f * expf(0.5f * l) * eps[idx] - 0.5f * beta * (1.0f - expf(l));
        float pm = m, pl = l;
        adam_update(pm, im[idx], iv[idx], g_mu, lr, bc1, bc2_sqrt);
        adam_update(pl, im[n_item_entries + idx], iv[n_item_entries + idx], g_lv, lr, bc1, bc2_sqrt);
        mu[idx] = pm;
        lv[idx] = pl;
    }
}

__global__ __launch_bounds__(256) void fill_normal_kernel(float* __restrict__ out, long long n, uint32_t seed_lo, uint32_t seed_hi,
                                                          const int32_t* __restrict__ step_count, uint32_t stream_id) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;        // group of 4 outputs
    if (4 * g >= n) return;
    store_normal4(out, n, g, philox_normal4(g, (uint32_t)(*step_count), stream_id, seed_lo, seed_hi));
}

}  // namespace vibo

using namespace vibo;

extern "C" int vibo_fill_normal(float* out, int64_t n, uint64_t seed, const int32_t* step_count, uint32_t stream_id, void* stream) {
    if (!out || !step_count || n < 0) return -5;
    if (n == 0) return 0;
    const long long groups = (n + 3) / 4;
    hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                       (long long)n, (uint32_t)seed, (uint32_t)(seed >> 32), step_count, stream_id);
    return (int)hipGetLastError();
}

static int item_dim_of(const vibo_desc* d) { return d->irt_model == 1 ? 1 : (d->irt_model == 2 ? d->ability_dim + 1 : d->ability_dim + 2); }

extern "C" int vibo_train_prologue(const vibo_desc* d, int hidden_dim, const float* mlp_params, const float* item_mu,
                                   const float* item_logvar, const float* eps_item, float* item_feat, float* table,
                                   float* saved_h, float* kl_parts, int32_t* step_count, void* stream) {
    if (!d || hidden_dim < 1 || hidden_dim > kMaxHidden || d->posterior != VIBO_POSTERIOR_UNCONDITIONAL || d->n_flows != 0) return -6;
    const int n = d->num_item * item_dim_of(d);
    const int blocks = 1 + (n + 255) / 256;
    hipLaunchKernelGGL(train_prologue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hidden_dim, 2 * d->ability_dim, n,
                       mlp_params, item_mu, item_logvar, eps_item, item_feat, table, saved_h, kl_parts, step_count, 0, 0u, 0u,
                       (float*)nullptr, (float*)nullptr, 0LL, 0u, (n + 255) / 256);
    return (int)hipGetLastError();
}

extern "C" int vibo_train_prologue_noise(const vibo_desc* d, int hidden_dim, const float* mlp_params, const float* item_mu,
                                         const float* item_logvar, float* eps_item, float* item_feat, float* table,
                                         float* saved_h, float* kl_parts, int32_t* step_count, uint64_t seed, float* eps_ability,
                                         uint32_t ability_stream_id, void* stream) {
    if (!d || hidden_dim < 1 || hidden_dim > kMaxHidden || d->posterior != VIBO_POSTERIOR_UNCONDITIONAL || d->n_flows != 0) return -6;
    if (!eps_item || !eps_ability || !step_count) return -5;
    const int n = d->num_item * item_dim_of(d);
    const int item_blocks = (n + 255) / 256;
    const long long n_ab = (long long)d->num_person * d->ability_dim;
    const long long ab_blocks = ((n_ab + 3) / 4 + 255) / 256;
    hipLaunchKernelGGL(train_prologue_kernel, dim3((unsigned)(1 + item_blocks + ab_blocks)), dim3(256), 0, (hipStream_t)stream,
                       hidden_dim, 2 * d->ability_dim, n, mlp_params, item_mu, item_logvar, (const float*)eps_item, item_feat, table,
                       saved_h, kl_parts, step_count, 1, (uint32_t)seed, (uint32_t)(seed >> 32), eps_item, eps_ability, n_ab,
                       ability_stream_id, item_blocks);
    return (int)hipGetLastError();
}

extern "C" int vibo_train_epilogue(const vibo_desc* d, int hidden_dim, const float* flat, const float* saved_h,
                                   const float* kl_parts, const float* eps_item, const float* beta, const float* lr,
                                   const int32_t* step_count, float* mlp_params, float* mlp_m, float* mlp_v,
                                   float* item_mu, float* item_logvar, float* item_m, float* item_v, float* loss_out,
                                   void* stream) {
    if (!d || hidden_dim < 1 || hidden_dim > kMaxHidden || d->posterior != VIBO_POSTERIOR_UNCONDITIONAL || d->n_flows != 0) return -6;
    const int n = d->num_item * item_dim_of(d);
    const int parts = (n + 255) / 256;                       // the prologue's item-KL partial sums (256 entries each)
    hipLaunchKernelGGL(train_epilogue_kernel, dim3(1 + (n + kEpiThreads - 1) / kEpiThreads), dim3(kEpiThreads), 0, (hipStream_t)stream,
                       hidden_dim, 2 * d->ability_dim, n, parts, flat, saved_h, kl_parts, eps_item, beta, lr, step_count, mlp_params, mlp_m, mlp_v, item_mu,
                       item_logvar, item_m, item_v, loss_out);
    return (int)hipGetLastError();
}
