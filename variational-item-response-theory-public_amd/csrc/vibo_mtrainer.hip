// vibo_mtrainer.hip -- the O(I) + O(1) part of a train step of the --ability-merge mean encoder (unconditional posterior, IRT
// decoder, no flows) as native kernels: what vibo_trainer.hip is for the product-of-experts encoder.
//
// Reference (models.py:584-594, 631-650; vibo.py:243-268): hid_pi = elu(mlp1([x_pi])), mlp1 = Linear(1,H) -> ELU -> Linear(H,H);
// hid_mean_p = mean over the person's observed items; (mu_p | logvar_p) = mlp2(hid_mean_p), mlp2 = Linear(H,H) -> ELU ->
// Linear(H,2A).  A Bernoulli response takes two values, so the per-term features are two H-vectors h0, h1 and the first layer of
// mlp2 is affine in w_p = n_correct / n_observed:  z_p = u + w_p v,  u = W20 h0 + b20,  v = W20 (h1 - h0)  (vibo_mean.hip does the
// per-person rest).  Here:
//   mt_prologue_kernel   block 0: the 2-row mlp1 forward, u, v (+ the activations kept for the backward);
//                        other blocks: item sample / item KL / optional Philox noise, as train_prologue_kernel
//   mt_reduce_kernel     fixed-order sums of vibo_mean_encoder_backward's per-wave records -> d/du | d/dv | d/dW22 | d/db22
//   mt_epilogue_kernel   block 0: loss, the backward through u, v, mlp2[0] and the 2-row mlp1 by hand, Adam on all of it;
//                        other blocks: item backward + Adam, as train_epilogue_kernel
// Parameter layout (one flat buffer, the nn.Parameters are views of it):
//   W10 [H] | b10 [H] | W12 [H][H] | b12 [H] | W20 [H][H] | b20 [H] | W22 [2A][H] | b22 [2A]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_finalize.hpp"
#include "vibo_philox.hpp"
#include "vibo_train_hook.hpp"

namespace vibo {

struct MOff {
    int w10, b10, w12, b12, w20, b20, w22, b22, total;
};
__host__ __device__ inline MOff moff(const int H, const int A2) {
    MOff o;
    o.w10 = 0; o.b10 = H; o.w12 = 2 * H; o.b12 = o.w12 + H * H; o.w20 = o.b12 + H; o.b20 = o.w20 + H * H; o.w22 = o.b20 + H;
    o.b22 = o.w22 + A2 * H; o.total = o.b22 + A2;
    return o;
}
constexpr int kMtMaxHidden = 128;      // (block 0 keeps four [2][H] activation sets and two H-vectors in LDS)

// saved activations: a [2][H] (first layer, after its ELU) | hf [2][H] (the two per-term features)
__global__ __launch_bounds__(256) void mt_prologue_kernel(int H, int A2, int I, int D, const float* __restrict__ P,
                                                          const float* __restrict__ mu, const float* __restrict__ lv,
                                                          const float* __restrict__ eps, float* __restrict__ item_feat,
                                                          float* __restrict__ uv, float* __restrict__ saved,
                                                          float* __restrict__ kl_parts, int32_t* step_count, int gen, uint32_t seed_lo,
                                                          uint32_t seed_hi, float* __restrict__ eps_w, float* __restrict__ eps_ab,
                                                          long long n_ab, uint32_t ab_stream, int n_item_blocks) {
    __shared__ float a[2 * kMtMaxHidden], hf[2 * kMtMaxHidden];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        if (tid == 0) *step_count += 1;
        const MOff o = moff(H, A2);
        for (int t = tid; t < 2 * H; t += 256) {
            const int r = t / H, j = t % H;
            a[t] = elu(fmaf(P[o.w10 + j], (float)r, P[o.b10 + j]));
        }
        __syncthreads();
        for (int t = tid; t < 2 * H; t += 256) {
            const int r = t / H, j = t % H;
            float s = P[o.b12 + j];
#pragma unroll 16
            for (int k = 0; k < H; ++k) s = fmaf(P[o.w12 + j * H + k], a[r * H + k], s);
            hf[t] = elu(s);                                   // elu(mlp1(x)): models.py:634
        }
        __syncthreads();
        for (int j = tid; j < H; j += 256) {
            float u = P[o.b20 + j], v = 0.f;
#pragma unroll 16
            for (int k = 0; k < H; ++k) {
                const float w = P[o.w20 + j * H + k];
                u = fmaf(w, hf[k], u);
                v = fmaf(w, hf[H + k] - hf[k], v);
            }
            uv[j] = u;
            uv[H + j] = v;
        }
        for (int t = tid; t < 2 * H; t += 256) {
            saved[t] = a[t];
            saved[2 * H + t] = hf[t];
        }
        return;
    }
    if ((int)blockIdx.x > n_item_blocks) {          // ability noise (stream ab_stream), 4 normals per thread
        const long long g = (long long)(blockIdx.x - 1 - n_item_blocks) * 256 + tid;
        if (4 * g < n_ab) store_normal4(eps_ab, n_ab, g, philox_normal4(g, (uint32_t)step_count[1], ab_stream, seed_lo, seed_hi));
        return;
    }
    // item side: as train_prologue_kernel (dimension-major entries, one KL part per wave)
    const int n_item_entries = I * D;
    const int k = (blockIdx.x - 1) * 256 + tid;
    float kl = 0.f;
    if (k < n_item_entries) {
        const int idx = item_entry_index(k, I, D);
        const float m = mu[idx], l = lv[idx];
        float e;
        if (gen) {
            e = philox_normal1(idx, (uint32_t)step_count[1], 0u, seed_lo, seed_hi);
            eps_w[idx] = e;
        } else {
            e = eps[idx];
        }
        item_feat[idx] = item_sample(m, l, e);
        kl = item_kl_term(m, l);
    }
    kl = wave_total(kl);
    if ((tid & 63) == 0 && 256 * ((int)blockIdx.x - 1) + (tid & ~63) < n_item_entries) kl_parts[4 * (blockIdx.x - 1) + (tid >> 6)] = kl;
}

// out[e] = sum over the records of part[r][e], fixed order (16 slices, fp64): 64 outputs per workgroup
__global__ __launch_bounds__(1024) void mt_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int n_rec, int n_out) {
    __shared__ double sl[16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    sl[slice][lane] = e < n_out ? record_slice_sum<16>(part, (size_t)n_out, e, 0, n_rec, slice) : 0.0;
    __syncthreads();
    if (slice == 0 && e < n_out) {
        double t = 0.0;
#pragma unroll
        for (int s = 0; s < 16; ++s) t += sl[s][lane];
        out[e] = (float)t;
    }
}

__device__ __forceinline__ void mt_adam(float& p, float& m, float& v, const float g, const float lr, const float bc1, const float bc2_sqrt) {
    m = fmaf(0.9f, m, 0.1f * g);
    v = fmaf(0.999f, v, (0.001f * g) * g);
    const float denom = sqrtf(v) / bc2_sqrt + 1e-8f;
    p -= (lr / bc1) * (m / denom);
}

constexpr int kMtThreads = 1024;
__global__ __launch_bounds__(kMtThreads) void mt_epilogue_kernel(int H, int A2, int n_item_entries, const float* __restrict__ flat,
                                                                 const float* __restrict__ gsum, const float* __restrict__ saved,
                                                                 const float* __restrict__ kl_parts, const float* __restrict__ eps,
                                                                 const float* __restrict__ beta_p, const float* __restrict__ lr_p,
                                                                 int32_t* step_count, long long n_table, float* P, float* M, float* V,
                                                                 float* mu, float* lv, float* im, float* iv, float* loss_out) {
    __shared__ float a[2][kMtMaxHidden], hf[2][kMtMaxHidden], du[kMtMaxHidden], dv[kMtMaxHidden];
    __shared__ float df[2][kMtMaxHidden], dpre[2][kMtMaxHidden];
    const int tid = threadIdx.x;
    const float beta = *beta_p, lr = *lr_p;
    const float t = (float)step_count[0];
    const float bc1 = 1.0f - powf(0.9f, t), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t));
    constexpr int BS = kMtThreads;
    if (blockIdx.x == 0) {
        if (tid == 0) step_count[1] += 1;
        const MOff o = moff(H, A2);
        for (int k = tid; k < 2 * H; k += BS) {
            a[k / H][k % H] = saved[k];
            hf[k / H][k % H] = saved[2 * H + k];
        }
        for (int k = tid; k < H; k += BS) {
            du[k] = gsum[k];
            dv[k] = gsum[H + k];
        }
        if (tid < 64) {                  // item KL: the prologue's partial sums, fixed order
            float kl = 0.f;
            const int n_parts = kl_part_count(n_item_entries);
            for (int k = tid; k < n_parts; k += 64) kl += kl_parts[k];
            kl = wave_total(kl);
            if (tid == 0) *loss_out = fmaf(beta, flat[VIBO_S_REG] + kl, -flat[VIBO_S_LL]);
        }
        __syncthreads();
        // d hf[c][k] = sum_j W20[j][k] (du[j] - dv[j] | dv[j]), through the ELU of the features
        for (int e = tid; e < 2 * H; e += BS) {
            const int c = e / H, k = e % H;
            float s = 0.f;
#pragma unroll 16
            for (int j = 0; j < H; ++j) s = fmaf(P[o.w20 + j * H + k], c == 0 ? du[j] - dv[j] : dv[j], s);
            const float h = hf[c][k];
            df[c][k] = s * (h > 0.f ? 1.0f : h + 1.0f);
        }
        __syncthreads();
        // d a[c][q] = sum_k W12[k][q] df[c][k], through the first layer's ELU
        for (int e = tid; e < 2 * H; e += BS) {
            const int c = e / H, q = e % H;
            float s = 0.f;
#pragma unroll 16
            for (int k = 0; k < H; ++k) s = fmaf(P[o.w12 + k * H + q], df[c][k], s);
            const float h = a[c][q];
            dpre[c][q] = s * (h > 0.f ? 1.0f : h + 1.0f);
        }
        __syncthreads();      // all reads of the OLD weights are done
        for (int k = tid; k < o.total; k += BS) {
            float g;
            if (k < o.b10) {                                  // W10[q]: the input of row c is c
                g = dpre[1][k - o.w10];
            } else if (k < o.w12) {
                const int q = k - o.b10;
                g = dpre[0][q] + dpre[1][q];
            } else if (k < o.b12) {
                const int kk = (k - o.w12) / H, q = (k - o.w12) % H;
                g = fmaf(df[0][kk], a[0][q], df[1][kk] * a[1][q]);
            } else if (k < o.w20) {
                const int kk = k - o.b12;
                g = df[0][kk] + df[1][kk];
            } else if (k < o.b20) {
                const int j = (k - o.w20) / H, kk = (k - o.w20) % H;
                g = fmaf(du[j], hf[0][kk], dv[j] * (hf[1][kk] - hf[0][kk]));
            } else if (k < o.w22) {
                g = du[k - o.b20];
            } else {                                          // W22 | b22: summed over the persons by vibo_mean_encoder_backward
                g = gsum[2 * H + (k - o.w22)];
            }
            float pv = P[k], mv = M[k], vv = V[k];
            mt_adam(pv, mv, vv, g, lr, bc1, bc2_sqrt);
            P[k] = pv; M[k] = mv; V[k] = vv;
        }
        return;
    }
    const int idx = (blockIdx.x - 1) * BS + tid;
    if (idx < n_item_entries) {
        const float gf = -flat[VIBO_NUM_SCALARS + 2 * n_table + idx];          // d loss / d item_feat = -dLL/ditem
        const float m = mu[idx], l = lv[idx];
        const float g_mu = fmaf(beta, m, gf);
        const float half_sd = 0.5f * expf(0.5f * l);
        const float klg = (0.5f * beta) * (1.0f - expf(l));
        const float g_lv = fmaf(gf * half_sd, eps[idx], -klg);
        float pm = m, pl = l;
        mt_adam(pm, im[idx], iv[idx], g_mu, lr, bc1, bc2_sqrt);
        mt_adam(pl, im[n_item_entries + idx], iv[n_item_entries + idx], g_lv, lr, bc1, bc2_sqrt);
        mu[idx] = pm;
        lv[idx] = pl;
    }
}

static int mt_item_dim(const vibo_desc* d) { return d->irt_model == 1 ? 1 : (d->irt_model == 2 ? d->ability_dim + 1 : d->ability_dim + 2); }
static int mt_check(const vibo_desc* d, int H) {
    if (!d || d->abi_version != VIBO_ABI_VERSION) return -2;
    if (d->ability_dim < 1 || d->ability_dim > VIBO_MAX_ABILITY_DIM || d->num_item < 1 || d->num_person < 1) return -3;
    if (H < 1 || H > kMtMaxHidden) return -6;
    if (d->posterior != VIBO_POSTERIOR_GIVEN || d->n_flows != 0 || d->reg_mode != VIBO_REG_KL) return -6;
    return 0;
}

}  // namespace vibo

using namespace vibo;

extern "C" int64_t vibo_mtrain_param_floats(const vibo_desc* d, int hidden_dim) {
    if (!d || hidden_dim < 1) return 0;
    return moff(hidden_dim, 2 * d->ability_dim).total;
}

extern "C" int vibo_mtrain_prologue(const vibo_desc* d, int hidden_dim, const float* params, const float* item_mu,
                                    const float* item_logvar, float* eps_item, uint64_t seed, int draw_noise, float* eps_ability,
                                    uint32_t ability_stream_id, float* item_feat, float* uv, float* saved, float* kl_parts,
                                    int32_t* step_count, void* stream) {
    const int rc = mt_check(d, hidden_dim);
    if (rc) return rc;
    if (!params || !item_mu || !item_logvar || !eps_item || !item_feat || !uv || !saved || !kl_parts || !step_count) return -5;
    if (draw_noise && !eps_ability) return -5;
    const int D = mt_item_dim(d), n = d->num_item * D;
    const int item_blocks = (n + 255) / 256;
    const long long n_ab = draw_noise ? (long long)d->num_person * d->ability_dim : 0;
    const long long ab_blocks = ((n_ab + 3) / 4 + 255) / 256;
    hipLaunchKernelGGL(mt_prologue_kernel, dim3((unsigned)(1 + item_blocks + ab_blocks)), dim3(256), 0, (hipStream_t)stream, hidden_dim,
                       2 * d->ability_dim, d->num_item, D, params, item_mu, item_logvar, (const float*)eps_item, item_feat, uv, saved,
                       kl_parts, step_count, draw_noise ? 1 : 0, (uint32_t)seed, (uint32_t)(seed >> 32), eps_item, eps_ability, n_ab,
                       ability_stream_id, item_blocks);
    return (int)hipGetLastError();
}

extern "C" int vibo_mtrain_epilogue(const vibo_desc* d, int hidden_dim, const float* flat, const float* partials, int n_partials,
                                    float* grad_sums, const float* saved, const float* kl_parts, const float* eps_item,
                                    const float* beta, const float* lr, int32_t* step_count, float* params, float* adam_m,
                                    float* adam_v, float* item_mu, float* item_logvar, float* item_m, float* item_v, float* loss_out,
                                    void* stream) {
    const int rc = mt_check(d, hidden_dim);
    if (rc) return rc;
    if (!flat || !partials || !grad_sums || !saved || !kl_parts || !eps_item || !beta || !lr || !step_count || !params || !adam_m ||
        !adam_v || !item_mu || !item_logvar || !item_m || !item_v || !loss_out || n_partials < 1)
        return -5;
    const int H = hidden_dim, A2 = 2 * d->ability_dim;
    const int n_out = 2 * H + A2 * H + A2;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mt_reduce_kernel, dim3((n_out + 63) / 64), dim3(1024), 0, s, partials, grad_sums, n_partials, n_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int n = d->num_item * mt_item_dim(d);
    const long long n_table = (long long)d->num_person * A2;       // floats per table-gradient set of the VIBO_POSTERIOR_GIVEN call
    hipLaunchKernelGGL(mt_epilogue_kernel, dim3(1 + (n + kMtThreads - 1) / kMtThreads), dim3(kMtThreads), 0, s, H, A2, n, flat, grad_sums,
                       saved, kl_parts, eps_item, beta, lr, step_count, n_table, params, adam_m, adam_v, item_mu, item_logvar, item_m, item_v,
                       loss_out);
    return (int)hipGetLastError();
}
