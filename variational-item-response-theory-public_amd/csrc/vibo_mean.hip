// vibo_mean.hip -- the --ability-merge mean encoder (reference models.py:584-594, 631-650) per person.
//
// Reference: hid_pi = elu(mlp1([x_pi]))  (Linear(1,H) -> ELU -> Linear(H,H), then ELU), hid_mean_p = mean over the
// person's observed items, (mu_p | logvar_p) = mlp2(hid_mean_p) with mlp2 = Linear(H,H) -> ELU -> Linear(H,2A).
// A Bernoulli response takes two values, so hid_mean_p = h0 + w_p (h1 - h0) with w_p = n_correct / n_observed, and the
// first layer of mlp2 is affine in w_p:  z_p = u + w_p v,  u = W1 h0 + b1,  v = W1 (h1 - h0)   (u, v: [H], computed by the
// caller with autograd -- two H-vectors).  What is left per person is  a = elu(u + w v),  out = W2 a + b2  (2A x H MACs):
// the "[B,64] x [64,64]" GEMM of the naive formulation does not survive the collapse, and 2A x H MACs per person against
// the 5 kB of response row the step streams anyway is not matrix-core work.
//
//   forward : thread = person.   u, v, W2, b2 are wave-uniform (scalar loads).
//   backward: wave = a slice of persons, lane = hidden unit k (H <= 256 in chunks of 64): lane k keeps W2[:, k] and
//             accumulates d/du[k], d/dv[k], d/dW2[:, k]; lanes j < 2A accumulate d/db2[j].  One partial record per
//             wave, summed by the caller in a fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"

namespace vibo {

constexpr int kMeanMaxHidden = 256;

__device__ __forceinline__ float elu1(float z) { return z > 0.f ? z : expm1f(z); }

__global__ __launch_bounds__(256) void mean_encoder_fwd_kernel(const int* __restrict__ counts, const float* __restrict__ u,
                                                               const float* __restrict__ v, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, float* __restrict__ post, long long B,
                                                               int H, int A2) {
    // the weights through LDS first (broadcast reads in the loop: with per-iteration global loads a 16-person call was a chain of
    // 64 dependent round trips, 18 - 43 us)
    __shared__ float su[kMeanMaxHidden], sv[kMeanMaxHidden], sw[2 * VIBO_MAX_ABILITY_DIM * kMeanMaxHidden];
    for (int k = threadIdx.x; k < H; k += 256) { su[k] = u[k]; sv[k] = v[k]; }
    for (int e = threadIdx.x; e < A2 * H; e += 256) sw[e] = w2[e];
    __syncthreads();
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= B) return;
    const int c = counts[p];
    const float w = (float)(c >> 16) / (float)(c & 0xffff);        // 0 observed items: 0/0 = NaN, as the reference's empty mean
    float out[2 * VIBO_MAX_ABILITY_DIM];
#pragma unroll
    for (int j = 0; j < 2 * VIBO_MAX_ABILITY_DIM; ++j) out[j] = j < A2 ? b2[j] : 0.f;
    for (int k = 0; k < H; ++k) {
        const float a = elu1(fmaf(w, sv[k], su[k]));
#pragma unroll
        for (int j = 0; j < 2 * VIBO_MAX_ABILITY_DIM; ++j)
            if (j < A2) out[j] = fmaf(sw[j * H + k], a, out[j]);
    }
#pragma unroll
    for (int j = 0; j < 2 * VIBO_MAX_ABILITY_DIM; ++j)
        if (j < A2) post[p * A2 + j] = out[j];
}

// the same for small minibatches: a wave per person, lane = hidden unit (the thread-per-person loop above is 64 x (expm1 + 2A
// fmas) in a row for a handful of active lanes: 18 - 44 us for 16 persons).  Its sums over the hidden units are wave reductions,
// so the two kernels agree to fp32 rounding, not bit for bit; the choice depends on the person count only.
template <int KCH>
__global__ __launch_bounds__(256) void mean_encoder_fwd_wave_kernel(const int* __restrict__ counts, const float* __restrict__ u,
                                                                    const float* __restrict__ v, const float* __restrict__ w2,
                                                                    const float* __restrict__ b2, float* __restrict__ post, long long B,
                                                                    int H, int A2) {
    const int lane = threadIdx.x & 63;
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= B) return;
    const int c = counts[p];
    const float w = (float)(c >> 16) / (float)(c & 0xffff);
    float a[KCH];
#pragma unroll
    for (int ch = 0; ch < KCH; ++ch) {
        const int k = ch * 64 + lane;
        a[ch] = k < H ? elu1(fmaf(w, v[k], u[k])) : 0.f;
    }
    for (int j = 0; j < A2; ++j) {
        float t = 0.f;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int k = ch * 64 + lane;
            t = fmaf(k < H ? w2[j * H + k] : 0.f, a[ch], t);
        }
        t = wave_total(t);
        if (lane == 0) post[p * A2 + j] = t + b2[j];
    }
}

// partial record of a wave: [ d/du (H) | d/dv (H) | d/dW2 (A2 x H, row-major) | d/db2 (A2) ]
template <int KCH>
__global__ __launch_bounds__(256) void mean_encoder_bwd_kernel(const int* __restrict__ counts, const float* __restrict__ u,
                                                               const float* __restrict__ v, const float* __restrict__ w2,
                                                               const float* __restrict__ gpost, float* __restrict__ part,
                                                               long long B, int H, int A2,
                                                               // sets != null: d loss / d posterior = -sets[0] + beta sets[1]
                                                               // (the two gradient sets of a VIBO_POSTERIOR_GIVEN call), gpost unused
                                                               const float* __restrict__ sets, const float* __restrict__ beta_p) {
    const int lane = threadIdx.x & 63;
    const float beta = sets ? *beta_p : 0.f;
    const long long set1 = B * A2;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    float uu[KCH], vv[KCH], wcol[KCH][2 * VIBO_MAX_ABILITY_DIM];
    float gu[KCH], gv[KCH], gw[KCH][2 * VIBO_MAX_ABILITY_DIM];
    float gb = 0.f;
#pragma unroll
    for (int ch = 0; ch < KCH; ++ch) {
        const int k = ch * 64 + lane;
        const bool ok = k < H;
        uu[ch] = ok ? u[k] : 0.f;
        vv[ch] = ok ? v[k] : 0.f;
        gu[ch] = gv[ch] = 0.f;
#pragma unroll
        for (int j = 0; j < 2 * VIBO_MAX_ABILITY_DIM; ++j) {
            wcol[ch][j] = (ok && j < A2) ? w2[j * H + k] : 0.f;
            gw[ch][j] = 0.f;
        }
    }
    // persons in a fixed, contiguous slice per wave
    const long long per = (B + n_waves - 1) / n_waves;
    const long long p0 = wave * per, p1 = (p0 + per < B) ? p0 + per : B;
    for (long long p = p0; p < p1; ++p) {
        const int c = counts[p];
        const float w = (float)(c >> 16) / (float)(c & 0xffff);
        float g[2 * VIBO_MAX_ABILITY_DIM];
#pragma unroll
        for (int j = 0; j < 2 * VIBO_MAX_ABILITY_DIM; ++j)                                             // wave-uniform
            g[j] = j < A2 ? (sets ? fmaf(beta, sets[set1 + p * A2 + j], -sets[p * A2 + j]) : gpost[p * A2 + j]) : 0.f;
        if (lane < A2) gb += sets ? fmaf(beta, sets[set1 + p * A2 + lane], -sets[p * A2 + lane]) : gpost[p * A2 + lane];
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const float z = fmaf(w, vv[ch], uu[ch]);
            const float a = elu1(z);
            const float da = z > 0.f ? 1.0f : a + 1.0f;
            float ga = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * VIBO_MAX_ABILITY_DIM; ++j) {
                if (j < A2) {
                    ga = fmaf(wcol[ch][j], g[j], ga);
                    gw[ch][j] = fmaf(g[j], a, gw[ch][j]);
                }
            }
            const float gz = ga * da;
            gu[ch] += gz;
            gv[ch] = fmaf(w, gz, gv[ch]);
        }
    }
    float* rec = part + (size_t)wave * (2 * H + A2 * H + A2);
#pragma unroll
    for (int ch = 0; ch < KCH; ++ch) {
        const int k = ch * 64 + lane;
        if (k < H) {
            rec[k] = gu[ch];
            rec[H + k] = gv[ch];
#pragma unroll
            for (int j = 0; j < 2 * VIBO_MAX_ABILITY_DIM; ++j)
                if (j < A2) rec[2 * H + j * H + k] = gw[ch][j];
        }
    }
    if (lane < A2) rec[2 * H + A2 * H + lane] = gb;
}

static int mean_check(const vibo_desc* d, int hidden) {
    if (!d || d->abi_version != VIBO_ABI_VERSION) return -2;
    if (d->num_person < 1 || d->ability_dim < 1 || d->ability_dim > VIBO_MAX_ABILITY_DIM) return -3;
    if (hidden < 1 || hidden > kMeanMaxHidden) return -6;
    return 0;
}

}  // namespace vibo

using namespace vibo;

extern "C" int vibo_mean_encoder_partials(const vibo_desc* d) {
    if (!d || d->num_person < 1) return 0;
    int dev = 0, n = 0;
    const int cus = (hipGetDevice(&dev) == hipSuccess &&
                     hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    long long waves = (long long)cus * 4 * 4;             // 4 workgroups of 4 waves per CU
    const long long need = ((long long)d->num_person + 63) / 64;      // at least 64 persons per wave
    if (waves > need) waves = need;
    if (waves < 4) waves = 4;
    return (int)((waves + 3) / 4 * 4);
}

extern "C" int vibo_mean_encoder_forward(const vibo_desc* d, int hidden, const int32_t* counts, const float* u, const float* v,
                                         const float* w2, const float* b2, float* posterior, void* stream) {
    const int rc = mean_check(d, hidden);
    if (rc) return rc;
    if (!counts || !u || !v || !w2 || !b2 || !posterior) return -5;
    const long long B = d->num_person;
    const int A2 = 2 * d->ability_dim;
    hipStream_t s = (hipStream_t)stream;
    if (B <= 2048) {                          // launch-bound minibatches: a wave per person
        const dim3 grid((unsigned)((B + 3) / 4)), block(256);
        switch ((hidden + 63) / 64) {
            case 1: hipLaunchKernelGGL(mean_encoder_fwd_wave_kernel<1>, grid, block, 0, s, counts, u, v, w2, b2, posterior, B, hidden, A2); break;
            case 2: hipLaunchKernelGGL(mean_encoder_fwd_wave_kernel<2>, grid, block, 0, s, counts, u, v, w2, b2, posterior, B, hidden, A2); break;
            case 3: hipLaunchKernelGGL(mean_encoder_fwd_wave_kernel<3>, grid, block, 0, s, counts, u, v, w2, b2, posterior, B, hidden, A2); break;
            default: hipLaunchKernelGGL(mean_encoder_fwd_wave_kernel<4>, grid, block, 0, s, counts, u, v, w2, b2, posterior, B, hidden, A2); break;
        }
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(mean_encoder_fwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, counts, u, v, w2, b2, posterior, B,
                       hidden, A2);
    return (int)hipGetLastError();
}

static int mean_backward_launch(const vibo_desc* d, int hidden, const int32_t* counts, const float* u, const float* v, const float* w2,
                                const float* grad_posterior, const float* sets, const float* beta, float* partials, int n_partials,
                                void* stream) {
    const int rc = mean_check(d, hidden);
    if (rc) return rc;
    if (!counts || !u || !v || !w2 || (!grad_posterior && !sets) || (sets && !beta) || !partials) return -5;
    if (n_partials < 4 || n_partials % 4 != 0) return -3;
    const int kch = (hidden + 63) / 64;
    const dim3 grid(n_partials / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    const long long B = d->num_person;
    const int A2 = 2 * d->ability_dim;
    switch (kch) {
        case 1: hipLaunchKernelGGL(mean_encoder_bwd_kernel<1>, grid, block, 0, s, counts, u, v, w2, grad_posterior, partials, B, hidden, A2, sets, beta); break;
        case 2: hipLaunchKernelGGL(mean_encoder_bwd_kernel<2>, grid, block, 0, s, counts, u, v, w2, grad_posterior, partials, B, hidden, A2, sets, beta); break;
        case 3: hipLaunchKernelGGL(mean_encoder_bwd_kernel<3>, grid, block, 0, s, counts, u, v, w2, grad_posterior, partials, B, hidden, A2, sets, beta); break;
        default: hipLaunchKernelGGL(mean_encoder_bwd_kernel<4>, grid, block, 0, s, counts, u, v, w2, grad_posterior, partials, B, hidden, A2, sets, beta); break;
    }
    return (int)hipGetLastError();
}

extern "C" int vibo_mean_encoder_backward(const vibo_desc* d, int hidden, const int32_t* counts, const float* u, const float* v,
                                          const float* w2, const float* grad_posterior, float* partials, int n_partials,
                                          void* stream) {
    return mean_backward_launch(d, hidden, counts, u, v, w2, grad_posterior, nullptr, nullptr, partials, n_partials, stream);
}

extern "C" int vibo_mean_encoder_backward_sets(const vibo_desc* d, int hidden, const int32_t* counts, const float* u, const float* v,
                                               const float* w2, const float* grad_sets, const float* beta, float* partials,
                                               int n_partials, void* stream) {
    return mean_backward_launch(d, hidden, counts, u, v, w2, nullptr, grad_sets, beta, partials, n_partials, stream);
}
