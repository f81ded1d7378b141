// vibo_train_hook.hpp -- the O(I) head of a plain VIBO train step (item sample, item KL, the 2-row encoder MLP) as device
// routines shared by train_prologue_kernel (its own launch) and by the row-split ELBO kernels, which run it in their own
// prologue when ElboParams::th is set ("train hook": one launch fewer per step, and no launch latency between the MLP and
// the kernel that consumes its table).  Both forms execute the same statements in the same order, so they agree bit for
// bit (tests/test_gpu_trainer.py::test_folded_step_equals_the_unfolded_step).
//
// Reference statements (models.py:356-361, 575-582, 713-726, 506-510; utils.py:85-88):
//     item_feat = item_mu + exp(0.5 item_logvar) * eps_item
//     KL_item   = sum -0.5 (1 + logvar - mu^2 - exp(logvar))
//     table[c]  = W2 . elu(W1 . elu(W0 c + b0) + b1) + b2        c in {0, 1}
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vibo_device.hpp"
#include "vibo_params.hpp"

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

// LDS scratch of the 2-row MLP: h1 [2][H] | h2 [2][H] | table [2][O]   (floats; H <= kMaxHidden, O <= 2 VIBO_MAX_ABILITY_DIM)
constexpr int kHookScratchFloats = 4 * kMaxHidden + 4 * VIBO_MAX_ABILITY_DIM;
__device__ __forceinline__ float* hook_h1(float* s) { return s; }
__device__ __forceinline__ float* hook_h2(float* s) { return s + 2 * kMaxHidden; }
__device__ __forceinline__ float* hook_tab(float* s) { return s + 4 * kMaxHidden; }

// layer 0: the input of row r is the response value r in {0, 1}.  (A workgroup barrier separates the stages.)
__device__ __forceinline__ void hook_mlp_layer0(const float* __restrict__ P, const int H, const int O, float* s, const int tid,
                                                const int nthr) {
    const MlpOffsets o = mlp_offsets(H, O);
    for (int t = tid; t < 2 * H; t += nthr) {
        const int r = t / H, j = t % H;
        hook_h1(s)[r * H + j] = elu(fmaf(P[o.w0 + j], (float)r, P[o.b0 + j]));
    }
}
__device__ __forceinline__ void hook_mlp_layer1(const float* __restrict__ P, const int H, const int O, float* s, const int tid,
                                                const int nthr) {
    const MlpOffsets o = mlp_offsets(H, O);
    const float* h1 = hook_h1(s);
    for (int t = tid; t < 2 * H; t += nthr) {
        const int r = t / H, j = t % H;
        float a = P[o.b1 + j];
#pragma unroll 16
        for (int k = 0; k < H; ++k) a = fmaf(P[o.w1 + j * H + k], h1[r * H + k], a);
        hook_h2(s)[r * H + j] = elu(a);
    }
}
// layer 2 -> table (LDS copy in hook_tab(s); the writer also stores table [2][O] and the activations saved_h = h1 | h2)
__device__ __forceinline__ void hook_mlp_layer2(const float* __restrict__ P, const int H, const int O, float* s, const int tid,
                                                const int nthr, float* __restrict__ table, float* __restrict__ saved_h) {
    const MlpOffsets o = mlp_offsets(H, O);
    const float* h2 = hook_h2(s);
    for (int t = tid; t < 2 * O; t += nthr) {
        const int r = t / O, q = t % O;
        float a = P[o.b2 + q];
#pragma unroll 16
        for (int k = 0; k < H; ++k) a = fmaf(P[o.w2 + q * H + k], h2[r * H + k], a);
        hook_tab(s)[t] = a;
        if (table) table[t] = a;
    }
    if (saved_h) {
        for (int t = tid; t < 2 * H; t += nthr) {
            saved_h[t] = hook_h1(s)[t];
            saved_h[2 * H + t] = h2[t];
        }
    }
}

// entry idx of the [I][D] item sample (models.py:506-510)
__device__ __forceinline__ float item_sample(const float m, const float l, const float e) { return fmaf(expf(0.5f * l), e, m); }
__device__ __forceinline__ float item_kl_term(const float m, const float l) { return -0.5f * (1.0f + l - m * m - expf(l)); }
__device__ __forceinline__ float hook_item(const TrainHook& th, const float* __restrict__ item_raw, const size_t idx) {
    return th.mlp ? item_sample(th.item_mu[idx], th.item_lv[idx], th.eps_item[idx]) : item_raw[idx];
}

// The item side as ONE workgroup runs it: item_feat for every entry and kl_parts[b] = the KL terms of entries
// [256 b, 256 b + 256) summed as four 64-lane wave totals, ((t0 + t1) + t2) + t3 -- the order train_prologue_kernel's
// 256-thread item blocks produce.  Wave-granular: no LDS, no barrier.
__device__ __forceinline__ void hook_item_side(const TrainHook& th, const int n_entries, const int wave, const int lane,
                                               const int n_waves) {
    const int n_blocks = (n_entries + 255) / 256;
    for (int b = wave; b < n_blocks; b += n_waves) {
        float t[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int idx = 256 * b + 64 * sub + lane;
            float kl = 0.f;
            if (idx < n_entries) {
                const float m = th.item_mu[idx], l = th.item_lv[idx];
                th.item_feat[idx] = item_sample(m, l, th.eps_item[idx]);
                kl = item_kl_term(m, l);
            }
            t[sub] = wave_total(kl);
        }
        if (lane == 0) th.kl_parts[b] = t[0] + t[1] + t[2] + t[3];
    }
}

}  // namespace vibo
