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

// LDS scratch of the 2-row MLP.
//   STAGED (the row-split kernels, H <= kHookMaxHidden): W1 [H][H+1] | W2 [O][H+1] | h1 [2][H] | h2 [2][H] | table [2][O] | b1 | b2 -- the
//     weights are copied in coalesced first (rows padded by one float: the per-row reads below are conflict-free).  Read
//     straight from global memory, thread j streams row j and every wave load touches 64 cache lines: ~8 k line requests per
//     workgroup, which 256 workgroups at once turn into ~8 us of L2 queueing.
//   not STAGED (train_prologue_kernel, one workgroup, H <= kMaxHidden): h1 | h2 | table only, weights read from global memory.
// Either way row j's dot product is the same chain of fmaf's over k = 0 .. H-1: the two forms agree bit for bit.
constexpr int kHookMaxHidden = 64;
constexpr int kHookLdsFloats = (kHookMaxHidden + 2 * VIBO_MAX_ABILITY_DIM) * (kHookMaxHidden + 1) + 5 * kHookMaxHidden + 6 * VIBO_MAX_ABILITY_DIM;
constexpr int kHookScratchFloats = 4 * kMaxHidden + 4 * VIBO_MAX_ABILITY_DIM;       // (not STAGED)
template <bool STAGED>
struct HookLds {
    float *w1, *w2, *h1, *h2, *tab;
    const float *b1, *b2;
    int ld;
    __device__ __forceinline__ HookLds(float* s, const float* P, const int H, const int O) {
        const MlpOffsets o = mlp_offsets(H, O);
        if constexpr (STAGED) {
            ld = H + 1;
            w1 = s; w2 = s + H * ld; h1 = w2 + O * ld; h2 = h1 + 2 * H; tab = h2 + 2 * H;
            b1 = tab + 2 * O; b2 = b1 + H;
        } else {
            ld = H;
            w1 = const_cast<float*>(P) + o.w1; w2 = const_cast<float*>(P) + o.w2;
            h1 = s; h2 = s + 2 * kMaxHidden; tab = s + 4 * kMaxHidden;
            b1 = P + o.b1; b2 = P + o.b2;
        }
    }
};

// stage 0 (STAGED only): W1, W2 -> LDS, coalesced; then layer 0: the input of row r is the response value r in {0, 1}.
// (A workgroup barrier separates the stages.)
template <bool STAGED>
__device__ __forceinline__ void hook_mlp_layer0(const float* __restrict__ P, const int H, const int O, float* s, const int tid,
                                                const int nthr) {
    const MlpOffsets o = mlp_offsets(H, O);
    const HookLds<STAGED> L(s, P, H, O);
    if constexpr (STAGED) {
        // W1 | b1 | W2 | b2 are contiguous in P: one coalesced copy, eight loads in flight per thread before the first LDS store
        // (a load-store pair per loop trip costs a memory round trip each: 16 us for 16 trips)
        const int n = o.total - o.w1;
        for (int t0 = 0; t0 < n; t0 += 8 * nthr) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * nthr + tid;
                v[u] = t < n ? P[o.w1 + t] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * nthr + tid;
                if (t >= n) continue;
                float* dst;
                if (t < H * H) dst = L.w1 + (t / H) * L.ld + t % H;
                else if (t < H * H + H) dst = const_cast<float*>(L.b1) + (t - H * H);
                else if (t < H * H + H + O * H) { const int k = t - H * H - H; dst = L.w2 + (k / H) * L.ld + k % H; }
                else dst = const_cast<float*>(L.b2) + (t - H * H - H - O * H);
                *dst = v[u];
            }
        }
    }
    for (int t = tid; t < 2 * H; t += nthr) {
        const int r = t / H, j = t % H;
        L.h1[r * H + j] = elu(fmaf(P[o.w0 + j], (float)r, P[o.b0 + j]));
    }
}
template <bool STAGED>
__device__ __forceinline__ void hook_mlp_layer1(const float* __restrict__ P, const int H, const int O, float* s, const int tid,
                                                const int nthr) {
    const HookLds<STAGED> L(s, P, H, O);
    for (int t = tid; t < 2 * H; t += nthr) {
        const int r = t / H, j = t % H;
        float a = L.b1[j];
#pragma unroll 16
        for (int k = 0; k < H; ++k) a = fmaf(L.w1[j * L.ld + k], L.h1[r * H + k], a);
        L.h2[r * H + j] = elu(a);
    }
}
// layer 2 -> table (LDS copy in .tab; the writer also stores table [2][O] and the activations saved_h = h1 | h2)
template <bool STAGED>
__device__ __forceinline__ void hook_mlp_layer2(const float* __restrict__ P, const int H, const int O, float* s, const int tid,
                                                const int nthr, float* __restrict__ table, float* __restrict__ saved_h) {
    const HookLds<STAGED> L(s, P, H, O);
    for (int t = tid; t < 2 * O; t += nthr) {
        const int r = t / O, q = t % O;
        float a = L.b2[q];
#pragma unroll 16
        for (int k = 0; k < H; ++k) a = fmaf(L.w2[q * L.ld + k], L.h2[r * H + k], a);
        L.tab[t] = a;
        if (table) table[t] = a;
    }
    if (saved_h) {
        for (int t = tid; t < 2 * H; t += nthr) {
            saved_h[t] = L.h1[t];
            saved_h[2 * H + t] = L.h2[t];
        }
    }
}
template <bool STAGED>
__device__ __forceinline__ const float* hook_tab(float* s, const float* P, const int H, const int O) { return HookLds<STAGED>(s, P, H, O).tab; }

// entry idx of the [I][D] item sample (models.py:506-510)
__device__ __forceinline__ float item_sample(const float m, const float l, const float e) { return fmaf(expf(0.5f * l), e, m); }
__device__ __forceinline__ float item_kl_term(const float m, const float l) { return -0.5f * (1.0f + l - m * m - expf(l)); }
__device__ __forceinline__ float hook_item(const TrainHook& th, const float* __restrict__ item_raw, const size_t idx) {
    return th.mlp ? item_sample(th.item_mu[idx], th.item_lv[idx], th.eps_item[idx]) : item_raw[idx];
}

// The item side: item_feat for every entry and kl_parts[b] = the KL terms of entries [256 b, 256 b + 256) summed as four
// 64-lane wave totals, ((t0 + t1) + t2) + t3 -- the order train_prologue_kernel's 256-thread item blocks produce.
// Wave-granular (no LDS, no barrier): block b belongs to wave `gwave` of the launch's `total_waves` waves, so the ~36 blocks
// of a 1 000-item model cost one pass of ~2 us in 36 different workgroups instead of five serial passes in one.
__device__ __forceinline__ void hook_item_side(const TrainHook& th, const int n_entries, const int gwave, const int lane,
                                               const int total_waves) {
    const int n_blocks = (n_entries + 255) / 256;
    for (int b = gwave; b < n_blocks; b += total_waves) {
        float m[4], l[4], e[4], t[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {           // (all twelve loads first)
            const int idx = 256 * b + 64 * sub + lane;
            const bool ok = idx < n_entries;
            m[sub] = ok ? th.item_mu[idx] : 0.f;
            l[sub] = ok ? th.item_lv[idx] : 0.f;
            e[sub] = ok ? th.eps_item[idx] : 0.f;
        }
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int idx = 256 * b + 64 * sub + lane;
            float kl = 0.f;
            if (idx < n_entries) {
                th.item_feat[idx] = item_sample(m[sub], l[sub], e[sub]);
                kl = item_kl_term(m[sub], l[sub]);
            }
            t[sub] = wave_total(kl);
        }
        if (lane == 0) th.kl_parts[b] = t[0] + t[1] + t[2] + t[3];
    }
}

}  // namespace vibo
