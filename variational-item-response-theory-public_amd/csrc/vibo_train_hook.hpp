// vibo_train_hook.hpp -- the O(I) head of a plain VIBO train step (item sample, item KL, the 2-row encoder MLP) as device
// routines shared by train_prologue_kernel (the first step of a run, and every step of the four-launch form) and by
// train_epilogue_fused_kernel, which leaves the NEXT step's head behind (vibo_trainer.hip).  Both execute the same statements
// in the same order, so the two forms of the step agree bit for bit
// (tests/test_gpu_trainer.py::test_folded_step_equals_the_unfolded_step).
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

// Parameter layout W0 [H] | b0 [H] | W1 [H][ld] | b1 [H] | W2 [O][ld] | b2 [O]: ld = H in the caller's flat buffer; the epilogue's
// LDS copy pads the matrix rows by one float (ld = H + 1) so that "thread j walks row j" is conflict-free.
struct MlpOffsets {
    int w0, b0, w1, b1, w2, b2, total, ld;
};
__host__ __device__ inline MlpOffsets mlp_offsets(int H, int O, int ld = 0) {
    MlpOffsets o;
    o.ld = ld > 0 ? ld : H;
    o.w0 = 0; o.b0 = H; o.w1 = 2 * H; o.b1 = o.w1 + H * o.ld; o.w2 = o.b1 + H; o.b2 = o.w2 + O * o.ld; o.total = o.b2 + O;
    return o;
}
// index of flat parameter k (ld = H layout) in the layout `t`
__device__ __forceinline__ int mlp_reindex(const int k, const int H, const int O, const MlpOffsets& t) {
    const MlpOffsets o = mlp_offsets(H, O);
    if (k < o.w1) return k;
    if (k < o.b1) return t.w1 + ((k - o.w1) / H) * t.ld + (k - o.w1) % H;
    if (k < o.w2) return t.b1 + (k - o.b1);
    if (k < o.b2) return t.w2 + ((k - o.w2) / H) * t.ld + (k - o.w2) % H;
    return t.b2 + (k - o.b2);
}

__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : expm1f(x); }

// The 2-row MLP forward of ONE workgroup in three stages (a workgroup barrier between them).  h1, h2: [2][H] in LDS;
// P / o: the parameters and their layout (global memory, or the epilogue's padded LDS copy).
// Row j's dot product is one chain of fmaf's over k = 0 .. H-1 wherever it runs.
__device__ __forceinline__ void mlp2_layer0(const float* P, const MlpOffsets o, const int H, const int O, float* h1, const int tid,
                                            const int nthr) {
    for (int t = tid; t < 2 * H; t += nthr) {      // the input of row r is the response value r in {0, 1}
        const int r = t / H, j = t % H;
        h1[r * H + j] = elu(fmaf(P[o.w0 + j], (float)r, P[o.b0 + j]));
    }
}
__device__ __forceinline__ void mlp2_layer1(const float* P, const MlpOffsets o, const int H, const int O, const float* h1, float* h2,
                                            const int tid, const int nthr) {
    for (int t = tid; t < 2 * H; t += nthr) {
        const int r = t / H, j = t % H;
        float a = P[o.b1 + j];
#pragma unroll 16
        for (int k = 0; k < H; ++k) a = fmaf(P[o.w1 + j * o.ld + k], h1[r * H + k], a);
        h2[r * H + j] = elu(a);
    }
}
// layer 2 -> table [2][O]; the activations are kept for the backward: saved_h = h1 | h2
__device__ __forceinline__ void mlp2_layer2(const float* P, const MlpOffsets o, const int H, const int O, const float* h1, const float* h2,
                                            const int tid, const int nthr, float* table, float* saved_h) {
    for (int t = tid; t < 2 * O; t += nthr) {
        const int r = t / O, q = t % O;
        float a = P[o.b2 + q];
#pragma unroll 16
        for (int k = 0; k < H; ++k) a = fmaf(P[o.w2 + q * o.ld + k], h2[r * H + k], a);
        table[t] = a;
    }
    for (int t = tid; t < 2 * H; t += nthr) {
        saved_h[t] = h1[t];
        saved_h[2 * H + t] = h2[t];
    }
}

// entry idx of the [I][D] item sample (models.py:506-510) and its KL term (utils.py:85-88)
__device__ __forceinline__ float item_sample(const float m, const float l, const float e) { return fmaf(expf(0.5f * l), e, m); }
__device__ __forceinline__ float item_kl_term(const float m, const float l) { return -0.5f * (1.0f + l - m * m - expf(l)); }

// Item entries are walked dimension-major -- entry k = (dim k / I, item k % I), the order of the ELBO kernel's gradient records --
// in groups of 64 (one wave): kl_parts[g] = the wave total of the KL terms of entries [64 g, 64 g + 64).
__device__ __forceinline__ int item_entry_index(const int k, const int I, const int D) { return (k % I) * D + k / I; }
constexpr int kKlGroup = 64;
__host__ __device__ inline int kl_part_count(const int n_entries) { return (n_entries + kKlGroup - 1) / kKlGroup; }

}  // namespace vibo
