// vibo_ctrainer.hip -- the O(I) part of a VIBO train step for --conditional-posterior and / or --n-norm-flows
// (product-of-experts encoder, IRT decoder) as native kernels: what vibo_trainer.hip is for the plain model.
//
// Reference step (vibo.py:243-268) around the fused ELBO kernel:
//   item sample (models.py:361-362, 506-510) -> item-side planar flows (flows.py:21-66, models.py:346-348)
//   -> expert table = encoder MLP on the rows [c, item_i] (models.py:666-710; 2 rows [c] without the conditional posterior)
//   -> vibo_elbo_fwd_bwd -> loss (models.py:380-443) -> backward through table MLP / flows / sample -> Adam (vibo.py:221).
// The module path runs this as ~190 PyTorch launches per step (2.2 ms at 16 persons); here it is five launches around
// the ELBO call, all deterministic (fixed-order partial records), so a captured hipGraph contains no PyTorch autograd node.
//
// Flat parameter buffer (`params`, and the Adam moments in the same layout):
//   W0 [H][xin] | b0 [H] | W1 [H][H] | b1 [H] | W2 [O][H] | b2 [O] | ability flows F x (u[A] | w[A] | b) | item flows F x (u[D] | w[D] | b)
// with xin = 1 + D (conditional) or 1, O = 2 A.  The nn.Parameters of the drop-in module are views of it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_philox.hpp"

namespace vibo {

constexpr int kCtMaxDim = 10;          // item_feat_dim = ability_dim + 2 <= 10
constexpr int kCtItems = 256;          // items per workgroup of the item-side kernels

__host__ __device__ inline int ct_rows_per_wave(int rows) {
    const int r = (rows + 511) / 512;
    return r < 8 ? 8 : r;
}

struct CtLayout {
    int H, O, xin, A, D, F, I, cond, rows;
    int w0, b0, w1, b1, w2, b2, n_mlp, fa, fi, n_par;      // offsets into the flat parameter buffer
    int n_rb, n_ib;                                        // row blocks, item blocks
    // scratch (floats)
    size_t s_pack, s_tanh, s_parts, s_gx, s_mrec, s_frec, s_total;
};
__host__ __device__ inline CtLayout ct_layout(int I, int A, int irt, int cond, int F, int H) {
    CtLayout L;
    L.H = H; L.A = A; L.O = 2 * A; L.D = irt == 1 ? 1 : irt == 2 ? A + 1 : A + 2; L.F = F; L.I = I; L.cond = cond;
    L.xin = 1 + (cond ? L.D : 0);
    L.rows = 2 * (cond ? I : 1);
    L.w0 = 0; L.b0 = L.w0 + H * L.xin; L.w1 = L.b0 + H; L.b1 = L.w1 + H * H; L.w2 = L.b1 + H; L.b2 = L.w2 + L.O * H;
    L.n_mlp = L.b2 + L.O;
    L.fa = L.n_mlp; L.fi = L.fa + F * (2 * A + 1); L.n_par = L.fi + F * (2 * L.D + 1);
    L.n_rb = (L.rows + ct_rows_per_wave(L.rows) - 1) / ct_rows_per_wave(L.rows);
    L.n_ib = (I + kCtItems - 1) / kCtItems;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    L.s_pack = take((size_t)VIBO_MAX_FLOWS * (2 * kCtMaxDim + 2));
    L.s_tanh = take((size_t)I * (F > 0 ? F : 1));
    L.s_parts = take((size_t)L.n_ib * 4);
    L.s_gx = take((size_t)L.rows * kCtMaxDim);
    L.s_mrec = take((size_t)L.n_rb * L.n_mlp);
    L.s_frec = take((size_t)L.n_ib * (F > 0 ? F : 1) * (2 * L.D + 1));
    L.s_total = o;
    return L;
}

__device__ __forceinline__ float ct_elu(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float ct_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // F.softplus (threshold 20)
__device__ __forceinline__ float ct_tanh(float a) {
    return 1.0f - 2.0f / (1.0f + expf(2.0f * fminf(fmaxf(a, -15.f), 15.f)));      // (the flow kernels' form: vibo_flow.hip)
}

// uhat | w | b | w . uhat of one planar flow from (u | w | b)   (flows.py:23-26)
__device__ inline void ct_pack_flow(const float* __restrict__ p, int dim, float* __restrict__ out /* [2 dim + 2] */) {
    float s = 0.f, ww = 0.f;
    for (int d = 0; d < dim; ++d) { s = fmaf(p[dim + d], p[d], s); ww = fmaf(p[dim + d], p[dim + d], ww); }
    const float k = (ct_softplus(s) - 1.0f - s) / ww;
    float c = 0.f;
    for (int d = 0; d < dim; ++d) {
        const float uh = fmaf(k, p[dim + d], p[d]);
        out[d] = uh;
        out[dim + d] = p[dim + d];
        c = fmaf(p[dim + d], uh, c);
    }
    out[2 * dim] = p[2 * dim];
    out[2 * dim + 1] = c;
}
// backward of ct_pack_flow: (g_uhat | g_w at fixed uhat | g_b) -> (g_u | g_w | g_b)
__device__ inline void ct_unpack_flow_grad(const float* __restrict__ p, int dim, const float* __restrict__ g /* [2 dim + 1] */,
                                           float* __restrict__ out /* [2 dim + 1] */) {
    float s = 0.f, ww = 0.f, gw_dot = 0.f;
    for (int d = 0; d < dim; ++d) {
        s = fmaf(p[dim + d], p[d], s);
        ww = fmaf(p[dim + d], p[dim + d], ww);
        gw_dot = fmaf(g[d], p[dim + d], gw_dot);             // g_uhat . w
    }
    const float k = (ct_softplus(s) - 1.0f - s) / ww;
    const float dk_ds = (1.0f / (1.0f + expf(-s)) - 1.0f) / ww;   // (sigmoid(s) - 1) / |w|^2
    for (int d = 0; d < dim; ++d) {
        out[d] = fmaf(gw_dot * dk_ds, p[dim + d], g[d]);
        out[dim + d] = g[dim + d] + k * g[d] + gw_dot * (dk_ds * p[d] - 2.0f * k / ww * p[dim + d]);
    }
    out[2 * dim] = g[2 * dim];
}

__device__ __forceinline__ void ct_adam(float& p, float& m, float& v, const float g, const float lr, const float bc1, const float bc2_sqrt) {
    m = 0.9f * m + 0.1f * g;                       // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = 0.999f * v + 0.001f * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + 1e-8f;
    p -= (lr / bc1) * (m / denom);
}

// ---------------------------------------------------------------------------
// prologue: block 0 = flow packing (+ step counter); blocks 1..n_ib = items (sample, flows forward, partial sums);
// further blocks = ability noise
// ---------------------------------------------------------------------------
struct CtProParams {
    CtLayout L;
    const float* params; const float* mu; const float* lv;
    float* eps;                  // item noise: read (gen == 0) or written (gen != 0)
    float* item_feat; float* item_k; float* flow_packed; float* scratch;
    int32_t* step_count;
    int gen; uint32_t seed_lo, seed_hi;
    float* eps_ab; long long n_ab; uint32_t ab_stream;
};

__global__ __launch_bounds__(kCtItems) void ct_prologue_kernel(const CtProParams q) {
    const CtLayout& L = q.L;
    __shared__ float pk[VIBO_MAX_FLOWS][2 * kCtMaxDim + 2];
    __shared__ float red[4][3];
    const int tid = threadIdx.x;
    const int A = L.A, D = L.D, F = L.F;
    if (blockIdx.x == 0) {
        if (tid == 0) q.step_count[0] += 1;
        if (tid < F) {                               // ability flows -> the ELBO kernel's [F][2A+1] = uhat | w | b
            float o[2 * VIBO_MAX_ABILITY_DIM + 2];
            ct_pack_flow(q.params + L.fa + tid * (2 * A + 1), A, o);
            for (int e = 0; e < 2 * A + 1; ++e) q.flow_packed[tid * (2 * A + 1) + e] = o[e];
        }
        return;
    }
    if ((int)blockIdx.x > L.n_ib) {                  // ability noise (stream ab_stream), 4 normals per thread
        const long long g = (long long)(blockIdx.x - 1 - L.n_ib) * kCtItems + tid;
        if (4 * g < q.n_ab) store_normal4(q.eps_ab, q.n_ab, g, philox_normal4(g, (uint32_t)q.step_count[1], q.ab_stream, q.seed_lo, q.seed_hi));
        return;
    }
    if (tid < F) {
        ct_pack_flow(q.params + L.fi + tid * (2 * D + 1), D, pk[tid]);
        if (blockIdx.x == 1)                         // kept for the backward
            for (int e = 0; e < 2 * D + 2; ++e) q.scratch[L.s_pack + tid * (2 * kCtMaxDim + 2) + e] = pk[tid][e];
    }
    __syncthreads();
    const int i = (blockIdx.x - 1) * kCtItems + tid;
    float kl = 0.f, lq = 0.f, lp = 0.f;
    if (i < L.I) {
        float z[kCtMaxDim];
#pragma unroll
        for (int d = 0; d < kCtMaxDim; ++d) {
            z[d] = 0.f;
            if (d < D) {
                const int idx = i * D + d;
                const float m = q.mu[idx], l = q.lv[idx];
                float e;
                if (q.gen) { e = philox_normal1(idx, (uint32_t)q.step_count[1], 0u, q.seed_lo, q.seed_hi); q.eps[idx] = e; }
                else e = q.eps[idx];
                z[d] = fmaf(expf(0.5f * l), e, m);
                q.item_feat[idx] = z[d];
                kl += -0.5f * (1.0f + l - m * m - expf(l));
                lq += -0.5f * kLog2Pi - 0.5f * l - 0.5f * e * e;          // log N(item_feat; mu, exp(lv)) at the sample
            }
        }
        for (int f = 0; f < F; ++f) {
            float a = pk[f][2 * D];
#pragma unroll
            for (int d = 0; d < kCtMaxDim; ++d)
                if (d < D) a = fmaf(pk[f][D + d], z[d], a);
            const float t = ct_tanh(a);
            q.scratch[L.s_tanh + (size_t)i * F + f] = t;
            lq -= logf(fabsf(1.0f + (1.0f - t * t) * pk[f][2 * D + 1]) + 1e-8f);      // - log|det J|
#pragma unroll
            for (int d = 0; d < kCtMaxDim; ++d)
                if (d < D) z[d] = fmaf(pk[f][d], t, z[d]);
        }
#pragma unroll
        for (int d = 0; d < kCtMaxDim; ++d)
            if (d < D) {
                q.item_k[i * D + d] = z[d];
                lp += -0.5f * kLog2Pi - 0.5f * z[d] * z[d];
            }
    }
    kl = wave_total(kl); lq = wave_total(lq); lp = wave_total(lp);
    if ((tid & 63) == 0) { red[tid >> 6][0] = kl; red[tid >> 6][1] = lq; red[tid >> 6][2] = lp; }
    __syncthreads();
    if (tid < 3) q.scratch[L.s_parts + (size_t)(blockIdx.x - 1) * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ---------------------------------------------------------------------------
// table rows [c, item_i] (or [c]): one wave per workgroup, LANE = HIDDEN UNIT (H <= 64), rows one after the other.
// Lane j keeps row j of W0 / W1 (and, backward, column j of W1 and W2) in registers; the activation vectors travel
// through 64-float LDS vectors (broadcast reads).  (A first version with a thread per row and the activations in
// per-thread arrays spilled them to scratch: 64 + 110 us per step instead of ~10.)
// ---------------------------------------------------------------------------
// input of row r: [c, item_feat[i][0..D)) with c = r / I (conditional) or [r]   (r wave-uniform)
__device__ inline void ct_row_input(const CtLayout& L, const float* __restrict__ item_feat, int r, float (&x)[kCtMaxDim + 1]) {
#pragma unroll
    for (int d = 0; d < kCtMaxDim + 1; ++d) x[d] = 0.f;
    if (L.cond) {
        const int c = r / L.I, i = r - c * L.I;
        x[0] = (float)c;
#pragma unroll
        for (int d = 0; d < kCtMaxDim; ++d)
            if (d < L.D) x[1 + d] = item_feat[(size_t)i * L.D + d];
    } else {
        x[0] = (float)r;
    }
}
// h_out[lane] = elu(bias + sum_k wrow[k] * vec[k]) with vec in LDS (broadcast reads)
template <int H>
__device__ __forceinline__ float ct_dot_lds(const float (&wrow)[H], const float* __restrict__ vec) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < H; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(vec + k);
        a0 = fmaf(wrow[k], v.x, a0); a1 = fmaf(wrow[k + 1], v.y, a1); a2 = fmaf(wrow[k + 2], v.z, a2); a3 = fmaf(wrow[k + 3], v.w, a3);
    }
    return (a0 + a1) + (a2 + a3);
}

template <int H>
__global__ __launch_bounds__(64) void ct_table_kernel(const CtLayout L, const float* __restrict__ P, const float* __restrict__ item_feat,
                                                      float* __restrict__ table, int rpw) {
    __shared__ __attribute__((aligned(16))) float S0[64], S1[64];
    const int lane = threadIdx.x;
    const bool act = lane < H;
    const int j = act ? lane : 0;
    float w0r[kCtMaxDim + 1], w1r[H], w2r[H];
#pragma unroll
    for (int d = 0; d < kCtMaxDim + 1; ++d) w0r[d] = (act && d < L.xin) ? P[L.w0 + j * L.xin + d] : 0.f;
#pragma unroll
    for (int k = 0; k < H; ++k) w1r[k] = act ? P[L.w1 + j * H + k] : 0.f;
    const bool outl = lane < L.O;                    // lane q < O owns output q (row q of W2)
#pragma unroll
    for (int k = 0; k < H; ++k) w2r[k] = outl ? P[L.w2 + lane * H + k] : 0.f;
    const float b0j = act ? P[L.b0 + j] : 0.f, b1j = act ? P[L.b1 + j] : 0.f, b2q = outl ? P[L.b2 + lane] : 0.f;
    const int r0 = blockIdx.x * rpw;
    for (int rr = 0; rr < rpw; ++rr) {
        const int r = r0 + rr;
        if (r >= L.rows) break;
        float x[kCtMaxDim + 1];
        ct_row_input(L, item_feat, r, x);
        float a = b0j;
#pragma unroll
        for (int d = 0; d < kCtMaxDim + 1; ++d) a = fmaf(w0r[d], x[d], a);
        S0[lane] = act ? ct_elu(a) : 0.f;
        __syncthreads();
        const float h1 = ct_elu(b1j + ct_dot_lds<H>(w1r, S0));
        S1[lane] = act ? h1 : 0.f;
        __syncthreads();
        const float o = b2q + ct_dot_lds<H>(w2r, S1);
        if (outl) table[(size_t)r * L.O + lane] = o;
    }
}

// backward of the table rows: d loss / d table = -dLL + coef dREG (flat: [8 scalars | grad_table set 0 | set 1 | ...]);
// one partial record of MLP-parameter gradients per workgroup (fixed order), d loss / d item_feat of the row -> gx
template <int H>
__global__ __launch_bounds__(64) void ct_rows_backward_kernel(const CtLayout L, const float* __restrict__ P, const float* __restrict__ item_feat,
                                                              const float* __restrict__ flat, const float* __restrict__ beta_p,
                                                              float* __restrict__ scratch, int rpw) {
    __shared__ __attribute__((aligned(16))) float S0[64], S1[64], Sg[2 * VIBO_MAX_ABILITY_DIM];
    const int lane = threadIdx.x;
    const bool act = lane < H;
    const int j = act ? lane : 0;
    const float coef = L.F > 0 ? 1.0f : *beta_p;            // (flows: the annealing factor is ignored, models.py:406-424)
    const size_t n_table = (size_t)L.rows * L.O;
    float w0r[kCtMaxDim + 1], w1r[H], w1c[H], w2c[2 * VIBO_MAX_ABILITY_DIM];
#pragma unroll
    for (int d = 0; d < kCtMaxDim + 1; ++d) w0r[d] = (act && d < L.xin) ? P[L.w0 + j * L.xin + d] : 0.f;
#pragma unroll
    for (int k = 0; k < H; ++k) {
        w1r[k] = act ? P[L.w1 + j * H + k] : 0.f;          // row j:    W1[j][k]
        w1c[k] = act ? P[L.w1 + k * H + j] : 0.f;          // column j: W1[k][j]
    }
#pragma unroll
    for (int q = 0; q < 2 * VIBO_MAX_ABILITY_DIM; ++q) w2c[q] = (act && q < L.O) ? P[L.w2 + q * H + j] : 0.f;
    const float b0j = act ? P[L.b0 + j] : 0.f, b1j = act ? P[L.b1 + j] : 0.f;
    // accumulators of this lane's rows of the parameter gradients
    float aW0[kCtMaxDim + 1], aW1[H], aW2[2 * VIBO_MAX_ABILITY_DIM];
    float ab0 = 0.f, ab1 = 0.f, ab2 = 0.f;
#pragma unroll
    for (int d = 0; d < kCtMaxDim + 1; ++d) aW0[d] = 0.f;
#pragma unroll
    for (int k = 0; k < H; ++k) aW1[k] = 0.f;
#pragma unroll
    for (int q = 0; q < 2 * VIBO_MAX_ABILITY_DIM; ++q) aW2[q] = 0.f;
    const int r0 = blockIdx.x * rpw;
    for (int rr = 0; rr < rpw; ++rr) {
        const int r = r0 + rr;
        if (r >= L.rows) break;
        float x[kCtMaxDim + 1];
        ct_row_input(L, item_feat, r, x);
        float a = b0j;
#pragma unroll
        for (int d = 0; d < kCtMaxDim + 1; ++d) a = fmaf(w0r[d], x[d], a);
        const float h0 = act ? ct_elu(a) : 0.f;
        S0[lane] = h0;
        if (lane < 2 * VIBO_MAX_ABILITY_DIM)
            Sg[lane] = lane < L.O ? -flat[VIBO_NUM_SCALARS + (size_t)r * L.O + lane] + coef * flat[VIBO_NUM_SCALARS + n_table + (size_t)r * L.O + lane] : 0.f;
        __syncthreads();
        const float h1 = act ? ct_elu(b1j + ct_dot_lds<H>(w1r, S0)) : 0.f;
        // g1 = W2^T gout * elu'(z1)   (elu'(z) = 1 | e^z = h + 1); W2 / b2 gradients
        float g1 = 0.f;
#pragma unroll
        for (int q = 0; q < 2 * VIBO_MAX_ABILITY_DIM; ++q) {
            const float go = Sg[q];
            g1 = fmaf(w2c[q], go, g1);
            aW2[q] = fmaf(go, h1, aW2[q]);
        }
        if (lane < L.O) ab2 += Sg[lane];
        g1 *= h1 > 0.f ? 1.0f : h1 + 1.0f;
        if (!act) g1 = 0.f;
        ab1 += g1;
        // W1 gradient: row j accumulates g1[j] * h0[k]
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(S0 + k);
            aW1[k] = fmaf(g1, v.x, aW1[k]); aW1[k + 1] = fmaf(g1, v.y, aW1[k + 1]);
            aW1[k + 2] = fmaf(g1, v.z, aW1[k + 2]); aW1[k + 3] = fmaf(g1, v.w, aW1[k + 3]);
        }
        S1[lane] = g1;
        __syncthreads();
        // g0[k = lane] = sum_j W1[j][k] g1[j] * elu'(z0)
        float g0 = ct_dot_lds<H>(w1c, S1) * (h0 > 0.f ? 1.0f : h0 + 1.0f);
        if (!act) g0 = 0.f;
        ab0 += g0;
#pragma unroll
        for (int d = 0; d < kCtMaxDim + 1; ++d) aW0[d] = fmaf(g0, x[d], aW0[d]);
        // d loss / d x[1..] of this row (the conditional encoder sees the item sample): sums over the hidden units
        if (L.cond) {
#pragma unroll
            for (int d = 0; d < kCtMaxDim; ++d)
                if (d < L.D) {
                    const float t = wave_total(w0r[1 + d] * g0);
                    if (lane == 0) scratch[L.s_gx + (size_t)r * kCtMaxDim + d] = t;
                }
        }
        __syncthreads();            // S0 / S1 / Sg are rewritten by the next row
    }
    float* rec = scratch + L.s_mrec + (size_t)blockIdx.x * L.n_mlp;
    if (act) {
        for (int d = 0; d < L.xin; ++d) rec[L.w0 + j * L.xin + d] = aW0[d];
        rec[L.b0 + j] = ab0;
#pragma unroll
        for (int k = 0; k < H; ++k) rec[L.w1 + j * H + k] = aW1[k];
        rec[L.b1 + j] = ab1;
        for (int q = 0; q < L.O; ++q) rec[L.w2 + q * H + j] = aW2[q];
    }
    if (lane < L.O) rec[L.b2 + lane] = ab2;
}

// ---------------------------------------------------------------------------
// item side backward: d loss / d item_k -> planar flows (saved tanh) -> + encoder-input gradient -> sample -> Adam
// ---------------------------------------------------------------------------
struct CtItemParams {
    CtLayout L;
    const float* flat; const float* eps; const float* item_k;
    const float* beta_p; const float* lr_p; const int32_t* step_count;
    float* mu; float* lv; float* im; float* iv;
    float* scratch;
};
__global__ __launch_bounds__(kCtItems) void ct_item_backward_kernel(const CtItemParams q) {
    const CtLayout& L = q.L;
    __shared__ float pk[VIBO_MAX_FLOWS][2 * kCtMaxDim + 2];
    __shared__ float red[4][2 * kCtMaxDim + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int D = L.D, F = L.F;
    for (int e = tid; e < F * (2 * kCtMaxDim + 2); e += kCtItems) (&pk[0][0])[e] = q.scratch[L.s_pack + e];
    __syncthreads();
    const int i = blockIdx.x * kCtItems + tid;
    const bool ok = i < L.I;
    const float beta = *q.beta_p, lr = *q.lr_p;
    const float t_ = (float)q.step_count[0];
    const float bc1 = 1.0f - powf(0.9f, t_), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t_));
    const size_t o_item = VIBO_NUM_SCALARS + 2 * (size_t)L.rows * L.O;
    float zz[kCtMaxDim], gz[kCtMaxDim];
#pragma unroll
    for (int d = 0; d < kCtMaxDim; ++d) {
        zz[d] = (ok && d < D) ? q.item_k[(size_t)i * D + d] : 0.f;
        // d loss / d item_k = -dLL/d item_k  (+ item_k: -log p(item_k) with flows)
        gz[d] = (ok && d < D) ? -q.flat[o_item + (size_t)i * D + d] + (F > 0 ? zz[d] : 0.f) : 0.f;
    }
    const float gl = ok ? -1.0f : 0.f;                  // d loss / d ladj_i: loss holds + log q = ... - ladj
    for (int k = F - 1; k >= 0; --k) {
        const float t = ok ? q.scratch[L.s_tanh + (size_t)i * F + k] : 0.f;
        const float c = pk[k][2 * D + 1];
        const float omt = 1.0f - t * t;
        const float psi = 1.0f + omt * c;
        const float dl_dpsi = gl * ((psi >= 0.f) ? 1.0f : -1.0f) / (fabsf(psi) + 1e-8f);
        float g_t = dl_dpsi * (-2.0f * t * c);
#pragma unroll
        for (int d = 0; d < kCtMaxDim; ++d)
            if (d < D) {
                zz[d] = fmaf(-pk[k][d], t, zz[d]);              // the flow's input (to an ulp of the forward's value)
                g_t = fmaf(gz[d], pk[k][d], g_t);
            }
        const float g_a = g_t * omt;
        const float g_c = dl_dpsi * omt;
        float pg[2 * kCtMaxDim + 1];
#pragma unroll
        for (int d = 0; d < kCtMaxDim; ++d) {
            pg[d] = d < D ? fmaf(g_c, pk[k][D + d], gz[d] * t) : 0.f;                  // d/d uhat
            pg[kCtMaxDim + d] = d < D ? fmaf(g_c, pk[k][d], g_a * zz[d]) : 0.f;        // d/d w (at fixed uhat)
        }
        pg[2 * kCtMaxDim] = g_a;                                                       // d/d b
#pragma unroll
        for (int d = 0; d < kCtMaxDim; ++d)
            if (d < D) gz[d] = fmaf(g_a, pk[k][D + d], gz[d]);
#pragma unroll
        for (int e = 0; e < 2 * kCtMaxDim + 1; ++e) {
            const int dd = e < kCtMaxDim ? e : e - kCtMaxDim;
            if (e < 2 * kCtMaxDim && dd >= D) continue;
            const float tot = wave_total(pg[e]);
            if (lane == 0) red[wv][e] = tot;
        }
        __syncthreads();
        if (tid < 2 * D + 1) {
            const int e = tid < D ? tid : tid < 2 * D ? kCtMaxDim + (tid - D) : 2 * kCtMaxDim;
            q.scratch[L.s_frec + ((size_t)blockIdx.x * F + k) * (2 * D + 1) + tid] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        }
        __syncthreads();
    }
    if (!ok) return;
    const bool kl_mode = F == 0;
#pragma unroll
    for (int d = 0; d < kCtMaxDim; ++d)
        if (d < D) {
            const int idx = i * D + d;
            float gf = gz[d];
            if (L.cond) gf += q.scratch[L.s_gx + (size_t)i * kCtMaxDim + d] + q.scratch[L.s_gx + (size_t)(L.I + i) * kCtMaxDim + d];
            const float m = q.mu[idx], l = q.lv[idx];
            // KL mode: + beta KL(q(d) || N(0,1));  flows: + log q(d_0) = ... - lv / 2 (mu cancels through the sample)
            const float g_mu = kl_mode ? gf + beta * m : gf;
            const float g_lv = gf * 0.5f * expf(0.5f * l) * q.eps[idx] + (kl_mode ? -0.5f * beta * (1.0f - expf(l)) : -0.5f);
            float pm = m, pl = l;
            const int n = L.I * D;
            ct_adam(pm, q.im[idx], q.iv[idx], g_mu, lr, bc1, bc2_sqrt);
            ct_adam(pl, q.im[n + idx], q.iv[n + idx], g_lv, lr, bc1, bc2_sqrt);
            q.mu[idx] = pm;
            q.lv[idx] = pl;
        }
}

// ---------------------------------------------------------------------------
// finish: MLP partial records -> gradients -> Adam (64 parameters per workgroup); last workgroup: loss, flow parameters
// ---------------------------------------------------------------------------
struct CtFinParams {
    CtLayout L;
    const float* flat; const float* beta_p; const float* lr_p; int32_t* step_count;
    float* P; float* M; float* V; float* scratch; float* loss_out;
};
__global__ __launch_bounds__(256) void ct_finish_kernel(const CtFinParams q) {
    const CtLayout& L = q.L;
    __shared__ float part[4][64];
    __shared__ float fg[VIBO_MAX_FLOWS][2 * kCtMaxDim + 1];
    const int tid = threadIdx.x;
    const float lr = *q.lr_p;
    const float t_ = (float)q.step_count[0];
    const float bc1 = 1.0f - powf(0.9f, t_), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t_));
    const int n_pb = (L.n_mlp + 63) / 64;
    if ((int)blockIdx.x < n_pb) {
        const int e = tid & 63, sl = tid >> 6;
        const int k = blockIdx.x * 64 + e;
        // (fixed order; four records in flight per thread: the loop is latency-bound)
        float acc = 0.f;
        if (k < L.n_mlp) {
            const float* rp = q.scratch + L.s_mrec + k;
            int b = sl;
            for (; b + 12 < L.n_rb; b += 16) {
                const float v0 = rp[(size_t)b * L.n_mlp], v1 = rp[(size_t)(b + 4) * L.n_mlp];
                const float v2 = rp[(size_t)(b + 8) * L.n_mlp], v3 = rp[(size_t)(b + 12) * L.n_mlp];
                acc += (v0 + v1) + (v2 + v3);
            }
            for (; b < L.n_rb; b += 4) acc += rp[(size_t)b * L.n_mlp];
        }
        part[sl][e] = acc;
        __syncthreads();
        if (sl == 0 && k < L.n_mlp) {
            const float g = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
            float p = q.P[k], m = q.M[k], v = q.V[k];
            ct_adam(p, m, v, g, lr, bc1, bc2_sqrt);
            q.P[k] = p; q.M[k] = m; q.V[k] = v;
        }
        return;
    }
    // ---- last workgroup
    const int A = L.A, D = L.D, F = L.F;
    if (tid < 64) {                          // item-side scalars: the prologue's partial sums, fixed order
        float kl = 0.f, lq = 0.f, lp = 0.f;
        for (int b = tid; b < L.n_ib; b += 64) {
            kl += q.scratch[L.s_parts + (size_t)b * 4 + 0];
            lq += q.scratch[L.s_parts + (size_t)b * 4 + 1];
            lp += q.scratch[L.s_parts + (size_t)b * 4 + 2];
        }
        kl = wave_total(kl); lq = wave_total(lq); lp = wave_total(lp);
        if (tid == 0) {
            const float beta = *q.beta_p;
            // models.py:427-430 | 406-424
            *q.loss_out = F == 0 ? -q.flat[VIBO_S_LL] + beta * (q.flat[VIBO_S_REG] + kl)
                                 : -(q.flat[VIBO_S_LL] + lp - q.flat[VIBO_S_REG] - lq);
            q.step_count[1] += 1;            // completed steps: the noise counter of the NEXT step
        }
    }
    if (F == 0) return;
    // ability flows: d loss / d (uhat, w, b) = -dLL + dREG from the ELBO kernel
    const size_t o_flow = VIBO_NUM_SCALARS + 2 * (size_t)L.rows * L.O + (size_t)L.I * D;
    const int nfa = F * (2 * A + 1);
    if (tid < F) {
        float g[2 * VIBO_MAX_ABILITY_DIM + 1], o[2 * VIBO_MAX_ABILITY_DIM + 1];
        for (int e = 0; e < 2 * A + 1; ++e) g[e] = -q.flat[o_flow + tid * (2 * A + 1) + e] + q.flat[o_flow + nfa + tid * (2 * A + 1) + e];
        ct_unpack_flow_grad(q.P + L.fa + tid * (2 * A + 1), A, g, o);
        for (int e = 0; e < 2 * A + 1; ++e) fg[tid][e] = o[e];
    }
    __syncthreads();
    for (int e = tid; e < nfa; e += 256) {
        const int k = L.fa + e;
        float p = q.P[k], m = q.M[k], v = q.V[k];
        ct_adam(p, m, v, fg[e / (2 * A + 1)][e % (2 * A + 1)], lr, bc1, bc2_sqrt);
        q.P[k] = p; q.M[k] = m; q.V[k] = v;
    }
    __syncthreads();
    // item flows: the item blocks' partial records (d/d uhat | d/d w | d/d b), fixed order
    const int nfi = F * (2 * D + 1);
    for (int e = tid; e < nfi; e += 256) {
        float acc = 0.f;
        for (int b = 0; b < L.n_ib; ++b) acc += q.scratch[L.s_frec + (size_t)b * nfi + e];
        fg[e / (2 * D + 1)][e % (2 * D + 1)] = acc;
    }
    __syncthreads();
    if (tid < F) {
        float g[2 * kCtMaxDim + 1], o[2 * kCtMaxDim + 1];
        for (int e = 0; e < 2 * D + 1; ++e) g[e] = fg[tid][e];
        ct_unpack_flow_grad(q.P + L.fi + tid * (2 * D + 1), D, g, o);
        for (int e = 0; e < 2 * D + 1; ++e) fg[tid][e] = o[e];
    }
    __syncthreads();
    for (int e = tid; e < nfi; e += 256) {
        const int k = L.fi + e;
        float p = q.P[k], m = q.M[k], v = q.V[k];
        ct_adam(p, m, v, fg[e / (2 * D + 1)][e % (2 * D + 1)], lr, bc1, bc2_sqrt);
        q.P[k] = p; q.M[k] = m; q.V[k] = v;
    }
}

}  // namespace vibo

using namespace vibo;

static int ct_check(const vibo_desc* d, int hidden_dim, CtLayout* L) {
    if (!d || d->abi_version != VIBO_ABI_VERSION) return -2;
    if (d->posterior == VIBO_POSTERIOR_GIVEN) return -6;
    if (hidden_dim != 64 && hidden_dim != 32) return -6;              // (activations of a table row live in registers)
    if (d->num_item < 1 || d->ability_dim < 1 || d->ability_dim > VIBO_MAX_ABILITY_DIM || d->n_flows < 0 || d->n_flows > VIBO_MAX_FLOWS) return -3;
    *L = ct_layout(d->num_item, d->ability_dim, d->irt_model, d->posterior == VIBO_POSTERIOR_CONDITIONAL ? 1 : 0, d->n_flows, hidden_dim);
    return 0;
}

extern "C" {

int64_t vibo_ctrain_param_floats(const vibo_desc* d, int hidden_dim) {
    CtLayout L;
    return ct_check(d, hidden_dim, &L) ? 0 : (int64_t)L.n_par;
}
int64_t vibo_ctrain_scratch_floats(const vibo_desc* d, int hidden_dim) {
    CtLayout L;
    return ct_check(d, hidden_dim, &L) ? 0 : (int64_t)L.s_total;
}

int vibo_ctrain_prologue(const vibo_desc* d, int hidden_dim, const float* params, const float* item_mu, const float* item_logvar,
                         float* eps_item, uint64_t seed, int draw_noise, float* eps_ability, uint32_t ability_stream_id,
                         float* item_feat, float* item_k, float* table, float* flow_packed, float* scratch, int32_t* step_count,
                         void* stream) {
    CtLayout L;
    const int rc = ct_check(d, hidden_dim, &L);
    if (rc) return rc;
    if (!params || !item_mu || !item_logvar || !eps_item || !item_feat || !item_k || !table || !scratch || !step_count) return -5;
    if (L.F > 0 && !flow_packed) return -5;
    if (draw_noise && !eps_ability) return -5;
    CtProParams q;
    q.L = L; q.params = params; q.mu = item_mu; q.lv = item_logvar; q.eps = eps_item; q.item_feat = item_feat; q.item_k = item_k;
    q.flow_packed = flow_packed; q.scratch = scratch; q.step_count = step_count;
    q.gen = draw_noise ? 1 : 0; q.seed_lo = (uint32_t)seed; q.seed_hi = (uint32_t)(seed >> 32);
    q.eps_ab = eps_ability; q.n_ab = draw_noise ? (long long)d->num_person * d->ability_dim : 0; q.ab_stream = ability_stream_id;
    const long long ab_blocks = draw_noise ? ((q.n_ab + 3) / 4 + kCtItems - 1) / kCtItems : 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ct_prologue_kernel, dim3((unsigned)(1 + L.n_ib + ab_blocks)), dim3(kCtItems), 0, s, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rpw = ct_rows_per_wave(L.rows);
    if (hidden_dim == 64) hipLaunchKernelGGL(ct_table_kernel<64>, dim3(L.n_rb), dim3(64), 0, s, L, params, (const float*)item_feat, table, rpw);
    else hipLaunchKernelGGL(ct_table_kernel<32>, dim3(L.n_rb), dim3(64), 0, s, L, params, (const float*)item_feat, table, rpw);
    return (int)hipGetLastError();
}

int vibo_ctrain_epilogue(const vibo_desc* d, int hidden_dim, const float* flat, const float* eps_item, const float* item_feat,
                         const float* item_k, const float* beta, const float* lr, int32_t* step_count, float* params, float* adam_m,
                         float* adam_v, float* item_mu, float* item_logvar, float* item_m, float* item_v, float* scratch,
                         float* loss_out, void* stream) {
    CtLayout L;
    const int rc = ct_check(d, hidden_dim, &L);
    if (rc) return rc;
    if (!flat || !eps_item || !item_feat || !item_k || !beta || !lr || !step_count || !params || !adam_m || !adam_v || !item_mu ||
        !item_logvar || !item_m || !item_v || !scratch || !loss_out)
        return -5;
    hipStream_t s = (hipStream_t)stream;
    const int rpw = ct_rows_per_wave(L.rows);
    if (hidden_dim == 64) hipLaunchKernelGGL(ct_rows_backward_kernel<64>, dim3(L.n_rb), dim3(64), 0, s, L, (const float*)params, item_feat, flat, beta, scratch, rpw);
    else hipLaunchKernelGGL(ct_rows_backward_kernel<32>, dim3(L.n_rb), dim3(64), 0, s, L, (const float*)params, item_feat, flat, beta, scratch, rpw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    CtItemParams qi;
    qi.L = L; qi.flat = flat; qi.eps = eps_item; qi.item_k = item_k; qi.beta_p = beta; qi.lr_p = lr; qi.step_count = step_count;
    qi.mu = item_mu; qi.lv = item_logvar; qi.im = item_m; qi.iv = item_v; qi.scratch = scratch;
    hipLaunchKernelGGL(ct_item_backward_kernel, dim3(L.n_ib), dim3(kCtItems), 0, s, qi);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    CtFinParams qf;
    qf.L = L; qf.flat = flat; qf.beta_p = beta; qf.lr_p = lr; qf.step_count = step_count; qf.P = params; qf.M = adam_m; qf.V = adam_v;
    qf.scratch = scratch; qf.loss_out = loss_out;
    hipLaunchKernelGGL(ct_finish_kernel, dim3((L.n_mlp + 63) / 64 + 1), dim3(256), 0, s, qf);
    return (int)hipGetLastError();
}

}  // extern "C"
