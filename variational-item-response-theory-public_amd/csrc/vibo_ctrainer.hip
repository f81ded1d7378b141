// vibo_ctrainer.hip -- the O(I) part of a VIBO train step for --conditional-posterior and / or --n-norm-flows
// (product-of-experts encoder, IRT decoder) as native kernels: what vibo_trainer.hip is for the plain model.
//
// Reference step (vibo.py:243-268) around the fused ELBO kernel:
//   item sample (models.py:361-362, 506-510) -> item-side planar flows (flows.py:21-66, models.py:346-348)
//   -> expert table = encoder MLP on the rows [c, item_i] (models.py:666-710; 2 rows [c] without the conditional posterior)
//   -> vibo_elbo_fwd_bwd -> loss (models.py:380-443) -> backward through table MLP / flows / sample -> Adam (vibo.py:221).
// The module path runs this as ~190 PyTorch launches per step (2.2 ms at 16 persons); here it is three launches around
// the ELBO call (prologue + table tiles | backward | update), all deterministic (fixed-order partial records), so a captured hipGraph contains no PyTorch autograd node.
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

constexpr int kCtTile = 16;            // table rows per tile of the matrix-pipe MLP kernels
// tiles per workgroup: one, until there are more than 512 tiles (then the partial records stay at <= 512)
__host__ __device__ inline int ct_tiles_per_block(int rows) {
    const int nt = (rows + kCtTile - 1) / kCtTile;
    const int t = (nt + 511) / 512;
    return t < 1 ? 1 : t;
}

struct CtLayout {
    int H, O, xin, A, D, F, I, cond, rows;
    int w0, b0, w1, b1, w2, b2, n_mlp, fa, fi, n_par;      // offsets into the flat parameter buffer
    int n_rb, n_ib;                                        // row blocks, item blocks
    // scratch (floats)
    size_t s_pack, s_tanh, s_parts, s_gx, s_mrec, s_frec, s_gz, s_total;
};
__host__ __device__ inline CtLayout ct_layout(int I, int A, int irt, int cond, int F, int H) {
    CtLayout L;
    L.H = H; L.A = A; L.O = 2 * A; L.D = irt == 1 ? 1 : irt == 2 ? A + 1 : A + 2; L.F = F; L.I = I; L.cond = cond;
    L.xin = 1 + (cond ? L.D : 0);
    L.rows = 2 * (cond ? I : 1);
    L.w0 = 0; L.b0 = L.w0 + H * L.xin; L.w1 = L.b0 + H; L.b1 = L.w1 + H * H; L.w2 = L.b1 + H; L.b2 = L.w2 + L.O * H;
    L.n_mlp = L.b2 + L.O;
    L.fa = L.n_mlp; L.fi = L.fa + F * (2 * A + 1); L.n_par = L.fi + F * (2 * L.D + 1);
    L.n_rb = ((L.rows + kCtTile - 1) / kCtTile + ct_tiles_per_block(L.rows) - 1) / ct_tiles_per_block(L.rows);
    L.n_ib = (I + kCtItems - 1) / kCtItems;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    L.s_pack = take((size_t)VIBO_MAX_FLOWS * (2 * kCtMaxDim + 2));
    L.s_tanh = take((size_t)I * (F > 0 ? F : 1));
    L.s_parts = take((size_t)L.n_ib * 4);
    L.s_gx = take((size_t)L.rows * kCtMaxDim);
    L.s_mrec = take((size_t)L.n_rb * L.n_mlp);
    L.s_frec = take((size_t)L.n_ib * (F > 0 ? F : 1) * (2 * L.D + 1));
    L.s_gz = take((size_t)I * kCtMaxDim);      // flows: d loss / d item_feat behind the item-side flows (ct_backward_kernel -> ct_update_kernel)
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

// parameters of the prologue launch (ct_prologue_kernel, after the table tiles below)
struct CtProParams {
    CtLayout L;
    const float* params; const float* mu; const float* lv;
    float* eps;                  // item noise: read (gen == 0) or written (gen != 0)
    float* item_feat; float* item_k; float* flow_packed; float* scratch;
    int32_t* step_count;
    int gen; uint32_t seed_lo, seed_hi;
    float* eps_ab; long long n_ab; uint32_t ab_stream;
    float* table; int ab_blocks, tpb;
};

// ---------------------------------------------------------------------------
// table rows [c, item_i] (or [c]) on the matrix pipe: tiles of 16 rows, one workgroup of four waves per tile (a workgroup
// walks `tpb` tiles when there are more than 512 of them), every contraction of the 64-wide MLP as v_mfma_f32_16x16x4_f32
// (fp32 operands: no splitting, results of fp32 grade by construction).  Hidden widths below 64 run zero-padded.
//   operand layouts of one 16x16x4 step (lane = 16 kk + i16):  A[i16][kk], B[kk][i16], C: register r' = C[4 kk + r'][i16]
//   K = 64 contractions walk k = 16 kk + s over 16 steps (each lane reads 16 consecutive floats of its row / loads 16
//   consecutive weights once per workgroup), K = 16 (rows of the tile, or outputs) walk k = 4 kk + s over 4 steps.
// (Round 3's version -- one wave per 8 rows, lane = hidden unit, dot products on the VALU through LDS broadcasts -- took
//  13 us forward and 25 us backward for 2 000 rows; it is gone.)
// ---------------------------------------------------------------------------
typedef float ct_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ct_f32x4 ct_mfma(const float a, const float b, const ct_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
constexpr int kCtLd = 68;               // LDS row stride (floats) of the 16 x 64 tiles
struct CtTileLds {
    float X[kCtTile][16];               // inputs [c | item_feat | 0 ...], column 15 = 1 (bias column of the W0-gradient contraction)
    float H0[kCtTile][kCtLd];
    float H1[kCtTile][kCtLd];
};
// the weights a lane keeps for the forward of a tile
struct CtFwdRegs {
    float w0r[kCtMaxDim + 1];           // row j = tid & 63 of W0
    float b0j;
    float b1v[16];                      // W1[16 w + i16][16 kk + s]
    float b1u;                          // b1[16 w + i16]
};
__device__ __forceinline__ void ct_load_fwd(const CtLayout& L, const float* __restrict__ P, const int tid, CtFwdRegs& R) {
    const int lane = tid & 63, w = tid >> 6, i16 = lane & 15, kk = lane >> 4, j = tid & 63, unit = 16 * w + i16;
#pragma unroll
    for (int d = 0; d < kCtMaxDim + 1; ++d) R.w0r[d] = (j < L.H && d < L.xin) ? P[L.w0 + j * L.xin + d] : 0.f;
    R.b0j = j < L.H ? P[L.b0 + j] : 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) R.b1v[s] = (unit < L.H && 16 * kk + s < L.H) ? P[L.w1 + unit * L.H + 16 * kk + s] : 0.f;
    R.b1u = unit < L.H ? P[L.b1 + unit] : 0.f;
}
// K = 64 contraction: A = row i16 of a 16 x 64 LDS tile, B = the lane's 16 weights (two accumulator chains)
__device__ __forceinline__ ct_f32x4 ct_contract64(const float (*T)[kCtLd], const float (&bw)[16], const int i16, const int kk) {
    ct_f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const float4 v = *reinterpret_cast<const float4*>(&T[i16][16 * kk + 4 * s4]);
        a0 = ct_mfma(v.x, bw[4 * s4], a0);
        a1 = ct_mfma(v.y, bw[4 * s4 + 1], a1);
        a0 = ct_mfma(v.z, bw[4 * s4 + 2], a0);
        a1 = ct_mfma(v.w, bw[4 * s4 + 3], a1);
    }
    return a0 + a1;
}
// rows r0 .. r0 + 15: inputs and layer 0 on the VALU, layer 1 on the matrix pipe; leaves X, H0, H1 in LDS (the caller
// synchronises before reading H1) and the lane's piece of H1 in h1c (rows 4 kk + r' of column 16 w + i16)
// feat(i, d): entry d of item i's sample (an array read, or -- in the prologue launch -- the sample formed on the spot)
template <class Feat>
__device__ __forceinline__ void ct_tile_forward(const CtLayout& L, const Feat& feat, const int r0, CtTileLds& S,
                                                const CtFwdRegs& R, ct_f32x4& h1c, const int tid) {
    const int lane = tid & 63, w = tid >> 6, i16 = lane & 15, kk = lane >> 4, unit = 16 * w + i16;
    {
        const int r = tid >> 4, d = tid & 15, row = r0 + r;
        float x = 0.f;
        if (row < L.rows) {
            if (L.cond) {
                const int c = row / L.I, i = row - c * L.I;
                x = d == 0 ? (float)c : (d <= L.D ? feat(i, d - 1) : 0.f);
            } else {
                x = d == 0 ? (float)row : 0.f;
            }
        }
        S.X[r][d] = d == 15 ? 1.0f : x;
    }
    __syncthreads();
    {
        const int j = tid & 63;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = (tid >> 6) + 4 * m;
            float a = R.b0j;
#pragma unroll
            for (int d = 0; d < kCtMaxDim + 1; ++d) a = fmaf(R.w0r[d], S.X[r][d], a);
            S.H0[r][j] = j < L.H ? ct_elu(a) : 0.f;
        }
    }
    __syncthreads();
    const ct_f32x4 c = ct_contract64(S.H0, R.b1v, i16, kk);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const float h = unit < L.H ? ct_elu(c[rr] + R.b1u) : 0.f;
        h1c[rr] = h;
        S.H1[4 * kk + rr][unit] = h;
    }
}

template <class Feat>
__device__ __forceinline__ void ct_table_body(const CtLayout& L, const float* __restrict__ P, const Feat& feat, float* __restrict__ table,
                                              const int tpb, const int block, CtTileLds& S) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i16 = lane & 15, kk = lane >> 4;
    CtFwdRegs R;
    ct_load_fwd(L, P, tid, R);
    float w2v[16];                      // wave 0: W2[q = i16][16 kk + s]
#pragma unroll
    for (int s = 0; s < 16; ++s) w2v[s] = (w == 0 && i16 < L.O && 16 * kk + s < L.H) ? P[L.w2 + i16 * L.H + 16 * kk + s] : 0.f;
    const float b2q = i16 < L.O ? P[L.b2 + i16] : 0.f;
    for (int t = 0; t < tpb; ++t) {
        const int r0 = (block * tpb + t) * kCtTile;
        if (r0 >= L.rows) break;
        ct_f32x4 h1c;
        ct_tile_forward(L, feat, r0, S, R, h1c, tid);
        __syncthreads();
        if (w == 0) {
            const ct_f32x4 c = ct_contract64(S.H1, w2v, i16, kk);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = r0 + 4 * kk + rr;
                if (row < L.rows && i16 < L.O) table[(size_t)row * L.O + i16] = c[rr] + b2q;
            }
        }
        __syncthreads();                // (the next tile rewrites X / H0 / H1)
    }
}

// ---------------------------------------------------------------------------
// prologue launch: block 0 = flow packing (+ step counter); blocks 1..n_ib = items (sample, flows forward, partial sums);
// then ab_blocks of ability noise; then the tiles of the expert table (they form the item sample of their 16 rows themselves,
// with the item blocks' statements: no wait for them -- round 3 ran the table as a launch of its own)
// ---------------------------------------------------------------------------
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
    if ((int)blockIdx.x > L.n_ib + q.ab_blocks) {    // expert-table tiles
        __shared__ __attribute__((aligned(16))) CtTileLds S;
        const uint32_t ctr = (uint32_t)q.step_count[1];
        auto feat = [&](const int i, const int d) -> float {
            const int idx = i * D + d;
            const float e = q.gen ? philox_normal1(idx, ctr, 0u, q.seed_lo, q.seed_hi) : q.eps[idx];
            return fmaf(expf(0.5f * q.lv[idx]), e, q.mu[idx]);
        };
        ct_table_body(L, q.params, feat, q.table, q.tpb, (int)blockIdx.x - 1 - L.n_ib - q.ab_blocks, S);
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

// backward of the table rows: d loss / d table = -dLL + coef dREG (flat: [8 scalars | grad_table set 0 | set 1 | ...]);
// one partial record of MLP-parameter gradients per workgroup (fixed order), d loss / d item_feat of the row -> gx
struct CtBwdLds {
    CtTileLds F;
    float G1[kCtTile][kCtLd];           // d loss / d (layer-1 pre-activation)
    float G0[kCtTile][kCtLd];           // d loss / d (layer-0 pre-activation)
    float Sg[kCtTile][16];              // d loss / d table row (zero past O)
};
__device__ __forceinline__ void ct_rows_backward_body(const CtLayout& L, const float* __restrict__ P, const float* __restrict__ item_feat,
                                                      const float* __restrict__ flat, const float* __restrict__ beta_p,
                                                      float* __restrict__ scratch, const int tpb, const int block, CtBwdLds& S) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i16 = lane & 15, kk = lane >> 4, unit = 16 * w + i16;
    const float coef = L.F > 0 ? 1.0f : *beta_p;            // (flows: the annealing factor is ignored, models.py:406-424)
    const size_t n_table = (size_t)L.rows * L.O;
    CtFwdRegs R;
    ct_load_fwd(L, P, tid, R);
    auto feat = [&](const int i, const int d) -> float { return item_feat[(size_t)i * L.D + d]; };
    float w1t[16], w2t[4], w0t[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int j = 16 * kk + s;
        w1t[s] = (j < L.H && unit < L.H) ? P[L.w1 + j * L.H + unit] : 0.f;                                 // W1[j][unit]
        w0t[s] = (L.cond && w == 0 && j < L.H && i16 < L.D) ? P[L.w0 + j * L.xin + 1 + i16] : 0.f;        // W0[j][1 + d], d = i16
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) w2t[s] = (4 * kk + s < L.O && unit < L.H) ? P[L.w2 + (4 * kk + s) * L.H + unit] : 0.f;   // W2[q][unit]
    // gradient accumulators (C tiles, carried over the workgroup's tiles)
    const ct_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    ct_f32x4 aW1[4] = {zero4, zero4, zero4, zero4};          // rows 16 w + 4 kk + r' of gW1, columns 16 cb + i16
    ct_f32x4 aW2 = zero4;                                    // gW2[q = 4 kk + r'][unit]
    ct_f32x4 aW0 = zero4;                                    // gW0[16 w + 4 kk + r'][d = i16], d = 15: gb0
    float ab = 0.f;                                          // tid < 64: gb1[tid]; 64 <= tid < 80: gb2[tid - 64]
    for (int t = 0; t < tpb; ++t) {
        const int r0 = (block * tpb + t) * kCtTile;
        if (r0 >= L.rows) break;
        {
            const int r = tid >> 4, q = tid & 15, row = r0 + r;
            float g = 0.f;
            if (row < L.rows && q < L.O)
                g = -flat[VIBO_NUM_SCALARS + (size_t)row * L.O + q] + coef * flat[VIBO_NUM_SCALARS + n_table + (size_t)row * L.O + q];
            S.Sg[r][q] = g;
        }
        ct_f32x4 h1c;
        ct_tile_forward(L, feat, r0, S.F, R, h1c, tid);          // (its barriers publish Sg as well)
        {   // G1 = (gO W2) * elu'(z1)   (elu'(z) = 1 | e^z = h + 1)
            ct_f32x4 c = zero4;
            const float4 v = *reinterpret_cast<const float4*>(&S.Sg[i16][4 * kk]);
            c = ct_mfma(v.x, w2t[0], c); c = ct_mfma(v.y, w2t[1], c); c = ct_mfma(v.z, w2t[2], c); c = ct_mfma(v.w, w2t[3], c);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) S.G1[4 * kk + rr][unit] = c[rr] * (h1c[rr] > 0.f ? 1.0f : h1c[rr] + 1.0f);
        }
        __syncthreads();
        // gW2 += gO^T H1,  gW1 += G1^T H0   (K = the tile's rows)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = 4 * kk + s;
            aW2 = ct_mfma(S.Sg[r][i16], S.F.H1[r][unit], aW2);
            const float ga = S.G1[r][unit];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) aW1[cb] = ct_mfma(ga, S.F.H0[r][16 * cb + i16], aW1[cb]);
        }
        if (tid < 64) {
#pragma unroll
            for (int r = 0; r < kCtTile; ++r) ab += S.G1[r][tid];
        } else if (tid < 80) {
#pragma unroll
            for (int r = 0; r < kCtTile; ++r) ab += S.Sg[r][tid - 64];
        }
        {   // G0 = (G1 W1) * elu'(z0)
            const ct_f32x4 c = ct_contract64(S.G1, w1t, i16, kk);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float h0 = S.F.H0[4 * kk + rr][unit];
                S.G0[4 * kk + rr][unit] = c[rr] * (h0 > 0.f ? 1.0f : h0 + 1.0f);
            }
        }
        __syncthreads();
        // gW0 | gb0 += G0^T [X | 1]
#pragma unroll
        for (int s = 0; s < 4; ++s) aW0 = ct_mfma(S.G0[4 * kk + s][unit], S.F.X[4 * kk + s][i16], aW0);
        // d loss / d x[1..] of the rows (the conditional encoder sees the item sample): G0 W0[:, 1:]
        if (L.cond && w == 0) {
            const ct_f32x4 c = ct_contract64(S.G0, w0t, i16, kk);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = r0 + 4 * kk + rr;
                if (row < L.rows && i16 < L.D) scratch[L.s_gx + (size_t)row * kCtMaxDim + i16] = c[rr];
            }
        }
        __syncthreads();                // (the next tile rewrites the LDS tiles)
    }
    float* rec = scratch + L.s_mrec + (size_t)block * L.n_mlp;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int jr = 16 * w + 4 * kk + rr;            // hidden unit = row of gW1 / gW0
        if (jr < L.H) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                if (16 * cb + i16 < L.H) rec[L.w1 + jr * L.H + 16 * cb + i16] = aW1[cb][rr];
            if (i16 < L.xin) rec[L.w0 + jr * L.xin + i16] = aW0[rr];
            if (i16 == 15) rec[L.b0 + jr] = aW0[rr];
        }
        const int q = 4 * kk + rr;
        if (q < L.O && unit < L.H) rec[L.w2 + q * L.H + unit] = aW2[rr];
    }
    if (tid < 64) {
        if (tid < L.H) rec[L.b1 + tid] = ab;
    } else if (tid < 80) {
        if (tid - 64 < L.O) rec[L.b2 + tid - 64] = ab;
    }
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
// flows (F > 0): d loss / d item_k back through the item-side planar flows -> s_gz, and this block's flow-parameter records
__device__ __forceinline__ void ct_item_flows_backward_body(const CtItemParams& q, const int block) {
    const CtLayout& L = q.L;
    __shared__ float pk[VIBO_MAX_FLOWS][2 * kCtMaxDim + 2];
    __shared__ float red[4][2 * kCtMaxDim + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int D = L.D, F = L.F;
    const int i = block * kCtItems + tid;
    const bool ok = i < L.I;
    const size_t o_item = VIBO_NUM_SCALARS + 2 * (size_t)L.rows * L.O;
    float zz[kCtMaxDim], gz[kCtMaxDim], th[VIBO_MAX_FLOWS];
#pragma unroll
    for (int d = 0; d < kCtMaxDim; ++d) {
        const bool on = ok && d < D;
        zz[d] = on ? q.item_k[(size_t)i * D + d] : 0.f;
        // d loss / d item_k = -dLL/d item_k + item_k (-log p(item_k))
        gz[d] = on ? -q.flat[o_item + (size_t)i * D + d] + zz[d] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < VIBO_MAX_FLOWS; ++k) th[k] = (ok && k < F) ? q.scratch[L.s_tanh + (size_t)i * F + k] : 0.f;
    for (int e = tid; e < F * (2 * kCtMaxDim + 2); e += kCtItems) (&pk[0][0])[e] = q.scratch[L.s_pack + e];
    __syncthreads();
    const float gl = ok ? -1.0f : 0.f;                  // d loss / d ladj_i: loss holds + log q = ... - ladj
#pragma unroll
    for (int k = VIBO_MAX_FLOWS - 1; k >= 0; --k) {
        if (k >= F) continue;                           // (uniform)
        const float t = th[k];
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
            q.scratch[L.s_frec + ((size_t)block * F + k) * (2 * D + 1) + tid] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        }
        __syncthreads();
    }
    if (!ok) return;
#pragma unroll
    for (int d = 0; d < kCtMaxDim; ++d)
        if (d < D) q.scratch[L.s_gz + (size_t)i * kCtMaxDim + d] = gz[d];
}

// d loss / d item_feat (+ the encoder-input gradient of the conditional posterior) -> sample -> Adam
__device__ __forceinline__ void ct_item_update_body(const CtItemParams& q, const int block) {
    const CtLayout& L = q.L;
    const int tid = threadIdx.x;
    const int D = L.D, F = L.F;
    const int i = block * kCtItems + tid;
    if (i >= L.I) return;
    const int n = L.I * D;
    const size_t o_item = VIBO_NUM_SCALARS + 2 * (size_t)L.rows * L.O;
    const float beta = *q.beta_p, lr = *q.lr_p;
    const float t_ = (float)q.step_count[0];
    const float bc1 = 1.0f - powf(0.9f, t_), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t_));
    const bool kl_mode = F == 0;
#pragma unroll
    for (int d = 0; d < kCtMaxDim; ++d)
        if (d < D) {
            const size_t idx = (size_t)i * D + d;
            float gf = kl_mode ? -q.flat[o_item + idx] : q.scratch[L.s_gz + (size_t)i * kCtMaxDim + d];
            if (L.cond) gf += q.scratch[L.s_gx + (size_t)i * kCtMaxDim + d] + q.scratch[L.s_gx + (size_t)(L.I + i) * kCtMaxDim + d];
            const float m = q.mu[idx], l = q.lv[idx];
            // KL mode: + beta KL(q(d) || N(0,1));  flows: + log q(d_0) = ... - lv / 2 (mu cancels through the sample)
            const float g_mu = kl_mode ? gf + beta * m : gf;
            const float g_lv = gf * 0.5f * expf(0.5f * l) * q.eps[idx] + (kl_mode ? -0.5f * beta * (1.0f - expf(l)) : -0.5f);
            float nm = m, nl = l;
            float m0 = q.im[idx], v0 = q.iv[idx], m1 = q.im[n + idx], v1 = q.iv[n + idx];
            ct_adam(nm, m0, v0, g_mu, lr, bc1, bc2_sqrt);
            ct_adam(nl, m1, v1, g_lv, lr, bc1, bc2_sqrt);
            q.mu[idx] = nm;
            q.lv[idx] = nl;
            q.im[idx] = m0; q.iv[idx] = v0;
            q.im[n + idx] = m1; q.iv[n + idx] = v1;
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
// records -> gradient -> Adam of 64 MLP parameters (fixed order)
__device__ __forceinline__ void ct_mlp_adam_body(const CtFinParams& q, const int block) {
    const CtLayout& L = q.L;
    __shared__ float part[4][64];
    const int tid = threadIdx.x;
    const float lr = *q.lr_p;
    const float t_ = (float)q.step_count[0];
    const float bc1 = 1.0f - powf(0.9f, t_), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t_));
    const int e = tid & 63, sl = tid >> 6;
    const int k = block * 64 + e;
    // (fixed order; four records in flight per thread: the loop is latency-bound)
    float acc = 0.f;
    float p = 0.f, m = 0.f, v = 0.f;
    if (k < L.n_mlp) {
        if (sl == 0) { p = q.P[k]; m = q.M[k]; v = q.V[k]; }
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
        ct_adam(p, m, v, g, lr, bc1, bc2_sqrt);
        q.P[k] = p; q.M[k] = m; q.V[k] = v;
    }
}

// the update launch's last workgroup: loss, flow parameters
__device__ __forceinline__ void ct_finish_last_body(const CtFinParams& q) {
    const CtLayout& L = q.L;
    __shared__ float fg[VIBO_MAX_FLOWS][2 * kCtMaxDim + 1];
    const int tid = threadIdx.x;
    const float lr = *q.lr_p;
    const float t_ = (float)q.step_count[0];
    const float bc1 = 1.0f - powf(0.9f, t_), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t_));
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


// ---------------------------------------------------------------------------
// the epilogue's two launches (round 3: four):
//   ct_backward_kernel  workgroups [0, n_rb) = table-row tiles (MLP backward), then -- with flows -- n_ib item blocks that take
//                       d loss / d item_k back through the item-side flows (independent of the tiles)
//   ct_update_kernel    [0, n_ib) = item sample backward + Adam, then the MLP records -> Adam, then ONE workgroup for the loss
//                       and the flow parameters; everything they read comes from earlier launches
// ---------------------------------------------------------------------------
struct CtEpiParams {
    CtItemParams it;
    CtFinParams fin;
    const float* item_feat;
    int tpb;
};
__global__ __launch_bounds__(kCtItems) void ct_backward_kernel(const CtEpiParams q) {
    const CtLayout& L = q.fin.L;
    __shared__ __attribute__((aligned(16))) CtBwdLds S;
    const int b = blockIdx.x;
    if (b < L.n_rb) ct_rows_backward_body(L, q.fin.P, q.item_feat, q.fin.flat, q.fin.beta_p, q.fin.scratch, q.tpb, b, S);
    else ct_item_flows_backward_body(q.it, b - L.n_rb);
}
__global__ __launch_bounds__(kCtItems) void ct_update_kernel(const CtEpiParams q) {
    const CtLayout& L = q.fin.L;
    const int b = blockIdx.x, n_pb = (L.n_mlp + 63) / 64;
    if (b < L.n_ib) ct_item_update_body(q.it, b);
    else if (b < L.n_ib + n_pb) ct_mlp_adam_body(q.fin, b - L.n_ib);
    else ct_finish_last_body(q.fin);
}

}  // namespace vibo

using namespace vibo;

static int ct_check(const vibo_desc* d, int hidden_dim, CtLayout* L) {
    if (!d || d->abi_version != VIBO_ABI_VERSION) return -2;
    if (d->posterior == VIBO_POSTERIOR_GIVEN) return -6;
    if (hidden_dim < 1 || hidden_dim > 64) return -6;                 // (one 64-wide tile of the matrix-pipe MLP kernels; narrower: zero-padded)
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
    q.table = table; q.ab_blocks = (int)ab_blocks; q.tpb = ct_tiles_per_block(L.rows);
    hipLaunchKernelGGL(ct_prologue_kernel, dim3((unsigned)(1 + L.n_ib + ab_blocks + L.n_rb)), dim3(kCtItems), 0, s, q);
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
    CtItemParams qi;
    qi.L = L; qi.flat = flat; qi.eps = eps_item; qi.item_k = item_k; qi.beta_p = beta; qi.lr_p = lr; qi.step_count = step_count;
    qi.mu = item_mu; qi.lv = item_logvar; qi.im = item_m; qi.iv = item_v; qi.scratch = scratch;
    CtFinParams qf;
    qf.L = L; qf.flat = flat; qf.beta_p = beta; qf.lr_p = lr; qf.step_count = step_count; qf.P = params; qf.M = adam_m; qf.V = adam_v;
    qf.scratch = scratch; qf.loss_out = loss_out;
    CtEpiParams qe;
    qe.it = qi; qe.fin = qf; qe.item_feat = item_feat; qe.tpb = ct_tiles_per_block(L.rows);
    hipLaunchKernelGGL(ct_backward_kernel, dim3(L.n_rb + (L.F > 0 ? L.n_ib : 0)), dim3(kCtItems), 0, s, qe);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ct_update_kernel, dim3(L.n_ib + (L.n_mlp + 63) / 64 + 1), dim3(kCtItems), 0, s, qe);
    return (int)hipGetLastError();
}

}  // extern "C"
