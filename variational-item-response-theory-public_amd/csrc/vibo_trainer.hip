// vibo_trainer.hip -- the O(I) part of a VIBO train step as two kernels (see include/vibo_hip.h):
// prologue (item sample, item KL, 2-row encoder MLP forward) and epilogue (loss, MLP backward, item
// backward, Adam).  One workgroup (block 0) owns the MLP; the other workgroups own 256 item entries each.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_finalize.hpp"
#include "vibo_philox.hpp"
#include "vibo_train_hook.hpp"

namespace vibo {

__global__ __launch_bounds__(256) void train_prologue_kernel(int H, int O, int I, int D, const float* __restrict__ P,
                                                             const float* __restrict__ mu, const float* __restrict__ lv,
                                                             const float* __restrict__ eps, float* __restrict__ item_feat,
                                                             float* __restrict__ table, float* __restrict__ saved_h,
                                                             float* __restrict__ kl_parts, int32_t* step_count,
                                                             // tick: step_count[0] += 1 (the four-launch step); tick == 0 primes the
                                                             // folded step (whose ELBO launch ticks) and selects the half of the
                                                             // double-buffered kl_parts the coming step will read
                                                             int tick,
                                                             // noise (gen != 0): eps is written here instead of read, and the
                                                             // blocks past the item blocks fill eps_ab [n_ab]; both draws are
                                                             // the streams vibo_fill_normal gives for step_count[1]
                                                             int gen, uint32_t seed_lo, uint32_t seed_hi, float* __restrict__ eps_w,
                                                             float* __restrict__ eps_ab, long long n_ab, uint32_t ab_stream,
                                                             int n_item_blocks) {
    __shared__ float h1[2 * kMaxHidden], h2[2 * kMaxHidden];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        if (tick && tid == 0) *step_count += 1;
        const MlpOffsets o = mlp_offsets(H, O);
        mlp2_layer0(P, o, H, O, h1, tid, 256);
        __syncthreads();
        mlp2_layer1(P, o, H, O, h1, h2, tid, 256);
        __syncthreads();
        mlp2_layer2(P, o, H, O, h1, h2, tid, 256, table, saved_h);
        return;
    }
    if ((int)blockIdx.x > n_item_blocks) {          // ability noise (stream ab_stream), 4 normals per thread
        const long long g = (long long)(blockIdx.x - 1 - n_item_blocks) * 256 + tid;
        if (4 * g < n_ab) store_normal4(eps_ab, n_ab, g, philox_normal4(g, (uint32_t)step_count[1], ab_stream, seed_lo, seed_hi));
        return;
    }
    // item side: 256 entries per workgroup, dimension-major (item_entry_index); one KL part per wave
    const int n_item_entries = I * D;
    const int k = (blockIdx.x - 1) * 256 + tid;
    float kl = 0.f;
    if (k < n_item_entries) {
        const int idx = item_entry_index(k, I, D);
        const float m = mu[idx], l = lv[idx];
        float e;
        if (gen) {                                   // entry idx of stream 0 (its group of 4 is recomputed by 4 threads: O(I) work)
            e = philox_normal1(idx, (uint32_t)step_count[1], 0u, seed_lo, seed_hi);
            eps_w[idx] = e;
        } else {
            e = eps[idx];
        }
        item_feat[idx] = item_sample(m, l, e);
        kl = item_kl_term(m, l);
    }
    kl = wave_total(kl);
    // (tick == 0: the part buffer of the step that is about to run, step_count[0] + 1)
    float* parts = kl_parts + (tick ? 0 : (((*step_count + 1) & 1) ? kl_part_count(n_item_entries) : 0));
    if ((tid & 63) == 0 && 256 * ((int)blockIdx.x - 1) + (tid & ~63) < n_item_entries) parts[4 * (blockIdx.x - 1) + (tid >> 6)] = kl;
}

// torch.optim.Adam's update (betas 0.9 / 0.999, eps 1e-8).  Every product-sum is pinned to one fma: the two epilogue
// kernels below must agree bit for bit, and the contraction hipcc picks for a sum of two products depends on the
// surrounding code.
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, const float g, const float lr, const float bc1,
                                            const float bc2_sqrt) {
    m = fmaf(0.9f, m, 0.1f * g);                   // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(0.999f, v, (0.001f * g) * g);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / bc2_sqrt + 1e-8f;
    p -= (lr / bc1) * (m / denom);
}

// item entry idx: d loss / d item_feat = gf -> (item_mu, item_logvar) through the sample and the item KL, Adam in place
__device__ __forceinline__ void epi_item_update(const int idx, const int n_item_entries, const float gf, const float e, const float beta,
                                                const float lr, const float bc1, const float bc2_sqrt, float* mu, float* lv, float* im,
                                                float* iv, float& pm, float& pl) {
    const float m = mu[idx], l = lv[idx];
    const float g_mu = fmaf(beta, m, gf);
    const float half_sd = 0.5f * expf(0.5f * l);
    const float klg = (0.5f * beta) * (1.0f - expf(l));
    const float g_lv = fmaf(gf * half_sd, e, -klg);
    pm = m; pl = l;
    adam_update(pm, im[idx], iv[idx], g_mu, lr, bc1, bc2_sqrt);
    adam_update(pl, im[n_item_entries + idx], iv[n_item_entries + idx], g_lv, lr, bc1, bc2_sqrt);
    mu[idx] = pm;
    lv[idx] = pl;
}

constexpr int kEpiThreads = 1024;      // block 0's chain of small dependent stages is latency-bound: more lanes per stage, fewer passes
struct EpiLds {
    float h1[2][kMaxHidden], h2[2][kMaxHidden], gh2[2][kMaxHidden], gh1[2][kMaxHidden], gout[2][2 * VIBO_MAX_ABILITY_DIM_WIDE];
};

// Block 0 of the epilogue: loss, the 2-row MLP backward by hand, Adam on the MLP parameters.
//   sc: the 8 ELBO scalars (VIBO_S_*); gtab: d LL / d table [2][O] then d REG / d table [2][O]
//   W / ow: where the backward reads the weights and their layout -- the caller's flat buffer P (ld = H), or the fused
//   epilogue's padded LDS copy, which then also receives the updated values (Wout) for the next step's forward
//   pv / mv / vv: parameter, first and second moment of elements tid + 1024 u, loaded by the caller (as early as it can)
//   HC: the hidden width as a compile-time constant (64: the reference default), or 0 for a runtime width -- the Adam loop
//   decodes six flat indices per thread with / H and % H, ~40 instructions each when H is not a constant (5 us of the chain)
constexpr int kEpiU = 8;
template <int HC>
__device__ __forceinline__ void epi_mlp_block(EpiLds& L, const int H_, const int O, const int n_kl_parts, const float* sc, const float* gtab,
                                              const float* __restrict__ saved_h, const float* __restrict__ kl_parts, const float beta,
                                              const float lr, const float bc1, const float bc2_sqrt, const float* W, const MlpOffsets ow,
                                              float* Wout, float* P, float* M, float* V, float (&pv)[kEpiU], float (&mv)[kEpiU],
                                              float (&vv)[kEpiU], float* loss_out, const int tid) {
    constexpr int BS = kEpiThreads;
    const int H = HC > 0 ? HC : H_;
    const int n_table = 2 * O;
    const MlpOffsets o = mlp_offsets(H, O);
    for (int k = tid; k < 2 * H; k += BS) {
        L.h1[k / H][k % H] = saved_h[k];
        L.h2[k / H][k % H] = saved_h[2 * H + k];
    }
    // d loss / d table = -dLL + beta dREG
    for (int k = tid; k < n_table; k += BS) L.gout[k / O][k % O] = fmaf(beta, gtab[n_table + k], -gtab[k]);
    if (tid < 64) {                  // item KL: the prologue's partial sums, fixed order (lane-strided, then the wave sum)
        float kl = 0.f;
        for (int k = tid; k < n_kl_parts; k += 64) kl += kl_parts[k];
        kl = wave_total(kl);
        if (tid == 0) *loss_out = fmaf(beta, sc[VIBO_S_REG] + kl, -sc[VIBO_S_LL]);
    }
    __syncthreads();
    // g_h2 = W2^T g_out * elu'(pre2),  elu'(x) = x > 0 ? 1 : elu(x) + 1
    for (int k = tid; k < 2 * H; k += BS) {
        const int r = k / H, j = k % H;
        float a = 0.f;
#pragma unroll 16
        for (int q = 0; q < O; ++q) a = fmaf(W[ow.w2 + q * ow.ld + j], L.gout[r][q], a);      // 16 loads in flight
        const float h = L.h2[r][j];
        L.gh2[r][j] = a * (h > 0.f ? 1.0f : h + 1.0f);
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
                for (int q = part * len; q < q1; ++q) a = fmaf(W[ow.w1 + q * ow.ld + j], L.gh2[r][q], a);
            }
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            a += __shfl_xor(a, 4);
            if (out < 2 * H && part == 0) {
                const float h = L.h1[r][j];
                L.gh1[r][j] = a * (h > 0.f ? 1.0f : h + 1.0f);
            }
        }
    }
    __syncthreads();      // all reads of the OLD weights are done: parameters may now be updated in place
    // Adam over the MLP parameters: 8 independent elements per thread and pass (the first pass's loads were issued by the caller)
    constexpr int U = kEpiU;
    for (int k0 = tid; k0 < o.total; k0 += BS * U) {
        if (k0 != tid) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + BS * u;
                const bool ok = k < o.total;
                pv[u] = ok ? P[k] : 0.f;
                mv[u] = ok ? M[k] : 0.f;
                vv[u] = ok ? V[k] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + BS * u;
            if (k >= o.total) continue;
            float g;
            if (k < o.b0) {                         // W0[j]: input of row r is r
                g = L.gh1[1][k - o.w0];
            } else if (k < o.w1) {
                const int j = k - o.b0;
                g = L.gh1[0][j] + L.gh1[1][j];
            } else if (k < o.b1) {
                const int j = (k - o.w1) / H, q = (k - o.w1) % H;
                g = fmaf(L.gh2[0][j], L.h1[0][q], L.gh2[1][j] * L.h1[1][q]);
            } else if (k < o.w2) {
                const int j = k - o.b1;
                g = L.gh2[0][j] + L.gh2[1][j];
            } else if (k < o.b2) {
                const int q = (k - o.w2) / H, j = (k - o.w2) % H;
                g = fmaf(L.gout[0][q], L.h2[0][j], L.gout[1][q] * L.h2[1][j]);
            } else {
                const int q = k - o.b2;
                g = L.gout[0][q] + L.gout[1][q];
            }
            adam_update(pv[u], mv[u], vv[u], g, lr, bc1, bc2_sqrt);
            P[k] = pv[u];
            M[k] = mv[u];
            V[k] = vv[u];
            if (Wout) Wout[mlp_reindex(k, H, O, ow)] = pv[u];      // (ow.ld = H + 1)
        }
    }
}
// the first pass's parameter / moment loads of epi_mlp_block
__device__ __forceinline__ void epi_mlp_prefetch(const int total, const float* P, const float* M, const float* V, float (&pv)[kEpiU],
                                                 float (&mv)[kEpiU], float (&vv)[kEpiU], const int tid) {
#pragma unroll
    for (int u = 0; u < kEpiU; ++u) {
        const int k = tid + kEpiThreads * u;
        const bool ok = k < total;
        pv[u] = ok ? P[k] : 0.f;
        mv[u] = ok ? M[k] : 0.f;
        vv[u] = ok ? V[k] : 0.f;
    }
}

__global__ __launch_bounds__(kEpiThreads) void train_epilogue_kernel(int H, int O, int n_item_entries, int n_kl_parts,
                                                             const float* __restrict__ flat, const float* __restrict__ saved_h,
                                                             const float* __restrict__ kl_parts, const float* __restrict__ eps,
                                                             const float* __restrict__ beta_p, const float* __restrict__ lr_p,
                                                             const int32_t* __restrict__ step_count, float* P, float* M, float* V,
                                                             float* mu, float* lv, float* im, float* iv, float* loss_out) {
    __shared__ EpiLds L;
    const int tid = threadIdx.x;
    const float beta = *beta_p, lr = *lr_p;
    const float t = (float)(*step_count);
    const float bc1 = 1.0f - powf(0.9f, t), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t));
    const int n_table = 2 * O;
    constexpr int BS = kEpiThreads;
    if (blockIdx.x == 0) {
        if (tid == 0) const_cast<int32_t*>(step_count)[1] += 1;      // completed steps: the noise counter of the NEXT step
        // flat: [8 scalars | grad_table set 0 | set 1 | grad_item]
        float pv[kEpiU], mv[kEpiU], vv[kEpiU];
        const MlpOffsets o = mlp_offsets(H, O);
        epi_mlp_prefetch(o.total, P, M, V, pv, mv, vv, tid);
        if (H == 64) epi_mlp_block<64>(L, H, O, n_kl_parts, flat, flat + VIBO_NUM_SCALARS, saved_h, kl_parts, beta, lr, bc1, bc2_sqrt, P, o, nullptr,
                                       P, M, V, pv, mv, vv, loss_out, tid);
        else epi_mlp_block<0>(L, H, O, n_kl_parts, flat, flat + VIBO_NUM_SCALARS, saved_h, kl_parts, beta, lr, bc1, bc2_sqrt, P, o, nullptr, P, M, V,
                              pv, mv, vv, loss_out, tid);
        return;
    }
    const int idx = (blockIdx.x - 1) * BS + tid;
    if (idx < n_item_entries) {        // d loss / d item_feat = -dLL/ditem
        float pm, pl;
        epi_item_update(idx, n_item_entries, -flat[VIBO_NUM_SCALARS + 2 * n_table + idx], eps[idx], beta, lr, bc1, bc2_sqrt, mu, lv, im, iv, pm, pl);
    }
}

// The epilogue of the folded train step.  One launch does what finalize_kernel + train_epilogue_kernel do for THIS step and
// what vibo_train_prologue_noise does for the NEXT one (the step is software-pipelined across its own iterations: everything
// the ELBO kernel needs is already in memory when it starts):
//   * e.partial != null ("fused finalize", one GPU): the fixed-order fp64 sums over the ELBO kernel's per-workgroup partial
//     records happen here -- block 0 sums the 8 scalars and the table gradient, item block b the 64 item-gradient entries
//     it then applies -- in finalize_kernel<64>'s order (16 slices, record_slice_sum), so the step agrees bit for bit with
//     the four-launch form; the sums are also written to flat_out (what vibo_elbo_fwd_bwd would have returned).
//     e.partial == null (person-sharded: finalize_kernel ran before the all-reduce): the sums are read from flat_in.
//   * the next step's head: the thread that has just updated item entry idx redraws its noise (stream 0, counter
//     step_count[0] = what the next step's prologue would read from step_count[1]) and forms the next item sample and KL
//     term (kl_parts is double-buffered by step parity: this step's half is still being read by block 0); block 0, after
//     Adam, runs the 2-row MLP forward on the NEW parameters (table, saved_h); the blocks past the item blocks fill eps_ab.
constexpr int kEpiOut = 64, kEpiSlices = kEpiThreads / kEpiOut;
constexpr int kEpiStageHidden = 64;
constexpr int kEpiStageFloats = 3 * kEpiStageHidden + (kEpiStageHidden + 2 * VIBO_MAX_ABILITY_DIM_WIDE) * (kEpiStageHidden + 1) + 2 * VIBO_MAX_ABILITY_DIM_WIDE;
static_assert(kEpiStageFloats * 8 >= 0 && kEpiThreads * kEpiU >= kEpiStageFloats, "one prefetch pass covers the staged parameters");
static_assert(kEpiOut == kKlGroup, "one KL part per item block");

template <int HC>
__global__ __launch_bounds__(kEpiThreads) void train_epilogue_fused_kernel(const EpiParams e) {
    __shared__ EpiLds L;
    __shared__ double part[kEpiSlices][kEpiOut], part2[kEpiSlices][kEpiOut];
    __shared__ float sc[VIBO_NUM_SCALARS];
    __shared__ float gt[8 * VIBO_MAX_ABILITY_DIM_WIDE];
    __shared__ float Pl[kEpiStageFloats];        // block 0: padded copy of the MLP parameters (encoders up to kEpiStageHidden wide)
    const int tid = threadIdx.x;
    const int lane = tid % kEpiOut, slice = tid / kEpiOut;
    const int n_table = 2 * e.O;                 // floats per table-gradient set
    const int step = e.step_count[0];            // Adam's t of this step (ticked by the ELBO launch; nothing in this launch writes it)
    if ((int)blockIdx.x > e.n_item_blocks) {     // the next step's ability noise, 4 normals per thread
        const long long g = (long long)(blockIdx.x - 1 - e.n_item_blocks) * kEpiThreads + tid;
        if (4 * g < e.n_ab) store_normal4(e.eps_ab, e.n_ab, g, philox_normal4(g, (uint32_t)step, e.ab_stream, e.seed_lo, e.seed_hi));
        return;
    }
    const float beta = *e.beta, lr = *e.lr;
    const float t = (float)step;
    const float bc1 = 1.0f - powf(0.9f, t), bc2_sqrt = sqrtf(1.0f - powf(0.999f, t));
    const int n_parts = kl_part_count(e.n_item_entries);
    const float* kl_now = e.kl_parts + ((step & 1) ? n_parts : 0);
    float* kl_next = e.kl_parts + ((step & 1) ? 0 : n_parts);
    if (blockIdx.x == 0) {
        if (tid == 0) e.step_count[1] += 1;      // completed steps (nothing in this launch reads it)
        // Block 0 is one chain of small dependent stages: every global load it needs is issued up front (parameters and
        // Adam moments into registers, a padded copy of the parameters into LDS for the stages in between), so that the
        // chain pays memory latency once instead of per stage (25 -> ~12 us at 1 000 items).
        const int H = HC > 0 ? HC : e.H;
        const MlpOffsets og = mlp_offsets(H, e.O);
        const bool staged = H <= kEpiStageHidden;                   // (wider encoders read the weights from global memory)
        const MlpOffsets ow = staged ? mlp_offsets(H, e.O, H + 1) : og;
        float pv[kEpiU], mv[kEpiU], vv[kEpiU];
        epi_mlp_prefetch(og.total, e.P, e.M, e.V, pv, mv, vv, tid);
        const float* scp = e.flat_in;
        const float* gtp = e.flat_in + VIBO_NUM_SCALARS;
        if (e.partial) {
            // outputs [0, 8 + 2 n_table) of the logical vector: two sets of 64 outputs x 16 slices, their loads in flight together
            double* scd = &part2[0][0];          // (re-used below as 8 doubles, after the slice sums were consumed)
            double keep = 0.0;
            double ps[2];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int o = kEpiOut * pass + lane;
                const bool live = o < VIBO_NUM_SCALARS + 2 * n_table;
                ps[pass] = live ? record_slice_sum<kEpiSlices>(e.partial, (size_t)e.lay.stride, o, 0, e.nblk, slice) : 0.0;
            }
            part[slice][lane] = ps[0];
            part2[slice][lane] = ps[1];
            if (staged) {                        // (the parameter loads have landed by now)
#pragma unroll
                for (int u = 0; u < kEpiU; ++u) {
                    const int k = tid + kEpiThreads * u;
                    if (k < og.total) Pl[mlp_reindex(k, H, e.O, ow)] = pv[u];
                }
            }
            __syncthreads();
            if (slice < 2) {                     // (waves 0 and 1: one set each)
                const int o = kEpiOut * slice + lane;
                if (o < VIBO_NUM_SCALARS + 2 * n_table) {
                    double tsum = 0.0;
#pragma unroll
                    for (int s = 0; s < kEpiSlices; ++s) tsum += slice == 0 ? part[s][lane] : part2[s][lane];
                    if (o >= VIBO_NUM_SCALARS) {
                        gt[o - VIBO_NUM_SCALARS] = (float)tsum;
                        e.flat_out[o] = (float)tsum;
                    } else {
                        keep = tsum;
                    }
                }
            }
            __syncthreads();
            if (slice == 0 && lane < VIBO_NUM_SCALARS) scd[lane] = keep;
            __syncthreads();
            if (tid == 0) {
                // partial scalars: 0 ll, 1 kl, 2 logq0, 3 logp, 4 ladj, 5 nobs  (reg_mode KL: REG = KL)
                sc[VIBO_S_LL] = (float)scd[0]; sc[VIBO_S_REG] = (float)scd[1]; sc[VIBO_S_KL] = (float)scd[1];
                sc[VIBO_S_LOGQ0] = (float)scd[2]; sc[VIBO_S_LOGP] = (float)scd[3]; sc[VIBO_S_LADJ] = (float)scd[4];
                sc[VIBO_S_NOBS] = (float)scd[5]; sc[VIBO_S_RESERVED] = 0.f;
#pragma unroll
                for (int k = 0; k < VIBO_NUM_SCALARS; ++k) e.flat_out[k] = sc[k];
            }
            __syncthreads();
            scp = sc;
            gtp = gt;
        } else if (staged) {
#pragma unroll
            for (int u = 0; u < kEpiU; ++u) {
                const int k = tid + kEpiThreads * u;
                if (k < og.total) Pl[mlp_reindex(k, H, e.O, ow)] = pv[u];
            }
            __syncthreads();
        }
        const float* W = staged ? Pl : e.P;
        epi_mlp_block<HC>(L, H, e.O, n_parts, scp, gtp, e.saved_h, kl_now, beta, lr, bc1, bc2_sqrt, W, ow, staged ? Pl : nullptr, e.P, e.M, e.V,
                      pv, mv, vv, e.loss_out, tid);
        // the next step's expert table from the parameters just written (this workgroup's own stores: visible after the barrier)
        __syncthreads();
        float* h1 = &L.h1[0][0];
        float* h2 = &L.h2[0][0];
        mlp2_layer0(W, ow, H, e.O, h1, tid, kEpiThreads);
        __syncthreads();
        mlp2_layer1(W, ow, H, e.O, h1, h2, tid, kEpiThreads);
        __syncthreads();
        mlp2_layer2(W, ow, H, e.O, h1, h2, tid, kEpiThreads, e.table, e.saved_h);
        return;
    }
    // item block: entries k = 64 (block - 1) + lane in the records' order (dim-major: consecutive lanes = consecutive items)
    const int k = kEpiOut * ((int)blockIdx.x - 1) + lane;
    const bool live = k < e.n_item_entries;
    const int dd = live ? k / e.I : 0, i = live ? k % e.I : 0;
    const int idx = i * e.D + dd;
    float g = 0.f;
    if (e.partial) {
        part[slice][lane] = live ? record_slice_sum<kEpiSlices>(e.partial, (size_t)e.lay.stride, e.lay.off_item + dd * e.lay.i_pad + i, 0,
                                                                e.nblk, slice) : 0.0;
        __syncthreads();
        if (slice == 0 && live) {
            double tsum = 0.0;
#pragma unroll
            for (int s = 0; s < kEpiSlices; ++s) tsum += part[s][lane];
            g = (float)tsum;
            e.flat_out[VIBO_NUM_SCALARS + 2 * n_table + idx] = g;
        }
    } else if (slice == 0 && live) {
        g = e.flat_in[VIBO_NUM_SCALARS + 2 * n_table + idx];
    }
    if (slice == 0) {                            // (wave 0 of the block, all 64 lanes: the wave total below needs them)
        float kl = 0.f;
        if (live) {
            float pm, pl;
            epi_item_update(idx, e.n_item_entries, -g, e.eps_item[idx], beta, lr, bc1, bc2_sqrt, e.mu, e.lv, e.im, e.iv, pm, pl);
            const float en = philox_normal1(idx, (uint32_t)step, 0u, e.seed_lo, e.seed_hi);
            e.eps_item[idx] = en;
            e.item_feat[idx] = item_sample(pm, pl, en);
            kl = item_kl_term(pm, pl);
        }
        kl = wave_total(kl);
        if (lane == 0) kl_next[blockIdx.x - 1] = kl;
    }
}

__global__ __launch_bounds__(256) void fill_normal_kernel(float* __restrict__ out, long long n, uint32_t seed_lo, uint32_t seed_hi,
                                                          const int32_t* __restrict__ step_count, uint32_t stream_id) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;        // group of 4 outputs
    if (4 * g >= n) return;
    store_normal4(out, n, g, philox_normal4(g, (uint32_t)(*step_count), stream_id, seed_lo, seed_hi));
}

hipError_t launch_train_epilogue_fused(const EpiParams& e, hipStream_t s) {
    const long long ab_blocks = ((e.n_ab + 3) / 4 + kEpiThreads - 1) / kEpiThreads;
    if (e.H == 64) hipLaunchKernelGGL(train_epilogue_fused_kernel<64>, dim3((unsigned)(1 + e.n_item_blocks + ab_blocks)), dim3(kEpiThreads), 0, s, e);
    else hipLaunchKernelGGL(train_epilogue_fused_kernel<0>, dim3((unsigned)(1 + e.n_item_blocks + ab_blocks)), dim3(kEpiThreads), 0, s, e);
    return hipGetLastError();
}
int train_epilogue_item_blocks(int n_item_entries) { return (n_item_entries + kEpiOut - 1) / kEpiOut; }

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
    hipLaunchKernelGGL(train_prologue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hidden_dim, 2 * d->ability_dim, d->num_item,
                       item_dim_of(d), mlp_params, item_mu, item_logvar, eps_item, item_feat, table, saved_h, kl_parts, step_count, 1, 0, 0u, 0u,
                       (float*)nullptr, (float*)nullptr, 0LL, 0u, (n + 255) / 256);
    return (int)hipGetLastError();
}

extern "C" int vibo_train_prime(const vibo_desc* d, int hidden_dim, const float* mlp_params, const float* item_mu,
                                const float* item_logvar, const float* eps_item, float* item_feat, float* table,
                                float* saved_h, float* kl_parts, int32_t* step_count, void* stream) {
    if (!d || hidden_dim < 1 || hidden_dim > kMaxHidden || d->posterior != VIBO_POSTERIOR_UNCONDITIONAL || d->n_flows != 0) return -6;
    if (!mlp_params || !item_mu || !item_logvar || !eps_item || !item_feat || !table || !saved_h || !kl_parts || !step_count) return -5;
    const int n = d->num_item * item_dim_of(d);
    const int blocks = 1 + (n + 255) / 256;
    hipLaunchKernelGGL(train_prologue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hidden_dim, 2 * d->ability_dim, d->num_item,
                       item_dim_of(d), mlp_params, item_mu, item_logvar, eps_item, item_feat, table, saved_h, kl_parts, step_count, 0, 0, 0u, 0u,
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
                       hidden_dim, 2 * d->ability_dim, d->num_item, item_dim_of(d), mlp_params, item_mu, item_logvar, (const float*)eps_item,
                       item_feat, table, saved_h, kl_parts, step_count, 1, 1, (uint32_t)seed, (uint32_t)(seed >> 32), eps_item, eps_ability,
                       n_ab, ability_stream_id, item_blocks);
    return (int)hipGetLastError();
}

extern "C" int vibo_train_epilogue(const vibo_desc* d, int hidden_dim, const float* flat, const float* saved_h,
                                   const float* kl_parts, const float* eps_item, const float* beta, const float* lr,
                                   const int32_t* step_count, float* mlp_params, float* mlp_m, float* mlp_v,
                                   float* item_mu, float* item_logvar, float* item_m, float* item_v, float* loss_out,
                                   void* stream) {
    if (!d || hidden_dim < 1 || hidden_dim > kMaxHidden || d->posterior != VIBO_POSTERIOR_UNCONDITIONAL || d->n_flows != 0) return -6;
    const int n = d->num_item * item_dim_of(d);
    const int parts = kl_part_count(n);                      // the prologue's item-KL partial sums (one per 64 entries)
    hipLaunchKernelGGL(train_epilogue_kernel, dim3(1 + (n + kEpiThreads - 1) / kEpiThreads), dim3(kEpiThreads), 0, (hipStream_t)stream,
                       hidden_dim, 2 * d->ability_dim, n, parts, flat, saved_h, kl_parts, eps_item, beta, lr, step_count, mlp_params, mlp_m, mlp_v, item_mu,
                       item_logvar, item_m, item_v, loss_out);
    return (int)hipGetLastError();
}
