// vibo_multi_kernel.hpp -- SC forward evaluations of the ELBO heads per pass over the response rows.
//
// log_marginal (models.py:445-504) draws S fresh (item, ability) samples and evaluates
//   log w_s = sum ll + log p(theta_K) + log p(d) - log q(theta) - log q(d)
// for each of them: S full forwards over the same responses.  Everything that depends on the responses only -- the
// row loads, the fp8 codes, the counts / product of experts -- is sample independent, so this kernel does it once and
// runs the per-sample part (reparameterised sample, flows, decode, log-lik) SC times per row while the row's codes
// sit in LDS.  Same mapping as the row-split kernel (vibo_split_kernel.hpp: an item never leaves its lane, a row is
// shared by nq <= 4 waves, batches of 8 rows loaded one batch ahead); forward only, so no reductions of gradients
// and one workgroup barrier per batch.  Works under the panel mode (row_cnt) and the conditional posterior
// (pre_stats) exactly like the row-split kernel.
//   item parameters: SC prepped tables, sample s at item_prep + s * item_sstride (floats)
//   noise:           eps[s][B][A], sample s at eps + s * eps_sstride
//   partial record:  scalars of sample s at out[8 s + {ll, kl, logq0, logp, ladj, nobs}]
#pragma once
#include <hip/hip_runtime.h>
#include "vibo_multi.hpp"
#include "vibo_split_kernel.hpp"

namespace vibo {

struct alignas(16) MultiWaveLds {
    uint32_t codes[kSplitRows][64];
    float thl[4][64];                 // theta[r][d] of the batch, per sample
    int cntp[2][8];                   // double-buffered: only one barrier per batch
    float red[4][8];
};
struct alignas(16) MultiCommonLds {
    float ctab[4 * 2 * 8];
    float fpar[kMF][2][8];
    float fsc[kMF][2];
};
inline size_t multi_lds_bytes(int nq) { return sizeof(MultiCommonLds) + (size_t)nq * sizeof(MultiWaveLds); }

template <int AT, int IRT, bool FLOWS, int SC>
__global__ __launch_bounds__(256, 2) void multi_forward_kernel(const MultiParams mp) {
    const ElboParams& p = mp.e;
    constexpr int R = kSplitRows;
    constexpr int NE = R * AT;
    constexpr int H = AT / 2;
    constexpr float kLoS = kLogitLo * kLog2e;
    extern __shared__ __attribute__((aligned(16))) unsigned char multi_smem[];
    MultiCommonLds& cl = *reinterpret_cast<MultiCommonLds*>(multi_smem);
    MultiWaveLds* wls = reinterpret_cast<MultiWaveLds*>(multi_smem + sizeof(MultiCommonLds));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = blockDim.x >> 6;
    MultiWaveLds& wl = wls[q];
    const int I = p.I, A = p.A;
    const int n4 = (I + 3) >> 2;
    const int chunk = q * 64 + lane;
    const bool chunk_ok = chunk < n4;
    const uint32_t tail_mask = ((I & 3) && chunk == (I >> 2)) ? ((1u << (8 * (I & 3))) - 1u) : 0xFFFFFFFFu;

    if (tid < 2 * AT) {
        const int c = tid / AT, a = tid % AT;
        float m = 0.f, s = 0.f;
        if (a < A) { m = p.table[c * 2 * A + a]; s = p.table[c * 2 * A + A + a]; }
        const float tau = 1.0f / (__expf(s) + kPoeEps);
        cl.ctab[(0 * 2 + c) * AT + a] = tau;
        cl.ctab[(1 * 2 + c) * AT + a] = m * tau;
    }
    if constexpr (FLOWS) {
        if (tid < kMF * 8) {
            const int f = tid >> 3, a = tid & 7;
            const bool ok = f < p.n_flows && a < A;
            cl.fpar[f][0][a] = ok ? p.flow[(size_t)f * (2 * A + 1) + a] : 0.f;
            cl.fpar[f][1][a] = ok ? p.flow[(size_t)f * (2 * A + 1) + A + a] : 0.f;
        }
        if (tid < kMF) {
            float cwu = 0.f, b = 0.f;
            if (tid < p.n_flows) {
                const float* fp = p.flow + (size_t)tid * (2 * A + 1);
                for (int a = 0; a < A; ++a) cwu = fmaf(fp[A + a], fp[a], cwu);
                b = fp[2 * A];
            }
            cl.fsc[tid][0] = b;
            cl.fsc[tid][1] = cwu;
        }
    }

    // ---- this lane's 4 items, one parameter set per sample ----
    float2v na2[SC][4][H];
    float nb[SC][4];
    float gs[SC][4], om[SC][4];
#pragma unroll
    for (int s = 0; s < SC; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* ir = p.item_prep + s * mp.item_sstride + (size_t)(p.item0 + 4 * chunk + j) * p.DP;
#pragma unroll
            for (int a = 0; a < AT; ++a) na2[s][j][a >> 1][a & 1] = chunk_ok ? ir[a] : 0.f;
            nb[s][j] = chunk_ok ? ir[AT] : 0.f;
            gs[s][j] = (IRT == 3 && chunk_ok) ? ir[AT + 1] : 0.f;
            om[s][j] = (IRT == 3 && chunk_ok) ? ir[AT + 2] : 1.f;
        }
    const int er = (lane / AT) & (R - 1), ed = lane % AT;
    const bool e_ok = lane < NE && ed < A;
    float s_log[SC], s_logq0[SC], s_logp[SC], s_ladj[SC];
#pragma unroll
    for (int s = 0; s < SC; ++s) s_log[s] = s_logq0[s] = s_logp[s] = s_ladj[s] = 0.f;
    float s_kl = 0.f, s_nobs = 0.f;
    int unobs = 0;
    __syncthreads();
    const float tau0 = cl.ctab[(0 * 2 + 0) * AT + ed], tau1 = cl.ctab[(0 * 2 + 1) * AT + ed];
    const float mt0 = cl.ctab[(1 * 2 + 0) * AT + ed], mt1 = cl.ctab[(1 * 2 + 1) * AT + ed];

    const long long n_batches = ((long long)p.B + R - 1) / R;
    const bool cell_codes = p.mask_dtype == 3;          // VIBO_MASK_CODES: 1-byte cell codes through p.mask
    float4 x[R];
    uint32_t m[R];
    float epn[SC];
    auto load_batch = [&](const long long bt) {
        const long long row0 = bt * R;
        long long srcs[R];               // (row indices first: see the row-split kernel)
#pragma unroll
        for (int r = 0; r < R; ++r) srcs[r] = row0 + r;
        if (p.row_index) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (row0 + r < p.B) srcs[r] = p.row_index[row0 + r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long row = row0 + r;
            x[r] = float4{0.f, 0.f, 0.f, 0.f};
            m[r] = cell_codes ? kAllMissing4 : 0u;
            if (row < p.B && chunk_ok) {
                const long long src = srcs[r];
                if (!cell_codes) x[r] = reinterpret_cast<const float4*>(p.response + src * p.resp_stride + p.item0)[chunk];
                if (p.mask_dtype == 0 || cell_codes)
                    m[r] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride + p.item0)[chunk];
                else
                    m[r] = 0x01010101u;
            }
        }
        const long long erow = row0 + er;
#pragma unroll
        for (int s = 0; s < SC; ++s) epn[s] = (e_ok && erow < p.B) ? p.eps[s * mp.eps_sstride + erow * A + ed] : 0.f;
    };

    long long bt = blockIdx.x;
    int par = 0;
    if (bt < n_batches) load_batch(bt);
    for (; bt < n_batches; bt += gridDim.x, par ^= 1) {
        const long long row0 = bt * R;
        int pk[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            pk[r] = 0;
            wl.codes[r][lane] = cell_codes ? pack_cell_codes4(m[r], tail_mask, pk[r]) : pack_codes4(x[r], m[r] & tail_mask, pk[r]);
        }
        float eps_c[SC];
#pragma unroll
        for (int s = 0; s < SC; ++s) eps_c[s] = epn[s];
        if (bt + gridDim.x < n_batches) load_batch(bt + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int tot = bfly8(pk, lane);
            if ((lane & 7) == 0) wl.cntp[par][lane >> 3] = tot;
            if constexpr (IRT != 3) {      // this lane's unobserved cells: each adds exactly log2(1 + 2^0) = 1 below
                int obs8 = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) obs8 += pk[r];
                unobs += 4 * R - (obs8 & 0xffff);
            }
        }
        __syncthreads();

        // ---- product of experts (sample independent), then SC reparameterised samples + flows ----
        const bool live = e_ok && (row0 + er) < p.B;
        int cnt = 0;
        if (p.row_cnt) {
            cnt = live ? p.row_cnt[row0 + er] : 0;
        } else {
            for (int w = 0; w < nq; ++w) cnt += wls[w].cntp[par][er];
        }
        const float n1 = (float)(cnt >> 16);
        float nobs = (float)(cnt & 0xffff);
        const float n0 = nobs - n1;
        float lam = n0 * tau0 + n1 * tau1, smu = n0 * mt0 + n1 * mt1;
        if (p.pre_stats) {
            lam = 0.f; smu = 0.f; nobs = 0.f;
            if (live) {
                for (int pn = 0; pn < p.pre_panels; ++pn) {
                    const float* st = p.pre_stats + ((size_t)pn * p.B + (row0 + er)) * (2 * A + 1);
                    lam += st[ed]; smu += st[A + ed]; nobs += st[2 * A];
                }
            }
        }
        const float nmiss = (float)p.I_total - nobs;
        if (p.missing_mode == 0) lam += nmiss * (1.0f / (1.0f + kPoeEps));
        if (!live) lam = 1.0f;
        const float inv_lam = 1.0f / lam;
        const float amu = smu * inv_lam;
        const float sig = fast_rsq(lam);
        const float alv = -kLn2 * fast_log2(lam);
        const bool head = q == 0 && live && p.primary;
        if (head) {
            s_kl += -0.5f * (1.0f + alv - amu * amu - inv_lam);
            if (ed == 0) s_nobs += nobs;
        }
#pragma unroll
        for (int s = 0; s < SC; ++s) {
            float thv = live ? amu + sig * eps_c[s] : 0.f;
            float ladj = 0.f;
            if constexpr (FLOWS) {
#pragma unroll
                for (int f = 0; f < kMF; ++f) {
                    if (f < p.n_flows) {
                        const float ud = cl.fpar[f][0][ed], wd = cl.fpar[f][1][ed];
                        const float aa = group_sum<AT>(thv * wd) + cl.fsc[f][0];
                        const float t = 1.0f - 2.0f * fast_rcp(1.0f + fast_exp2((2.0f * kLog2e) * med3(aa, -15.f, 15.f)));
                        const float psi = 1.0f + (1.0f - t * t) * cl.fsc[f][1];
                        ladj += kLn2 * fast_log2(fabsf(psi) + 1e-8f);
                        thv = fmaf(ud, t, thv);
                    }
                }
            }
            if (head) {
                s_logq0[s] += -0.5f * kLog2Pi - 0.5f * alv - 0.5f * eps_c[s] * eps_c[s];
                s_logp[s] += -0.5f * kLog2Pi - 0.5f * thv * thv;
                if (ed == 0) s_ladj[s] += ladj;
            }
            wl.thl[s][lane] = thv;
        }

        // ---- decode + masked Bernoulli log-lik: the row's codes are read once, then SC logit sets ----
#pragma unroll 2
        for (int r = 0; r < R; ++r) {
            const uint32_t cw = wl.codes[r][lane];
            const float2v w01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)cw, false);
            const float2v w23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)cw, true);
            const float w[4] = {w01[0], w01[1], w23[0], w23[1]};
#pragma unroll
            for (int s = 0; s < SC; ++s) {
                float2v th2[H];
                if constexpr (AT >= 4) {
#pragma unroll
                    for (int a = 0; a < AT; a += 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(&wl.thl[s][r * AT + a]);
                        th2[a / 2] = float2v{t4.x, t4.y};
                        th2[a / 2 + 1] = float2v{t4.z, t4.w};
                    }
                } else {
                    th2[0] = *reinterpret_cast<const float2v*>(&wl.thl[s][r * AT]);
                }
                float prod = 1.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float2v l2 = float2v{nb[s][t], 0.f};
#pragma unroll
                    for (int h = 0; h < H; ++h) l2 = na2[s][t][h] * th2[h] + l2;
                    const float l = l2[0] + l2[1];
                    if constexpr (IRT != 3) {
                        // value clamp of the reference's Bernoulli (utils.py:46-49 -> torch) at logit +-15.94
                        const float lc = med3(l, -kLoS, kLoS);
                        prod *= 1.0f + fast_exp2(-w[t] * lc);
                    } else {
                        const float e = fast_exp2(-fabsf(l));
                        const float rr_ = fast_rcp(1.0f + e);
                        const float er_ = e * rr_;
                        const float sp = (l >= 0.f) ? rr_ : er_, sn = (l >= 0.f) ? er_ : rr_;
                        const float pc = med3(fmaf(om[s][t], sp, gs[s][t]), kEps32, 1.0f - kEps32);
                        const float arg = (w[t] > 0.f) ? pc : med3(om[s][t] * sn, kEps32, 1.0f - kEps32);
                        prod *= (w[t] != 0.f) ? arg : 1.0f;
                    }
                }
                s_log[s] += fast_log2(prod);
            }
        }
    }

    // ================= workgroup reduction -> partial record: 8 scalars per sample ======
    float* out = p.partial + (size_t)blockIdx.x * p.lay.stride;
    const float t_kl = wave_total(s_kl), t_no = wave_total(s_nobs);
#pragma unroll
    for (int s = 0; s < SC; ++s) {
        const float ll = (IRT == 3) ? kLn2 * wave_total(s_log[s]) : -kLn2 * wave_total(s_log[s] - (float)unobs);
        const float t_q0 = wave_total(s_logq0[s]), t_lp = wave_total(s_logp[s]), t_la = wave_total(s_ladj[s]);
        if (lane == 0) {
            wl.red[s][0] = ll; wl.red[s][1] = t_kl; wl.red[s][2] = t_q0; wl.red[s][3] = t_lp; wl.red[s][4] = t_la;
            wl.red[s][5] = t_no; wl.red[s][6] = 0.f; wl.red[s][7] = 0.f;
        }
    }
    __syncthreads();
    if (tid < 8 * SC) {
        float t = 0.f;
        for (int w = 0; w < nq; ++w) t += wls[w].red[tid >> 3][tid & 7];
        out[tid] = t;
    }
}

template <int AT, int IRT, int SC>
static hipError_t launch_multi_flows(const MultiParams& mp, int nq, int grid, hipStream_t s) {
    const size_t lds = multi_lds_bytes(nq);
    if (mp.e.n_flows > 0) hipLaunchKernelGGL((multi_forward_kernel<AT, IRT, true, SC>), dim3(grid), dim3(64 * nq), lds, s, mp);
    else hipLaunchKernelGGL((multi_forward_kernel<AT, IRT, false, SC>), dim3(grid), dim3(64 * nq), lds, s, mp);
    return hipGetLastError();
}

// sc in {1, 2, 4} (4 only for template widths <= 4)
template <int AT>
static hipError_t launch_multi_at(const MultiParams& mp, int irt, int sc, int nq, int grid, hipStream_t s) {
#define VIBO_MULTI_IRT(SCV)                                                              \
    if (irt == 1) return launch_multi_flows<AT, 1, SCV>(mp, nq, grid, s);                  \
    if (irt == 2) return launch_multi_flows<AT, 2, SCV>(mp, nq, grid, s);                  \
    return launch_multi_flows<AT, 3, SCV>(mp, nq, grid, s);
    if (sc == 1) { VIBO_MULTI_IRT(1) }
    if (sc == 2) { VIBO_MULTI_IRT(2) }
    if constexpr (AT <= 4) {
        if (sc == 4) { VIBO_MULTI_IRT(4) }
    }
#undef VIBO_MULTI_IRT
    return hipErrorInvalidValue;
}

}  // namespace vibo
