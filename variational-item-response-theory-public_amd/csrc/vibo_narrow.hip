// vibo_narrow.hip -- fused ELBO forward+backward for NARROW response rows (4 <= I <= 128 items: BASELINE configs[0]'s 100 items,
// configs[3]'s CritLangAcq with 95), unconditional posterior, 1PL/2PL/3PL, ability_dim <= 4, fp32 / gathered / 1-byte rows.
//
// Why a kernel of its own.  The row-split kernels give a response row to a whole wave (items in lanes, 4 per lane): a 96-item
// row keeps 24 of the 64 lanes busy, and every 8-row batch pays a 64-lane butterfly for the counts, an LDS transpose for
// d LL/d theta and two workgroup barriers (profiles/r04_other_paths_kernel_trace.txt: 535 596 x 96 in 142.6 us = 1.85 TB/s, 0.21
// of the roofline, for a matrix that fits the Infinity Cache).  Here a row belongs to a 16-lane DPP row:
//   * lane (g = lane >> 4, j = lane & 15) owns items IL j .. IL j + IL - 1 (IL = 4 up to 64 items, 8 up to 128) of the FOUR rows
//     4 k + g a wave takes per pass -- a wave load instruction reads 4 adjacent rows (<= 2 KB contiguous), and 12 of 16 lanes
//     work on a 96-item row instead of 24 of 64;
//   * every cross-lane sum of the per-person math (answer counts, d LL/d theta) is four DPP steps inside the 16-lane row
//     (quad_perm, quad_perm, row_half_mirror, row_mirror) that leave the total in all 16 lanes: no LDS, no barrier, no
//     cross-row traffic anywhere in the loop; waves never wait for each other;
//   * lane j of a row also holds ability dim j & (AT - 1) of its person: the product of experts + reparameterised sample
//     (models.py:596-629) is computed per dim, theta comes back to every lane by quad broadcasts, the per-person backward
//     (sample -> PoE -> table gradient) runs in the same lanes and accumulates in registers (lanes j < AT only).
// Item parameters and item-gradient accumulators stay in registers for the whole kernel (IL x (A + 1 [+ 1]) each); the four
// row groups of a wave and the four waves of a workgroup are summed once at the end, in a fixed order, into the per-workgroup
// partial record of the other kernels (vibo_params.hpp): results are bitwise reproducible.
// The per-cell arithmetic is the VALU row-split kernel's (vibo_split_kernel.hpp): logits in log2 units, one exp2 + one rcp per
// cell, one log2 per 4 cells, the reference's Bernoulli probability clamp (utils.py:46-49 -> torch) as value clamp + rare
// wave-uniform gradient fix-up (1PL/2PL) or on p itself (3PL, models.py:758-765).
#include <hip/hip_runtime.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_params.hpp"
#include "vibo_launch.hpp"

namespace vibo {

// sum over the 16 lanes of a DPP row; every lane of the row gets the total (same tree in every lane: same bits)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xb1>(v);                           // quad_perm [1,0,3,2]
    v += dpp_f<0x4e>(v);                           // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);                          // row_half_mirror
    v += dpp_f<0x140>(v);                          // row_mirror
    return v;
}
__device__ __forceinline__ int row16_sum(int v) {
    v += dpp_i<0xb1>(v);
    v += dpp_i<0x4e>(v);
    v += dpp_i<0x141>(v);
    v += dpp_i<0x140>(v);
    return v;
}
// element `a` of the lane's quad (quad_perm [a, a, a, a])
template <int A_>
__device__ __forceinline__ float quad_bcast(float v) {
    return dpp_f<(A_ | (A_ << 2) | (A_ << 4) | (A_ << 6))>(v);
}

constexpr int kNwWaves = 4;            // waves per workgroup (they only meet in the epilogue)
#ifndef VIBO_NW_PASS
#define VIBO_NW_PASS 1
#endif
constexpr int kNwPass = VIBO_NW_PASS;  // passes (of 4 rows) per loop iteration = rows in flight per wave / 4

// (narrow_waves_per_simd: vibo_launch.hpp, shared with the planner)

struct alignas(16) NarrowLds {
    float red[kNwWaves][8];
    float tred[kNwWaves][8][64];
    float item[kNwWaves][6][128];      // [wave][gradient row <= AT + 2][item]
};

// AT: template ability width (1, 2, 4; runtime p.A <= AT).  IL: items per lane (4: I <= 64, 8: I <= 128).
// RM: 0 fp32 rows in order, 1 fp32 rows through p.row_index, 2 cell codes (through p.mask), with or without p.row_index.
template <int AT, int IRT, bool GRAD, int RM, int IL>
__global__ __launch_bounds__(64 * kNwWaves, narrow_waves_per_simd(AT, IL, IRT == 3 && GRAD)) void narrow_kernel(const ElboParams p) {
    constexpr bool CODES = RM == 2;
    constexpr int NC = IL / 4;                     // 16-byte chunks per lane and row
    constexpr float kLoS = kLogitLo * kLog2e, kHiS = kLogitHi * kLog2e;
    __shared__ NarrowLds sm;

    const int tid = threadIdx.x;
    insitu_enter(p.insitu);                        // (measurement hook, vibo_set_insitu_timer: null unless a benchmark armed it)
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int ed = j & (AT - 1);
    const bool own = j < AT;                       // the lane that accumulates / stores its person's dim ed
    const int I = p.I, A = p.A;
    const int n4 = (I + 3) >> 2;

    if (p.step_tick && blockIdx.x == 0 && tid == 0) *p.step_tick += 1;

    // ---- encoder table constants of this lane's dim (utils.py:105-113): tau = 1 / (exp(logvar) + eps), mu tau ----
    float tau[2], mt[2], te[2], mm[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float m_ = 0.f, s_ = 0.f;
        if (ed < A) { m_ = p.table[c * 2 * A + ed]; s_ = p.table[c * 2 * A + A + ed]; }
        const float es = __expf(s_);
        tau[c] = 1.0f / (es + kPoeEps);
        mt[c] = m_ * tau[c];
        te[c] = tau[c] * tau[c] * es;
        mm[c] = m_;
    }
    const float prior_w = p.missing_mode == 0 ? 1.0f / (1.0f + kPoeEps) : 0.f;

    // ---- this lane's IL items, brought to the kernel's form (log2 units: na = -a log2 e, or +log2 e for 1PL; nb = b log2 e) ----
    float na[IL][AT], nb[IL], acc_a[IL][AT], acc_b[IL];
    float gs[IL], om[IL], acc_g[IL];               // 3PL: guess, 1 - guess, d / d guess-logit
    uint32_t tail_mask[NC];
    bool chunk_ok[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int chunk = NC * j + k;
        chunk_ok[k] = chunk < n4;
        tail_mask[k] = !chunk_ok[k] ? 0u : ((I & 3) && chunk == (I >> 2)) ? ((1u << (8 * (I & 3))) - 1u) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int t = 0; t < IL; ++t) {
        const int il = IL * j + t;
        const bool ok = il < I;
        const float* ir = p.item_raw + (size_t)(p.item0 + (ok ? il : 0)) * p.D;
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            float v = 0.f;
            if (ok && a < A) v = IRT == 1 ? kLog2e : -ir[a] * kLog2e;          // models.py:731 / 744,759
            na[t][a] = v;
            acc_a[t][a] = 0.f;
        }
        nb[t] = ok ? ir[IRT == 1 ? 0 : A] * kLog2e : 0.f;
        acc_b[t] = 0.f;
        acc_g[t] = 0.f;
        float gv = 0.f;
        if (IRT == 3 && ok) gv = 1.0f / (1.0f + expf(-ir[A + 1]));            // models.py:758
        gs[t] = gv;
        om[t] = (IRT == 3 && ok) ? 1.0f - gv : (IRT == 3 ? 0.f : 1.f);
    }
    float acc_t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc_t[k] = 0.f;
    float s_log = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_nobs = 0.f;
    int unobs = 0;                                 // cells of this lane without an observation (log-lik correction, 1PL/2PL)

    // ---- rows: unit = kNwPass passes of 4 rows; a wave walks units wave_id, wave_id + n_waves, ... ----
    constexpr int UR = 4 * kNwPass;                // rows per unit
    const long long n_units = ((long long)p.B + UR - 1) / UR;
    const long long wave_id = (long long)blockIdx.x * kNwWaves + q, n_waves = (long long)gridDim.x * kNwWaves;
    // (plain arrays, not a struct of them: hipcc left a struct with the float4 rows in scratch memory)
    constexpr int NX = CODES ? 1 : kNwPass * NC, NM = kNwPass * NC;
    // Loads carry no predicate: rows past the matrix' end read its last row, chunks past the row's end its last chunk, and the
    // cells are switched off when they are packed (a predicated load is a branch around it -- 40 of them per unit serialised the
    // whole prefetch).  Rows in order: one uniform 64-bit base per unit + a 32-bit lane offset.
    int cclamp[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) cclamp[k] = min(NC * j + k, n4 - 1);
    const bool have_mask = CODES || p.mask_dtype == 0;                     // (wave-uniform)
    auto load_unit = [&](const long long un, float4 (&rx)[NX], uint32_t (&rm)[NM], float (&re)[kNwPass]) __attribute__((always_inline)) {
        const long long row0 = un * UR;                                    // (uniform)
        const int last = (int)min((long long)UR, p.B - row0) - 1;          // last existing row of the unit
        bool linear = RM == 0;
        if constexpr (RM == 2) linear = p.row_index == nullptr;
#pragma unroll
        for (int ps = 0; ps < kNwPass; ++ps) {
            const int rl = min(4 * ps + g, last);
            const float* rbase = nullptr;
            const uint8_t* mbase = nullptr;
            if (linear) {
                if constexpr (!CODES) rbase = p.response + row0 * p.resp_stride + p.item0 + (unsigned)rl * (unsigned)p.resp_stride;
                mbase = static_cast<const uint8_t*>(p.mask) + row0 * p.mask_stride + p.item0 + (unsigned)rl * (unsigned)p.mask_stride;
            } else {
                const long long src = p.row_index[row0 + rl];
                if constexpr (!CODES) rbase = p.response + src * p.resp_stride + p.item0;
                mbase = static_cast<const uint8_t*>(p.mask) + src * p.mask_stride + p.item0;
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if constexpr (!CODES) rx[ps * NC + k] = reinterpret_cast<const float4*>(rbase)[cclamp[k]];
                rm[ps * NC + k] = have_mask ? reinterpret_cast<const uint32_t*>(mbase)[cclamp[k]] : 0x01010101u;
            }
            re[ps] = p.eps[(row0 + rl) * A + min(ed, A - 1)];
        }
    };

    auto process = [&](const long long un, float4 (&rx)[NX], uint32_t (&rm)[NM], float (&re)[kNwPass]) __attribute__((always_inline)) {
        // pack first: the raw row registers die here, the next unit's loads go out into them and fly under the math
        uint32_t cw[kNwPass][NC];
        int cnt[kNwPass];
        float eps_c[kNwPass];
#pragma unroll
        for (int ps = 0; ps < kNwPass; ++ps) {
            int pk = 0;
            const bool rok = un * UR + 4 * ps + g < p.B;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const uint32_t keep = rok ? tail_mask[k] : 0u;             // (0 as well for chunks past the row's end)
                if constexpr (CODES) cw[ps][k] = pack_cell_codes4(rm[ps * NC + k], keep, pk);
                else cw[ps][k] = pack_codes4(rx[CODES ? 0 : ps * NC + k], rm[ps * NC + k] & keep, pk);
            }
            if constexpr (IRT != 3) unobs += IL - (pk & 0xffff);
            cnt[ps] = row16_sum(pk);
            eps_c[ps] = (rok && ed < A) ? re[ps] : 0.f;
        }
        if (un + n_waves < n_units) load_unit(un + n_waves, rx, rm, re);      // (into the registers the pack just freed)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ps = 0; ps < kNwPass; ++ps) {
            const long long row = un * UR + 4 * ps + g;
            const bool live = ed < A && row < p.B;
            // ---- product of experts + reparameterised sample of (person, dim ed)  (models.py:596-629) ----
            const float n1 = (float)(cnt[ps] >> 16);
            const float nobs = (float)(cnt[ps] & 0xffff);
            const float n0 = nobs - n1;
            float lam = n0 * tau[0] + n1 * tau[1];
            const float smu = n0 * mt[0] + n1 * mt[1];
            lam = fmaf((float)p.I_total - nobs, prior_w, lam);
            if (!live) lam = 1.0f;
            const float inv_lam = 1.0f / lam;
            const float amu = smu * inv_lam;
            const float sig = fast_rsq(lam);
            const float thv = live ? amu + sig * eps_c[ps] : 0.f;
            if (own && live && p.primary) {
                const long long o = row * A + ed;
                const float alv = -kLn2 * fast_log2(lam);
                p.ability_mu[o] = amu;
                p.ability_logvar[o] = alv;
                p.ability[o] = thv;
                s_kl += -0.5f * (1.0f + alv - amu * amu - inv_lam);
                s_logq0 += -0.5f * kLog2Pi - 0.5f * alv - 0.5f * eps_c[ps] * eps_c[ps];
                s_logp += -0.5f * kLog2Pi - 0.5f * thv * thv;
                if (ed == 0) s_nobs += nobs;
            }
            float th[AT];
            if constexpr (AT == 1) th[0] = thv;
            if constexpr (AT >= 2) { th[0] = quad_bcast<0>(thv); th[1] = quad_bcast<1>(thv); }
            if constexpr (AT == 4) { th[2] = quad_bcast<2>(thv); th[3] = quad_bcast<3>(thv); }

            // ---- decode, masked Bernoulli log-lik, backward: IL items of the lane ----
            float gth[AT];
#pragma unroll
            for (int a = 0; a < AT; ++a) gth[a] = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const float2v w01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)cw[ps][k], false);
                const float2v w23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)cw[ps][k], true);
                const float w[4] = {w01[0], w01[1], w23[0], w23[1]};
                float lg[4], gls[4];
                float lmax = 0.f, prod = 1.0f;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int t = 4 * k + t4;
                    float l = nb[t];
#pragma unroll
                    for (int a = 0; a < AT; ++a) l = fmaf(na[t][a], th[a], l);
                    lg[t4] = l;
                    if constexpr (IRT != 3) lmax = fmaxf(lmax, fabsf(l));
                    float gl = 0.f;
                    if constexpr (IRT != 3) {
                        const float lc = med3(l, -kLoS, kLoS);
                        const float eu = fast_exp2(-w[t4] * lc);          // exactly 1 for a missing cell (w = 0)
                        const float tt = 1.0f + eu;
                        prod *= tt;             // <= (1 + 2^23)^4: one log2 per 4 cells; the 2s of unobserved cells are taken out
                                                // of the lane's sum at the end (unobs)
                        if constexpr (GRAD) gl = w[t4] * (eu * fast_rcp(tt));   // d ll / d logit
                    } else {
                        // 3PL: p = guess + (1 - guess) sigmoid(l)  (models.py:758-765), torch's probability clamp on p itself
                        const float e = fast_exp2(-fabsf(l));
                        const float rr_ = fast_rcp(1.0f + e);
                        const float er_ = e * rr_;
                        const float sp = (l >= 0.f) ? rr_ : er_;       // sigmoid(l)
                        const float sn = (l >= 0.f) ? er_ : rr_;       // sigmoid(-l)
                        const float pr = fmaf(om[t], sp, gs[t]);       // P(correct)
                        const float qr = om[t] * sn;                   // P(wrong)
                        const float pc = med3(pr, kEps32, 1.0f - kEps32);
                        const float arg = (w[t4] > 0.f) ? pc : med3(qr, kEps32, 1.0f - kEps32);
                        prod *= (w[t4] != 0.f) ? arg : 1.0f;           // >= eps32^4: no underflow
                        if constexpr (GRAD) {
                            const float wlv = (pr == pc) ? w[t4] : 0.f;                 // the clamp kills the gradient
                            const float common = wlv * fast_rcp(arg) * om[t] * sn;     // (x/p - (1-x)/(1-p)) (1-g) sig(-l)
                            gl = common * sp;                                          // x d p / d logit
                            acc_g[t] = fmaf(common, gs[t], acc_g[t]);                  // x d p / d guess-logit
                        }
                    }
                    if constexpr (GRAD) {
                        gls[t4] = gl;
#pragma unroll
                        for (int a = 0; a < AT; ++a) {
                            gth[a] = fmaf(na[t][a], gl, gth[a]);                       // x log2 e, removed below
                            if (IRT != 1) acc_a[t][a] = fmaf(th[a], gl, acc_a[t][a]);  // = -d / d a_ia
                        }
                        acc_b[t] += gl;
                    }
                }
                s_log += fast_log2(prod);
                if constexpr (GRAD && IRT != 3) {
                    if (__any(lmax > kLoS)) {
                        // rare: the reference's gradient is exactly zero outside [-kLogitLo, kLogitHi]: take those cells back out
#pragma unroll
                        for (int t4 = 0; t4 < 4; ++t4) {
                            const int t = 4 * k + t4;
                            const float gl = (lg[t4] < -kLoS || lg[t4] > kHiS) ? -gls[t4] : 0.f;
#pragma unroll
                            for (int a = 0; a < AT; ++a) {
                                gth[a] = fmaf(na[t][a], gl, gth[a]);
                                if (IRT != 1) acc_a[t][a] = fmaf(th[a], gl, acc_a[t][a]);
                            }
                            acc_b[t] += gl;
                        }
                    }
                }
            }
            if constexpr (GRAD) {
                // ---- backward of (person, dim ed) through the sample and the product of experts ----
                float g0 = 0.f;
#pragma unroll
                for (int a = 0; a < AT; ++a) {
                    const float s_ = row16_sum(gth[a]);
                    if (a == ed) g0 = s_;
                }
                const float gz0 = live ? g0 * kLn2 : 0.f;                              // d LL / d theta
                const bool reg_on = live && p.primary;
                const float gz1 = (reg_on && p.reg_mode != 0) ? thv : 0.f;             // d REG / d theta (-log p)
                const float h = 0.5f * sig * eps_c[ps];
                float gmu[2], glv[2];
                gmu[0] = gz0;
                glv[0] = gz0 * h;
                if (p.reg_mode == 0) {
                    gmu[1] = amu;
                    glv[1] = -0.5f * (1.0f - inv_lam);
                } else {
                    gmu[1] = gz1;
                    glv[1] = gz1 * h - 0.5f;
                }
                if (!reg_on) { gmu[1] = 0.f; glv[1] = 0.f; }
                const float nn[2] = {n0, n1};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float nl = (live && own) ? nn[c] * inv_lam : 0.f;
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        acc_t[st * 4 + c * 2 + 0] = fmaf(gmu[st] * nl, tau[c], acc_t[st * 4 + c * 2 + 0]);
                        const float g_tau = nl * (gmu[st] * (mm[c] - amu) - glv[st]);
                        acc_t[st * 4 + c * 2 + 1] = fmaf(-g_tau, te[c], acc_t[st * 4 + c * 2 + 1]);
                    }
                }
            }
            // (one pass at a time: interleaved, the two passes' temporaries double the kernel's registers and halve its occupancy)
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    {
        // One unit of loads in flight per wave, 8-16 waves per CU.  Measured (same-box A/B, 535 596 x 96): a unit of one pass
        // (4 rows) runs as fast as one of two at ability_dim 1 (87 us) and 1.6x faster at ability_dim 4 (188 -> 118 us: no
        // spills, one more wave per SIMD); four passes or a second buffer of raw rows spill 150-360 registers (260 us).
        float4 xa[NX];
        uint32_t ma[NM];
        float ea[kNwPass];
        long long un = wave_id;
        if (un < n_units) load_unit(un, xa, ma, ea);
        for (; un < n_units; un += n_waves) process(un, xa, ma, ea);
    }

    // ================= workgroup reduction -> partial record (fixed order) =================
    float* out = p.partial + (size_t)blockIdx.x * p.lay.stride;
    {
        const float ll = (IRT == 3 ? kLn2 : -kLn2) * wave_total(s_log - (float)unobs);
        const float t_kl = wave_total(s_kl), t_q0 = wave_total(s_logq0), t_lp = wave_total(s_logp), t_no = wave_total(s_nobs);
        if (lane == 0) {
            sm.red[q][0] = ll; sm.red[q][1] = t_kl; sm.red[q][2] = t_q0; sm.red[q][3] = t_lp; sm.red[q][4] = 0.f; sm.red[q][5] = t_no;
            sm.red[q][6] = 0.f; sm.red[q][7] = 0.f;
        }
    }
    if constexpr (GRAD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sm.tred[q][k][lane] = acc_t[k];
        // item gradients: the four row groups of the wave hold the same items (lanes j, 16 + j, 32 + j, 48 + j)
        auto put = [&](const int row, const int t, float v) {
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) sm.item[q][row][IL * j + t] = v;
        };
#pragma unroll
        for (int t = 0; t < IL; ++t) {
            if constexpr (IRT != 1) {
#pragma unroll
                for (int a = 0; a < AT; ++a) put(a, t, -acc_a[t][a]);
            }
            put(IRT == 1 ? 0 : AT, t, acc_b[t]);
            if constexpr (IRT == 3) put(AT + 1, t, acc_g[t]);
        }
    }
    __syncthreads();
    if (tid < 8) {
        float t = 0.f;
        for (int w = 0; w < kNwWaves; ++w) t += sm.red[w][tid];
        out[tid] = (tid < 6) ? t : 0.f;
    }
    if constexpr (GRAD) {
        if (tid >= 64 && tid < 64 + 8 * A) {           // (second wave: the first one writes the scalars)
            const int a = (tid - 64) >> 3, k = (tid - 64) & 7;
            float t = 0.f;
            for (int w = 0; w < kNwWaves; ++w)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) t += sm.tred[w][k][16 * gg + a];
            const int st = k >> 2, c = (k >> 1) & 1, ms = k & 1;
            out[p.lay.off_table + (st * 2 + c) * 2 * A + ms * A + a] = t;
        }
        const int n_rows = IRT == 1 ? 1 : IRT == 2 ? A + 1 : A + 2;
        for (int e = tid; e < n_rows * 128; e += 64 * kNwWaves) {
            const int row = e >> 7, il = e & 127;
            if (il >= I) continue;
            // (record rows are the caller's item dims: a < A | b | guess; the staged rows are template-wide: a < AT | b | guess)
            const int srow = IRT == 1 ? 0 : row < A ? row : AT + (row - A);
            float t = 0.f;
            for (int w = 0; w < kNwWaves; ++w) t += sm.item[w][srow][il];
            out[p.lay.off_item + (size_t)row * p.lay.i_pad + il] = t;
        }
    }
    insitu_exit(p.insitu, gridDim.x);
}

template <int AT, int IRT, bool GRAD, int RM>
static hipError_t launch_narrow_il(const ElboParams& p, int grid, hipStream_t s) {
    if (p.I <= 64) hipLaunchKernelGGL((narrow_kernel<AT, IRT, GRAD, RM, 4>), dim3(grid), dim3(64 * kNwWaves), 0, s, p);
    else hipLaunchKernelGGL((narrow_kernel<AT, IRT, GRAD, RM, 8>), dim3(grid), dim3(64 * kNwWaves), 0, s, p);
    return hipGetLastError();
}
template <int AT, int RM>
static hipError_t launch_narrow_irt(const ElboParams& p, int irt, bool grad, int grid, hipStream_t s) {
    if (irt == 1) return grad ? launch_narrow_il<AT, 1, true, RM>(p, grid, s) : launch_narrow_il<AT, 1, false, RM>(p, grid, s);
    if (irt == 2) return grad ? launch_narrow_il<AT, 2, true, RM>(p, grid, s) : launch_narrow_il<AT, 2, false, RM>(p, grid, s);
    return grad ? launch_narrow_il<AT, 3, true, RM>(p, grid, s) : launch_narrow_il<AT, 3, false, RM>(p, grid, s);
}
template <int RM>
static hipError_t launch_narrow_at(const ElboParams& p, int irt, bool grad, int grid, hipStream_t s) {
    if (p.A <= 1) return launch_narrow_irt<1, RM>(p, irt, grad, grid, s);
    if (p.A <= 2) return launch_narrow_irt<2, RM>(p, irt, grad, grid, s);
    return launch_narrow_irt<4, RM>(p, irt, grad, grid, s);
}
// rows: fp32 in order / fp32 through p.row_index / cell codes (codes = true: through p.mask, with or without p.row_index)
hipError_t launch_elbo_narrow(const ElboParams& p, bool codes, int irt, bool grad, int grid, hipStream_t s) {
    if (p.I < 4 || p.I > 128 || p.A > 4 || p.n_flows > 0) return hipErrorInvalidValue;
    // the plain model on 1-byte masks / cell codes only: no hooks (conditional / given posterior, panels), no int64 mask
    if (p.row_cnt || p.pre_stats || p.post_coef || p.given_post || p.given_grad || p.panel_count > 1 || !p.primary) return hipErrorInvalidValue;
    if (!codes && p.mask_dtype == VIBO_MASK_I64) return hipErrorInvalidValue;
    if (codes) return launch_narrow_at<2>(p, irt, grad, grid, s);
    if (p.row_index) return launch_narrow_at<1>(p, irt, grad, grid, s);
    return launch_narrow_at<0>(p, irt, grad, grid, s);
}

}  // namespace vibo
