// vibo_split_kernel.hip -- fused ELBO forward+backward for wide ability (A = 3..8), items in lanes.
//
// Measured on gfx950 (tools/ubench): a v_mfma_f32_16x16x4_f32 retires 1024 MACs in ~32 cycles = the rate of
// plain v_fma_f32, and it does not overlap with VALU work.  So the three contractions of the tiled kernel buy
// nothing on the matrix cores, while its layout changes (LDS transposes, D-layout decode) cost VALU.  This
// kernel keeps the row kernel's mapping -- an item never leaves its lane -- and solves the register problem
// of wide A by splitting every response row over the NQ waves of a workgroup:
//   * wave q, lane l owns items 256q + 4l + {0..3} of EVERY row: item parameters (A+1 per item) and item
//     gradient accumulators (A+1 per item) stay in 8(A+1) registers; d LL / d item needs no reduction at all;
//   * rows are walked in batches of 8.  Per batch and wave: 8 x (16 B response + 4 B mask) per lane arrive as
//     coalesced loads issued one batch ahead, are packed to 8 fp8 code words, and the 8 packed counts are
//     summed over the wave by ONE 8-value butterfly (v_permlane32_swap / v_permlane16_swap / DPP, 18 ops)
//     that leaves row r's total in lanes 8r..8r+7;
//   * after one workgroup barrier lane (r, d) of every wave forms the product of experts and the sample
//     theta[r][d] (wave-redundant, 64 lanes wide); theta reaches the decode as SGPRs (v_readlane);
//   * decode / log-lik / backward per row are pure FMAs + 3 transcendentals per term; d LL / d theta is one
//     butterfly per row (A = 8) or per row pair (A = 4); a second barrier hands the NQ partials to wave 0,
//     whose lane (r, d) backpropagates through the sample and the PoE into the 8 table-gradient accumulators.
// Outputs use the same per-workgroup partial record as the other kernels (fixed order, bitwise reproducible).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "vibo_device.hpp"
#include "vibo_launch.hpp"
#include "vibo_params.hpp"

namespace vibo {

typedef unsigned uint2v __attribute__((ext_vector_type(2)));

// v_permlane{32,16}_swap through the builtin; the empty asm keeps hipcc (ROCm 7.2) from folding the two results
// into one register (it emits v_add v, v1, v1 for r[0] + r[1] otherwise).
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
    const uint2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
    asm("" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
    const uint2v r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
    asm("" : "+v"(a), "+v"(b));
}

// 8 values per lane -> lane l returns the 64-lane sum of value (l >> 3)
__device__ __forceinline__ float bfly8(const float (&v)[8], const int lane) {
    float w[4], u[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned a = __builtin_bit_cast(unsigned, v[k]), b = __builtin_bit_cast(unsigned, v[k + 4]);
        swap32(a, b);     // a = [v_k lanes 0-31 | v_k+4 lanes 0-31], b = [v_k lanes 32-63 | v_k+4 lanes 32-63]
        w[k] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned a = __builtin_bit_cast(unsigned, w[k]), b = __builtin_bit_cast(unsigned, w[k + 2]);
        swap16(a, b);     // odd 16-lane rows of a <-> even rows of b
        u[k] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
    const bool hi = (lane & 8) != 0;
    const float keep = hi ? u[1] : u[0], give = hi ? u[0] : u[1];
    float t = keep + dpp_f<0x128>(give);     // row_ror 8
    t += dpp_f<0x141>(t);                    // row_half_mirror
    t += dpp_f<0xb1>(t);                     // quad_perm [1,0,3,2]
    t += dpp_f<0x4e>(t);                     // quad_perm [2,3,0,1]
    return t;
}
__device__ __forceinline__ int bfly8(const int (&v)[8], const int lane) {
    int w[4], u[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned a = (unsigned)v[k], b = (unsigned)v[k + 4];
        swap32(a, b);
        w[k] = (int)(a + b);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned a = (unsigned)w[k], b = (unsigned)w[k + 2];
        swap16(a, b);
        u[k] = (int)(a + b);
    }
    const bool hi = (lane & 8) != 0;
    const int keep = hi ? u[1] : u[0], give = hi ? u[0] : u[1];
    int t = keep + dpp_i<0x128>(give);
    t += dpp_i<0x141>(t);
    t += dpp_i<0xb1>(t);
    t += dpp_i<0x4e>(t);
    return t;
}

// AT = template ability width (4 or 8; runtime p.A <= AT), NQ = waves per row (I <= 256 NQ)
template <int AT, int IRT, int NQ, bool GRAD>
__global__ __launch_bounds__(64 * NQ, 2) void split_kernel(const ElboParams p) {
    constexpr int R = 8;                   // rows per batch
    constexpr int RPB = 8 / AT;            // rows per d LL/d theta butterfly
    constexpr int NE = R * AT;             // (row, dim) lanes used by the per-person math
    constexpr float kLoS = kLogitLo * kLog2e, kHiS = kLogitHi * kLog2e;
    __shared__ int cntp[NQ][R];
    __shared__ float gthp[NQ][64];
    __shared__ float ctab[4 * 2 * AT];
    __shared__ float red[NQ][8];
    __shared__ float tred[8][64];
    __shared__ uint32_t codes[NQ][R][64];  // this batch's fp8 code words (read back one row at a time)
    __shared__ __attribute__((aligned(16))) float thl[NQ][64];         // theta[r][d] of the batch, per wave
    __shared__ __attribute__((aligned(16))) float gtl[NQ][2][8][68];   // d LL/d theta partials, transposed

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I = p.I, A = p.A;
    const int n4 = I >> 2;
    const int chunk = q * 64 + lane;                      // float4 chunk of the row this lane owns
    const bool chunk_ok = chunk < n4;

    if (tid < 2 * AT) {
        const int c = tid / AT, a = tid % AT;
        float m = 0.f, s = 0.f;
        if (a < A) { m = p.table[c * 2 * A + a]; s = p.table[c * 2 * A + A + a]; }
        const float es = __expf(s);
        const float tau = 1.0f / (es + kPoeEps);
        ctab[(0 * 2 + c) * AT + a] = tau;
        ctab[(1 * 2 + c) * AT + a] = m * tau;
        ctab[(2 * 2 + c) * AT + a] = tau * tau * es;
        ctab[(3 * 2 + c) * AT + a] = m;
    }

    // ---- this lane's 4 items (log2 units: rows prepped by item_prep_kernel) ----
    float2v na2[4][AT / 2];
    float nb[4];
    float2v acc_a2[4][AT / 2];
    float acc_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * chunk + j;
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            na2[j][a >> 1][a & 1] = chunk_ok ? p.item_prep[(size_t)i * p.DP + a] : 0.f;
            acc_a2[j][a >> 1][a & 1] = 0.f;
        }
        nb[j] = chunk_ok ? p.item_prep[(size_t)i * p.DP + AT] : 0.f;
        acc_b[j] = 0.f;
    }
    // lane e = (er, ed): person er of the batch, ability dim ed
    const int er = (lane / AT) & (R - 1), ed = lane % AT;
    const bool e_ok = lane < NE && ed < A;
    float acc_t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc_t[k] = 0.f;
    float s_log = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_nobs = 0.f;
    __syncthreads();
    const float tau0 = ctab[(0 * 2 + 0) * AT + ed], tau1 = ctab[(0 * 2 + 1) * AT + ed];
    const float mt0 = ctab[(1 * 2 + 0) * AT + ed], mt1 = ctab[(1 * 2 + 1) * AT + ed];

    const long long n_batches = ((long long)p.B + R - 1) / R;
    float4 x[R];
    uint32_t m[R];
    float epn = 0.f;
    auto load_batch = [&](const long long bt) {
        const long long row0 = bt * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long row = row0 + r;
            x[r] = float4{0.f, 0.f, 0.f, 0.f};
            m[r] = 0u;
            if (row < p.B && chunk_ok) {
                const long long src = p.row_index ? p.row_index[row] : row;
                x[r] = reinterpret_cast<const float4*>(p.response + src * p.resp_stride)[chunk];
                if (p.mask_dtype == 0)
                    m[r] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride)[chunk];
                else
                    m[r] = 0x01010101u;
            }
        }
        const long long erow = row0 + er;
        epn = (e_ok && erow < p.B) ? p.eps[erow * A + ed] : 0.f;
    };

    long long bt = blockIdx.x;
    if (bt < n_batches) load_batch(bt);
    for (; bt < n_batches; bt += gridDim.x) {
        const long long row0 = bt * R;
        // ---- pack the batch to fp8 codes (+1 correct / -1 wrong / 0 missing); the raw row registers die here,
        //      so the next batch's HBM loads are issued into them and fly under this batch's math
        int pk[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            pk[r] = 0;
            codes[q][r][lane] = pack_codes4(x[r], m[r], pk[r]);
        }
        const float eps_c = epn;
        if (bt + gridDim.x < n_batches) load_batch(bt + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int tot = bfly8(pk, lane);
            if ((lane & 7) == 0) {
                cntp[q][lane >> 3] = tot;
                // every lane-cell without an observation (missing, padding, rows past the end) contributes
                // log2(1 + 2^0) = 1 to the running log-lik sum below: take those out here, once per row
                s_log -= (float)(256 - (tot & 0xffff));
            }
        }
        __syncthreads();

        // ---- product of experts + reparameterised sample for (person er, dim ed)  (models.py:596-629) ----
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < NQ; ++w) cnt += cntp[w][er];
        const float n1 = (float)(cnt >> 16), nobs = (float)(cnt & 0xffff);
        const float n0 = nobs - n1, nmiss = (float)I - nobs;
        const bool live = e_ok && (row0 + er) < p.B;
        float lam = n0 * tau0 + n1 * tau1;
        if (p.missing_mode == 0) lam += nmiss * (1.0f / (1.0f + kPoeEps));
        if (!live) lam = 1.0f;              // rows past the end / padded dims: keep the arithmetic finite
        const float inv_lam = 1.0f / lam;
        const float amu = (n0 * mt0 + n1 * mt1) * inv_lam;
        const float sig = fast_rsq(lam);
        float thv = amu + sig * eps_c;
        if (!live) thv = 0.f;
        if (q == 0 && live) {
            const long long o = (row0 + er) * A + ed;
            const float alv = -kLn2 * fast_log2(lam);
            p.ability_mu[o] = amu;
            p.ability_logvar[o] = alv;
            p.ability[o] = thv;
            s_kl += -0.5f * (1.0f + alv - amu * amu - inv_lam);
            s_logq0 += -0.5f * kLog2Pi - 0.5f * alv - 0.5f * eps_c * eps_c;
            s_logp += -0.5f * kLog2Pi - 0.5f * thv * thv;
            if (ed == 0) s_nobs += nobs;
        }

        // ---- decode, masked Bernoulli log-lik, backward: 4 items per lane ----
        // theta goes through a wave-private LDS row and comes back as broadcast reads (v_readlane costs 3 VALU
        // slots each, tools/ubench2).  The reference clamps the Bernoulli probability (utils.py:46-49 -> torch):
        // log-lik value clamped at logit +-kLogitLo, gradient exactly zero outside [-kLogitLo, kLogitHi]; that
        // is rare, so the clamped arithmetic lives in a second copy of the row body behind a wave-uniform branch.
        thl[q][lane] = thv;
        // row group = RPB rows (one d LL/d theta reduction).  Software pipeline over groups: the LDS reads of
        // group g+1 (codes, theta) and the transposed partials of group g-1 are issued before group g's math.
        struct GroupIn {
            uint32_t cw[RPB];
            float2v th2[RPB][AT / 2];
        };
        auto fetch_group = [&](const int g, GroupIn& gi) {
#pragma unroll
            for (int rr = 0; rr < RPB; ++rr) {
                const int r = g * RPB + rr;
                gi.cw[rr] = codes[q][r][lane];
                if constexpr (AT >= 4) {
#pragma unroll
                    for (int a = 0; a < AT; a += 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(&thl[q][r * AT + a]);
                        gi.th2[rr][a / 2] = float2v{t4.x, t4.y};
                        gi.th2[rr][a / 2 + 1] = float2v{t4.z, t4.w};
                    }
                } else {
                    gi.th2[rr][0] = *reinterpret_cast<const float2v*>(&thl[q][r * AT]);
                }
            }
        };
        auto do_group = [&](const int g, const GroupIn& gi, GroupIn& nxt) {
            constexpr int G = R / RPB;
            if (g + 1 < G) fetch_group(g + 1, nxt);
            float4 ru = float4{0.f, 0.f, 0.f, 0.f}, rv = ru;
            if constexpr (GRAD) {
                if (g > 0) {
                    // transposed read-back of group g-1's 8 x 64 partials: lane (k = l >> 3, s = l & 7) sums 8
                    // lanes of value k, then 3 DPP steps finish the 64-lane sum
                    const float* src = &gtl[q][(g - 1) & 1][lane >> 3][(lane & 7) * 8];
                    ru = *reinterpret_cast<const float4*>(src);
                    rv = *reinterpret_cast<const float4*>(src + 4);
                }
            }
            float2v gth2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) gth2[k] = float2v{0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < RPB; ++rr) {
                const float2v(&th2)[AT / 2] = gi.th2[rr];
                float lg[4];
                float lmax = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float2v l2 = float2v{nb[t], 0.f};
#pragma unroll
                    for (int h = 0; h < AT / 2; ++h) l2 = na2[t][h] * th2[h] + l2;
                    lg[t] = l2[0] + l2[1];
                    lmax = fmaxf(lmax, fabsf(lg[t]));
                }
                const float2v w01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)gi.cw[rr], false);
                const float2v w23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)gi.cw[rr], true);
                const float w[4] = {w01[0], w01[1], w23[0], w23[1]};
                float prod = 1.0f;
                float gls[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // value: the clamp at +-kLogitLo is exactly the reference's; gradient: see the fix-up below
                    const float lc = med3(lg[t], -kLoS, kLoS);
                    const float eu = fast_exp2(-w[t] * lc);           // exactly 1 for a missing cell (w = 0)
                    const float tt = 1.0f + eu;
                    prod *= tt;                 // <= (1 + 2^23)^4: one log2 per 4 terms; the 2s of missing cells
                                                // are taken out per batch (s_log correction above)
                    if constexpr (GRAD) {
                        const float gl = w[t] * (eu * fast_rcp(tt));  // d ll / d logit
                        gls[t] = gl;
#pragma unroll
                        for (int h = 0; h < AT / 2; ++h) {
                            gth2[rr * (AT / 2) + h] = na2[t][h] * gl + gth2[rr * (AT / 2) + h];   // x log2e, removed below
                            if (IRT != 1) acc_a2[t][h] = th2[h] * gl + acc_a2[t][h];              // = -d/d a_ia
                        }
                        acc_b[t] += gl;
                    }
                }
                s_log += fast_log2(prod);
                if constexpr (GRAD) {
                    if (__any(lmax > kLoS)) {
                        // rare: the reference's gradient is exactly zero outside [-kLogitLo, kLogitHi]; take the
                        // contributions of those cells back out
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float gl = (lg[t] < -kLoS || lg[t] > kHiS) ? -gls[t] : 0.f;
#pragma unroll
                            for (int h = 0; h < AT / 2; ++h) {
                                gth2[rr * (AT / 2) + h] = na2[t][h] * gl + gth2[rr * (AT / 2) + h];
                                if (IRT != 1) acc_a2[t][h] = th2[h] * gl + acc_a2[t][h];
                            }
                            acc_b[t] += gl;
                        }
                    }
                }
            }
            if constexpr (GRAD) {
                if (g > 0) {
                    float t = ((ru.x + ru.y) + (ru.z + ru.w)) + ((rv.x + rv.y) + (rv.z + rv.w));
                    t += dpp_f<0xb1>(t);                     // quad_perm [1,0,3,2]
                    t += dpp_f<0x4e>(t);                     // quad_perm [2,3,0,1]
                    t += dpp_f<0x141>(t);                    // row_half_mirror
                    if ((lane & 7) == 0) gthp[q][(g - 1) * 8 + (lane >> 3)] = t;
                }
                // 64-lane sums of the 8 partials go through an LDS transpose, one group behind the math
#pragma unroll
                for (int k = 0; k < 8; ++k) gtl[q][g & 1][k][lane] = gth2[k >> 1][k & 1];
            }
        };
        {
            GroupIn ga, gb;
            fetch_group(0, ga);
#pragma unroll 1
            for (int g = 0; g < R / RPB; g += 2) {
                do_group(g, ga, gb);
                if (g + 1 < R / RPB) do_group(g + 1, gb, ga);
            }
        }
        if constexpr (GRAD) {
            constexpr int g = R / RPB - 1;
            const float* src = &gtl[q][g & 1][lane >> 3][(lane & 7) * 8];
            const float4 ru = *reinterpret_cast<const float4*>(src), rv = *reinterpret_cast<const float4*>(src + 4);
            float t = ((ru.x + ru.y) + (ru.z + ru.w)) + ((rv.x + rv.y) + (rv.z + rv.w));
            t += dpp_f<0xb1>(t);
            t += dpp_f<0x4e>(t);
            t += dpp_f<0x141>(t);
            if ((lane & 7) == 0) gthp[q][g * 8 + (lane >> 3)] = t;
        }
        if constexpr (GRAD) {
            __syncthreads();
            // ---- wave 0, lane (er, ed): backward through the sample and the PoE into the table gradients ----
            if (q == 0) {
                float g0 = 0.f;
#pragma unroll
                for (int w = 0; w < NQ; ++w) g0 += gthp[w][lane & (NE - 1)];
                g0 = live ? g0 * kLn2 : 0.f;
                const float h = 0.5f * sig * eps_c;
                float gmu[2], glv[2];
                gmu[0] = g0;
                glv[0] = g0 * h;
                if (p.reg_mode == 0) {
                    gmu[1] = amu;
                    glv[1] = -0.5f * (1.0f - inv_lam);
                } else {
                    gmu[1] = thv;
                    glv[1] = thv * h - 0.5f;
                }
                if (!live) { gmu[1] = 0.f; glv[1] = 0.f; }
                const float nn[2] = {n0, n1};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float tau = c ? tau1 : tau0;
                    const float te = ctab[(2 * 2 + c) * AT + ed], mm = ctab[(3 * 2 + c) * AT + ed];
                    const float nl = live ? nn[c] * inv_lam : 0.f;
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        acc_t[st * 4 + c * 2 + 0] = fmaf(gmu[st] * nl, tau, acc_t[st * 4 + c * 2 + 0]);
                        const float g_tau = nl * (gmu[st] * (mm - amu) - glv[st]);
                        acc_t[st * 4 + c * 2 + 1] = fmaf(-g_tau, te, acc_t[st * 4 + c * 2 + 1]);
                    }
                }
            }
        }
    }

    // ================= workgroup reduction -> partial record ======
    float* out = p.partial + (size_t)blockIdx.x * p.lay.stride;
    {
        const float ll = -(kLn2 * wave_total(s_log));
        const float t_kl = wave_total(s_kl), t_q0 = wave_total(s_logq0), t_lp = wave_total(s_logp), t_no = wave_total(s_nobs);
        if (lane == 0) {
            red[q][0] = ll; red[q][1] = t_kl; red[q][2] = t_q0; red[q][3] = t_lp; red[q][4] = 0.f; red[q][5] = t_no;
            red[q][6] = 0.f; red[q][7] = 0.f;
        }
        if (q == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) tred[k][lane] = acc_t[k];
        }
    }
    __syncthreads();
    if (tid < 8) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NQ; ++w) t += red[w][tid];
        out[tid] = (tid < 6) ? t : 0.f;
    }
    if constexpr (GRAD) {
        if (tid >= 64 * (NQ - 1) && lane < 8 * A) {        // last wave: lane = a * 8 + k
            const int a = lane >> 3, k = lane & 7;
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) t += tred[k][r * AT + a];
            const int st = k >> 2, c = (k >> 1) & 1, ms = k & 1;
            out[p.lay.off_table + (st * 2 + c) * 2 * A + ms * A + a] = t;
        }
        if (chunk_ok) {
            float* oi = out + p.lay.off_item + 4 * chunk;
            if (IRT == 1) {
                *reinterpret_cast<float4*>(oi) = float4{acc_b[0], acc_b[1], acc_b[2], acc_b[3]};
            } else {
#pragma unroll
                for (int a = 0; a < AT; ++a)
                    if (a < A)
                        *reinterpret_cast<float4*>(oi + (size_t)a * p.lay.i_pad) =
                            float4{-acc_a2[0][a >> 1][a & 1], -acc_a2[1][a >> 1][a & 1], -acc_a2[2][a >> 1][a & 1], -acc_a2[3][a >> 1][a & 1]};
                *reinterpret_cast<float4*>(oi + (size_t)A * p.lay.i_pad) = float4{acc_b[0], acc_b[1], acc_b[2], acc_b[3]};
            }
        }
    }
}

template <int AT, int IRT, bool GRAD>
static hipError_t launch_split_nq(const ElboParams& p, int nq, int grid, hipStream_t s) {
    switch (nq) {
        case 1: hipLaunchKernelGGL((split_kernel<AT, IRT, 1, GRAD>), dim3(grid), dim3(64), 0, s, p); break;
        case 2: hipLaunchKernelGGL((split_kernel<AT, IRT, 2, GRAD>), dim3(grid), dim3(128), 0, s, p); break;
        case 3: hipLaunchKernelGGL((split_kernel<AT, IRT, 3, GRAD>), dim3(grid), dim3(192), 0, s, p); break;
        default: hipLaunchKernelGGL((split_kernel<AT, IRT, 4, GRAD>), dim3(grid), dim3(256), 0, s, p); break;
    }
    return hipGetLastError();
}

// ability_dim 3..8 (template widths 4 / 8), irt in {1,2}, I <= 1024, I % 4 == 0, 16-byte aligned rows,
// mask u8 or none.  nq = ceil(I / 256) waves per workgroup.
hipError_t launch_elbo_split(const ElboParams& p, int at, int irt, bool grad, int nq, int grid, hipStream_t s) {
    if (at <= 2) {
        if (irt == 1) return grad ? launch_split_nq<2, 1, true>(p, nq, grid, s) : launch_split_nq<2, 1, false>(p, nq, grid, s);
        return grad ? launch_split_nq<2, 2, true>(p, nq, grid, s) : launch_split_nq<2, 2, false>(p, nq, grid, s);
    }
    if (at == 4) {
        if (irt == 1) return grad ? launch_split_nq<4, 1, true>(p, nq, grid, s) : launch_split_nq<4, 1, false>(p, nq, grid, s);
        return grad ? launch_split_nq<4, 2, true>(p, nq, grid, s) : launch_split_nq<4, 2, false>(p, nq, grid, s);
    }
    if (irt == 1) return grad ? launch_split_nq<8, 1, true>(p, nq, grid, s) : launch_split_nq<8, 1, false>(p, nq, grid, s);
    return grad ? launch_split_nq<8, 2, true>(p, nq, grid, s) : launch_split_nq<8, 2, false>(p, nq, grid, s);
}

}  // namespace vibo
