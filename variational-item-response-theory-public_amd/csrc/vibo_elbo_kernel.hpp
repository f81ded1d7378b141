// vibo_elbo_kernel.hpp -- fused VIBO ELBO forward+backward kernel for gfx950 (MI355X).
//
// One workgroup (NW waves) walks 64-person tiles of the response matrix:
//
//  phase A  "load + pack"  (lane = item chunk, coalesced):  every wave streams its
//           rows of the tile from HBM with 16-byte loads (float4 response + 4 mask
//           bytes per lane), packs each cell into ONE fp8 byte  w = +1 (correct) /
//           -1 (wrong) / 0 (missing)  in an LDS tile, and wave-reduces the per-row
//           counts (n_correct, n_observed) the unconditional product of experts
//           needs (models.py:596-629 collapses to those counts).
//  phase B  "person per lane": lane p owns person p of the tile (theta, dLL/dtheta
//           in registers); the NW waves split the items in 64-item blocks; item
//           parameters are wave-uniform and arrive through scalar loads; per term:
//           logit -> masked Bernoulli log-lik (softplus form, reference clamp
//           semantics) -> dLL/dlogit -> dLL/dtheta (in lane) and dLL/ditem (DPP
//           reduction over the 64 persons, parked in the lane that owns the item).
//  epilogue the backward through the reparameterised sample and the product of
//           experts is LINEAR in dLL/dtheta, so every wave applies it to its own
//           partial dLL/dtheta and accumulates table gradients privately: no
//           cross-wave reduction per tile.  The regulariser side (KL / log q - log p)
//           and the [B,A] posterior outputs are split across waves by ability dim.
//
// Each response row is read from HBM exactly once.  All reductions have a fixed
// order for a fixed grid, so results are bitwise reproducible.
#pragma once
#include "vibo_device.hpp"
#include "vibo_params.hpp"

namespace vibo {

// LDS constant table (per kernel, built once): for c in {0,1}, a < A
//   TAU = 1/(exp(s_ca)+1e-8)   MTAU = m_ca*TAU   TE = TAU^2*exp(s_ca)   M = m_ca
enum { CT_TAU = 0, CT_MTAU = 1, CT_TE = 2, CT_M = 3 };

template <int A>
struct PersonDim {
    float lam, inv_lam, amu, sig, eps, n0, n1;
};

// ---------------------------------------------------------------------------
// phase A
// ---------------------------------------------------------------------------
template <int MK>
__device__ __forceinline__ uint32_t load_mask4(const ElboParams& p, long long src, int q) {
    if constexpr (MK == 0) {          // u8 / bool, values 0|1
        const uint32_t* mp = reinterpret_cast<const uint32_t*>(
            static_cast<const uint8_t*>(p.mask) + src * p.mask_stride);
        return mp[q];
    } else if constexpr (MK == 1) {   // int64, nonzero = observed
        const longlong2* mp = reinterpret_cast<const longlong2*>(
            static_cast<const int64_t*>(p.mask) + src * p.mask_stride);
        const longlong2 a = mp[2 * q], b = mp[2 * q + 1];
        return (a.x != 0 ? 1u : 0u) | (a.y != 0 ? 0x100u : 0u) | (b.x != 0 ? 0x10000u : 0u) |
               (b.y != 0 ? 0x1000000u : 0u);
    } else {
        return 0x01010101u;
    }
}

template <int MK>
__device__ __forceinline__ uint32_t load_mask1(const ElboParams& p, long long src, int q) {
    if constexpr (MK == 0) {
        return static_cast<const uint8_t*>(p.mask)[src * p.mask_stride + q] != 0 ? 1u : 0u;
    } else if constexpr (MK == 1) {
        return static_cast<const int64_t*>(p.mask)[src * p.mask_stride + q] != 0 ? 1u : 0u;
    } else {
        return 1u;
    }
}

// 4 responses (fp32 0.0/1.0) + 4 mask bytes (0/1) -> 4 fp8 codes; counts packed n1<<16 | nobs.
__device__ __forceinline__ uint32_t pack_codes4(const float4 x, const uint32_t m, int& packed) {
    const uint32_t x0 = __builtin_bit_cast(uint32_t, x.x), x1 = __builtin_bit_cast(uint32_t, x.y);
    const uint32_t x2 = __builtin_bit_cast(uint32_t, x.z), x3 = __builtin_bit_cast(uint32_t, x.w);
    // byte 3 of 1.0f is 0x3F, of 0.0f is 0x00: bit 24 tells "correct"
    const uint32_t hi = __builtin_amdgcn_perm(x1, x0, 0x0c0c0703u) | __builtin_amdgcn_perm(x3, x2, 0x07030c0cu);
    const uint32_t xb = hi & 0x01010101u;
    const uint32_t code = (0xB8B8B8B8u ^ (xb << 7)) & (m * 0xFFu);
    packed += __builtin_popcount(m) + (__builtin_popcount(xb & m) << 16);
    return code;
}

template <int NW, int CMAX, int MK>
__device__ __forceinline__ void phase_a_vec(const ElboParams& p, const int tile, const int wave, const int lane,
                                            unsigned char* codes, uint32_t* counts) {
    constexpr int RPW = kTilePersons / NW;
    const int n4 = p.I >> 2;
#pragma unroll 1
    for (int j = 0; j < RPW; ++j) {
        const int r = wave * RPW + j;
        const long long grow = (long long)tile * kTilePersons + r;
        uint32_t* dst = reinterpret_cast<uint32_t*>(codes + r * p.lds_stride);
        int packed = 0;
        if (grow < p.B) {
            const long long src = p.row_index ? p.row_index[grow] : grow;
            const float4* rp = reinterpret_cast<const float4*>(p.response + src * p.resp_stride);
            float4 x[CMAX];
            uint32_t m[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                const int q = lane + 64 * c;
                if (q < n4) {
                    x[c] = rp[q];
                    m[c] = load_mask4<MK>(p, src, q);
                }
            }
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                const int q = lane + 64 * c;
                if (q < n4) dst[q] = pack_codes4(x[c], m[c], packed);
            }
        } else {
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                const int q = lane + 64 * c;
                if (q < n4) dst[q] = 0u;
            }
        }
        packed = wave_sum63(packed);
        if (lane == 63) counts[r] = (uint32_t)packed;
    }
}

// ragged / unaligned rows: one cell per lane per step
template <int NW, int MK>
__device__ __forceinline__ void phase_a_scalar(const ElboParams& p, const int tile, const int wave, const int lane,
                                               unsigned char* codes, uint32_t* counts) {
    constexpr int RPW = kTilePersons / NW;
#pragma unroll 1
    for (int j = 0; j < RPW; ++j) {
        const int r = wave * RPW + j;
        const long long grow = (long long)tile * kTilePersons + r;
        unsigned char* dst = codes + r * p.lds_stride;
        int packed = 0;
        if (grow < p.B) {
            const long long src = p.row_index ? p.row_index[grow] : grow;
            const float* rp = p.response + src * p.resp_stride;
#pragma unroll 4
            for (int q = lane; q < p.I; q += 64) {
                const uint32_t k = load_mask1<MK>(p, src, q);
                const uint32_t xb = (rp[q] == 1.0f) ? 1u : 0u;
                dst[q] = (unsigned char)(k ? (xb ? 0x38u : 0xB8u) : 0u);
                packed += (int)k + (int)((xb & k) << 16);
            }
        } else {
            for (int q = lane; q < p.I; q += 64) dst[q] = 0;
        }
        packed = wave_sum63(packed);
        if (lane == 63) counts[r] = (uint32_t)packed;
    }
}

// ---------------------------------------------------------------------------
// per-person posterior for one ability dim (product of experts on counts)
// ---------------------------------------------------------------------------
template <int A>
__device__ __forceinline__ PersonDim<A> person_dim(const ElboParams& p, const float* ctab, const uint32_t cnt,
                                                   const long long grow, const bool valid, const int a) {
    PersonDim<A> d;
    d.n1 = (float)(cnt >> 16);
    const float nobs = (float)(cnt & 0xffffu);
    d.n0 = nobs - d.n1;
    const float nmiss = (float)p.I - nobs;
    const float tau0 = ctab[(CT_TAU * 2 + 0) * A + a], tau1 = ctab[(CT_TAU * 2 + 1) * A + a];
    const float mt0 = ctab[(CT_MTAU * 2 + 0) * A + a], mt1 = ctab[(CT_MTAU * 2 + 1) * A + a];
    float lam = d.n0 * tau0 + d.n1 * tau1;
    if (p.missing_mode == 0) lam += nmiss * (1.0f / (1.0f + kPoeEps));   // N(0,1) prior experts
    const float s = d.n0 * mt0 + d.n1 * mt1;
    d.lam = lam;
    d.inv_lam = 1.0f / lam;
    d.amu = s * d.inv_lam;
    d.sig = fast_rsq(lam);
    d.eps = valid ? p.eps[grow * p.A + a] : 0.0f;
    return d;
}

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
template <int A, int IRT, int NW, int SLOTS, bool GRAD>
__global__ __launch_bounds__(NW * 64, 4) void elbo_kernel(const ElboParams p) {
    constexpr int DT = (IRT == 1) ? 1 : (IRT == 2 ? A + 1 : A + 2);   // item-grad accumulators per item
    constexpr int AP = (A + 1) / 2;                                   // float2 pairs
    constexpr int CMAX = (NW == 8) ? 2 * SLOTS : 2;                   // float4 chunks (of 64 lanes) per row
    constexpr int DPW = (A + NW - 1) / NW;                            // ability dims a wave owns (reg side)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* codes = smem;
    uint32_t* counts = reinterpret_cast<uint32_t*>(smem + kTilePersons * p.lds_stride);   // [2][64]
    float* ctab = reinterpret_cast<float*>(counts + 2 * kTilePersons);                     // [4][2][A]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I = p.I;
    const int Ar = p.A;
    const int stride = p.lds_stride;

    // ---- encoder-table constants -> LDS ------------------------------------
    if (tid < 2 * A) {
        const int c = tid / A, a = tid % A;
        float m = 0.f, s = 0.f;
        if (a < Ar) {
            m = p.table[c * 2 * Ar + a];
            s = p.table[c * 2 * Ar + Ar + a];
        }
        const float es = __expf(s);
        const float tau = 1.0f / (es + kPoeEps);
        ctab[(CT_TAU * 2 + c) * A + a] = tau;
        ctab[(CT_MTAU * 2 + c) * A + a] = m * tau;
        ctab[(CT_TE * 2 + c) * A + a] = tau * tau * es;
        ctab[(CT_M * 2 + c) * A + a] = m;
    }

    // ---- persistent per-lane accumulators -----------------------------------
    float acc_item[SLOTS][DT];
    float acc_t0[A][4];      // d LL / d table  : [a][c*2 + {m,s}]
    float acc_t1[DPW][4];    // d REG / d table
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int d = 0; d < DT; ++d) acc_item[s][d] = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc_t0[a][k] = 0.f;
#pragma unroll
    for (int a = 0; a < DPW; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc_t1[a][k] = 0.f;
    float s_lin = 0.f, s_log = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_nobs = 0.f;

    int buf = 0;
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x, buf ^= 1) {
        uint32_t* cnt_buf = counts + buf * kTilePersons;
        // ================= phase A ==========================================
        if (p.vec_ok) {
            if (p.mask_dtype == 0) phase_a_vec<NW, CMAX, 0>(p, tile, wave, lane, codes, cnt_buf);
            else if (p.mask_dtype == 1) phase_a_vec<NW, CMAX, 1>(p, tile, wave, lane, codes, cnt_buf);
            else phase_a_vec<NW, CMAX, 2>(p, tile, wave, lane, codes, cnt_buf);
        } else {
            if (p.mask_dtype == 0) phase_a_scalar<NW, 0>(p, tile, wave, lane, codes, cnt_buf);
            else if (p.mask_dtype == 1) phase_a_scalar<NW, 1>(p, tile, wave, lane, codes, cnt_buf);
            else phase_a_scalar<NW, 2>(p, tile, wave, lane, codes, cnt_buf);
        }
        __syncthreads();

        // ================= phase B ==========================================
        const long long grow = (long long)tile * kTilePersons + lane;
        const bool valid = grow < p.B;
        const uint32_t cnt = cnt_buf[lane];

        float th[2 * AP];
#pragma unroll
        for (int a = 0; a < 2 * AP; ++a) th[a] = 0.f;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            if (a < Ar) {
                const PersonDim<A> d = person_dim<A>(p, ctab, cnt, grow, valid, a);
                th[a] = valid ? (d.amu + d.sig * d.eps) : 0.f;
            }
        }
        float th_sum = 0.f;   // 1PL: logit = sum_a theta_a + b
#pragma unroll
        for (int a = 0; a < A; ++a) th_sum += th[a];
        float2v th2[AP];
#pragma unroll
        for (int j = 0; j < AP; ++j) th2[j] = float2v{th[2 * j], th[2 * j + 1]};

        float2v gth2[AP];     // d LL / d theta (this wave's items only)
#pragma unroll
        for (int j = 0; j < AP; ++j) gth2[j] = float2v{0.f, 0.f};
        float gth_sum = 0.f;  // 1PL

        const unsigned char* my_codes = codes + lane * stride;
        for (int s = 0; s < SLOTS; ++s) {
            const int gb = wave + NW * s;
            if (gb >= p.item_blocks) break;
            float cur[DT];   // item-grad sums of this 64-item block (lane j <-> item gb*64+j)
#pragma unroll
            for (int d = 0; d < DT; ++d) cur[d] = 0.f;
            for (int q = 0; q < 4; ++q) {
                const int i0 = gb * 64 + q * 16;
                if (i0 >= I) break;
                const uint4 cw = *reinterpret_cast<const uint4*>(my_codes + i0);
#pragma unroll 1
                for (int wq = 0; wq < 4; ++wq) {
                    if (i0 + 4 * wq >= I) break;
                    const uint32_t word = (wq == 0) ? cw.x : (wq == 1) ? cw.y : (wq == 2) ? cw.z : cw.w;
                    const int lane_rel = lane - (q * 16 + wq * 4);   // owner lane of item j is lane_rel == j
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + 4 * wq + j;
                    if (i < I) {
                        float w;
                        switch (j) {
                            case 0: w = code_to_f32<0>(word); break;
                            case 1: w = code_to_f32<1>(word); break;
                            case 2: w = code_to_f32<2>(word); break;
                            default: w = code_to_f32<3>(word); break;
                        }
                        const const_f32_ptr ip = as_constant(p.item_prep + (size_t)i * p.DP);   // wave-uniform -> s_load
                        // ---- logit ----
                        float l;
                        if constexpr (IRT == 1) {
                            l = ip[0] + th_sum;
                        } else if constexpr (A == 1) {
                            l = fmaf(ip[0], th[0], ip[1]);
                        } else {
                            float2v acc = float2v{ip[A], 0.f};
#pragma unroll
                            for (int jj = 0; jj < AP; ++jj)
                                acc = float2v{ip[2 * jj], ip[2 * jj + 1]} * th2[jj] + acc;
                            l = acc.x + acc.y;
                        }
                        float gl;   // d ll / d logit
                        float gguess = 0.f;
                        if constexpr (IRT != 3) {
                            // reference clamp (utils.py:46-49 via torch Bernoulli): value clamped at
                            // +-kLogitLo, gradient exactly zero outside [-kLogitLo, kLogitHi]
                            const float l2 = med3(l, -kLogitLo, kLogitHi);
                            const float lc = fminf(l2, kLogitLo);
                            const float u = -w * lc;                       // -(signed logit)
                            const float e = fast_exp2(-fabsf(lc) * kLog2e);
                            const float t = 1.0f + e;
                            s_lin += fmaxf(u, 0.f);
                            s_log = fmaf(fabsf(w), fast_log2(t), s_log);
                            if constexpr (GRAD) {
                                const float r = fast_rcp(t);
                                const float sg = (u >= 0.f) ? r : e * r;   // sigmoid(u)
                                const float wg = (l == l2) ? w : 0.f;
                                gl = wg * sg;
                            }
                        } else {
                            const float guess = ip[A + 1], omg = ip[A + 2];
                            const float e = fast_exp2(-fabsf(l) * kLog2e);
                            const float r = fast_rcp(1.0f + e);
                            const float er = e * r;
                            const float sp = (l >= 0.f) ? r : er;          // sigmoid(l)
                            const float sn = (l >= 0.f) ? er : r;          // sigmoid(-l)
                            const float pr = fmaf(omg, sp, guess);         // P(correct)
                            const float qr = omg * sn;                     // P(wrong)
                            const float pc = med3(pr, kEps32, 1.0f - kEps32);
                            const float arg = (w > 0.f) ? pc : med3(qr, kEps32, 1.0f - kEps32);
                            s_log = fmaf(fabsf(w), fast_log2(arg), s_log);
                            if constexpr (GRAD) {
                                const float wl = (pr == pc) ? w : 0.f;     // clamp kills the gradient
                                const float dll_dp = wl * fast_rcp(arg);   // x/p - (1-x)/(1-p)
                                const float common = dll_dp * omg * sn;
                                gl = common * sp;                          // * d p / d logit
                                gguess = common * guess;                   // * d p / d guess-logit
                            }
                        }
                        if constexpr (GRAD) {
                            // ---- d LL / d theta (kept in lane) ----
                            if constexpr (IRT == 1) {
                                gth_sum += gl;
                            } else if constexpr (A == 1) {
                                gth2[0].x = fmaf(gl, ip[0], gth2[0].x);
                            } else {
                                const float2v g2 = float2v{gl, gl};
#pragma unroll
                                for (int jj = 0; jj < AP; ++jj)
                                    gth2[jj] = g2 * float2v{ip[2 * jj], ip[2 * jj + 1]} + gth2[jj];
                            }
                            // ---- d LL / d item: reduce over the 64 persons, park in the owner lane ----
                            const float sel = (lane_rel == j) ? 1.0f : 0.f;
                            float v[DT];
                            if constexpr (IRT == 1) {
                                v[0] = gl;
                            } else {
#pragma unroll
                                for (int a = 0; a < A; ++a) v[a] = gl * th[a];   // = -d/d a_ia
                                v[A] = gl;
                                if constexpr (IRT == 3) v[A + 1] = gguess;
                            }
#pragma unroll
                            for (int d = 0; d < DT; ++d) cur[d] = fmaf(sel, wave_total(v[d]), cur[d]);
                        }
                    }
                }
                }
            }
            if constexpr (GRAD) {
#pragma unroll
                for (int ss = 0; ss < SLOTS; ++ss)
                    if (ss == s) {
#pragma unroll
                        for (int d = 0; d < DT; ++d) acc_item[ss][d] += cur[d];
                    }
            }
        }

        // ================= epilogue =========================================
        // set 0: every wave pushes ITS partial dLL/dtheta through sample + product of experts
        if constexpr (GRAD) {
            if (valid) {
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    if (a < Ar) {
                        const PersonDim<A> d = person_dim<A>(p, ctab, cnt, grow, valid, a);
                        float g;
                        if constexpr (IRT == 1) g = gth_sum;
                        else g = (a & 1) ? gth2[a >> 1].y : gth2[a >> 1].x;
                        const float h = 0.5f * d.sig * d.eps;         // d theta / d logvar
                        const float gi = g * d.inv_lam;
                        const float nn[2] = {d.n0, d.n1};
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const float tau = ctab[(CT_TAU * 2 + c) * A + a];
                            const float te = ctab[(CT_TE * 2 + c) * A + a];
                            const float mm = ctab[(CT_M * 2 + c) * A + a];
                            const float gn = gi * nn[c];
                            acc_t0[a][c * 2 + 0] = fmaf(gn, tau, acc_t0[a][c * 2 + 0]);
                            acc_t0[a][c * 2 + 1] = fmaf(-gn * te, (mm - d.amu) - h, acc_t0[a][c * 2 + 1]);
                        }
                    }
                }
            }
        }
        // set 1 + outputs: ability dims split across waves
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int a = wave + NW * k;
            if (a < Ar && valid) {
                const PersonDim<A> d = person_dim<A>(p, ctab, cnt, grow, valid, a);
                const float alv = -kLn2 * fast_log2(d.lam);
                const float theta0 = d.amu + d.sig * d.eps;
                p.ability_mu[grow * Ar + a] = d.amu;
                p.ability_logvar[grow * Ar + a] = alv;
                p.ability[grow * Ar + a] = theta0;
                const float evar = d.inv_lam;                        // exp(logvar)
                s_kl += -0.5f * (1.0f + alv - d.amu * d.amu - evar);
                s_logq0 += -0.5f * kLog2Pi - 0.5f * alv - 0.5f * d.eps * d.eps;
                s_logp += -0.5f * kLog2Pi - 0.5f * theta0 * theta0;
                if constexpr (GRAD) {
                    float g_mu, g_lv;
                    if (p.reg_mode == 0) {        // analytic KL
                        g_mu = d.amu;
                        g_lv = -0.5f * (1.0f - evar);
                    } else {                      // log q0(theta0) - log p(theta0)
                        g_mu = theta0;
                        g_lv = theta0 * 0.5f * d.sig * d.eps - 0.5f;
                    }
                    const float nn[2] = {d.n0, d.n1};
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float tau = ctab[(CT_TAU * 2 + c) * A + a];
                        const float te = ctab[(CT_TE * 2 + c) * A + a];
                        const float mm = ctab[(CT_M * 2 + c) * A + a];
                        const float nl = nn[c] * d.inv_lam;
                        acc_t1[k][c * 2 + 0] = fmaf(g_mu * nl, tau, acc_t1[k][c * 2 + 0]);
                        const float g_tau = nl * (g_mu * (mm - d.amu) - g_lv);
                        acc_t1[k][c * 2 + 1] = fmaf(-g_tau, te, acc_t1[k][c * 2 + 1]);
                    }
                }
            }
        }
        if (wave == 0 && valid) s_nobs += (float)(cnt & 0xffffu);
        __syncthreads();   // codes of this tile are dead: next phase A may overwrite
    }

    // ================= block-level reduction -> partial record ===============
    float* out = p.partial + (size_t)blockIdx.x * p.lay.stride;
    float* red = reinterpret_cast<float*>(smem);   // codes area is free now: [NW][8 + 4*A]
    constexpr int RW = 8 + 4 * A;
    {
        float ll;
        if constexpr (IRT != 3) ll = -(s_lin + kLn2 * s_log);
        else ll = kLn2 * s_log;
        const float vals[6] = {ll, s_kl, s_logq0, s_logp, 0.f, s_nobs};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float t = wave_total(vals[k]);
            if (lane == 0) red[wave * RW + k] = t;
        }
        if constexpr (GRAD) {
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = wave_total(acc_t0[a][k]);
                    if (lane == 0) red[wave * RW + 8 + a * 4 + k] = t;
                }
        }
    }
    __syncthreads();
    if (tid < RW) {
        float t = 0.f;
        for (int w = 0; w < NW; ++w) t += red[w * RW + tid];
        if (tid < 8) {
            out[tid] = (tid < 6) ? t : 0.f;
        } else if (GRAD) {
            const int a = (tid - 8) >> 2, k = (tid - 8) & 3, c = k >> 1, ms = k & 1;
            if (a < Ar) out[p.lay.off_table + (0 * 2 + c) * 2 * Ar + ms * Ar + a] = t;
        }
    }
    if constexpr (GRAD) {
        // set-1 table grads: owned by exactly one wave per ability dim
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int a = wave + NW * k;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float t = wave_total(acc_t1[k][kk]);
                if (lane == 0 && a < Ar) {
                    const int c = kk >> 1, ms = kk & 1;
                    out[p.lay.off_table + (1 * 2 + c) * 2 * Ar + ms * Ar + a] = t;
                }
            }
        }
        // item grads: lane j of wave w, slot s owns item (w + NW*s)*64 + j
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int gb = wave + NW * s;
            if (gb < p.item_blocks) {
                const int i = gb * 64 + lane;
                float* oi = out + p.lay.off_item + i;
                if constexpr (IRT == 1) {
                    oi[0] = acc_item[s][0];
                } else {
#pragma unroll
                    for (int a = 0; a < A; ++a)
                        if (a < Ar) oi[a * p.lay.i_pad] = -acc_item[s][a];   // d/d a_ia = -sum gl*theta
                    oi[Ar * p.lay.i_pad] = acc_item[s][A];
                    if constexpr (IRT == 3) oi[(Ar + 1) * p.lay.i_pad] = acc_item[s][A + 1];
                }
            }
        }
    }
}

}  // namespace vibo
