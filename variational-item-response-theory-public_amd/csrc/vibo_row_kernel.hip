// vibo_row_kernel.hip -- fused ELBO forward+backward, one wave per person row, items in lanes.
//
// For small ability dims (A <= 2, 1PL/2PL) the cheapest mapping is the plain one: lane l of a wave owns
// items 4(l+64c)+j of EVERY row the wave processes, so
//   * the row arrives with 16-byte coalesced loads (float4 response + 4 mask bytes per lane) and stays in
//     registers for both sweeps (counts -> posterior -> theta; decode -> log-lik -> backward);
//   * item parameters and item-gradient accumulators live in the lane's registers for the whole kernel:
//     d LL/d item needs NO cross-lane reduction;
//   * theta is wave-uniform; per row only (1 + A) wave sums are needed (packed counts, d LL/d theta).
// Measured instruction costs on gfx950 (tools/ubench): v_fma 2.5 cyc, v_exp/v_log/v_rcp ~10 cyc each,
// f32 MFMA shares the VALU -- so for A <= 2 this mapping (~80 cyc per 64 terms) sits at the HBM roofline
// where the tiled MFMA kernel (vibo_elbo_kernel.hpp, built for wide A) is VALU-bound.
//
// Outputs use the same per-workgroup partial record as the tiled kernel (fixed-order, bitwise reproducible).
#include <hip/hip_runtime.h>
#include "vibo_device.hpp"
#include "vibo_launch.hpp"
#include "vibo_params.hpp"

namespace vibo {

template <int A>
struct RowPost {
    float lam[A], inv_lam[A], amu[A], sig[A], eps[A], th[A];
    float n0, n1;
};

// CH = float4 chunks (of 64 lanes) per row: I <= 256*CH
template <int A, int IRT, int CH, int MK, bool GRAD>
__global__ __launch_bounds__(256, 2) void row_kernel(const ElboParams p) {
    constexpr int NI = 4 * CH;                                 // items per lane
    constexpr float kLoS = kLogitLo * kLog2e, kHiS = kLogitHi * kLog2e;
    __shared__ float red[4][64];                               // cross-wave staging for the final reduction
    __shared__ float ctab[4 * 2 * A];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I = p.I;
    const int n4 = I >> 2;

    if (tid < 2 * A) {
        const int c = tid / A, a = tid % A;
        const float m = p.table[c * 2 * A + a], s = p.table[c * 2 * A + A + a];
        const float es = __expf(s);
        const float tau = 1.0f / (es + kPoeEps);
        ctab[(0 * 2 + c) * A + a] = tau;
        ctab[(1 * 2 + c) * A + a] = m * tau;
        ctab[(2 * 2 + c) * A + a] = tau * tau * es;
        ctab[(3 * 2 + c) * A + a] = m;
    }

    // ---- this lane's item parameters (log2 units: rows prepped by item_prep_kernel) ----
    float na[NI][A], nb[NI];
    float acc_a[NI][A], acc_b[NI];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * (lane + 64 * c) + j;
            const bool ok = i < I;
#pragma unroll
            for (int a = 0; a < A; ++a) {
                na[4 * c + j][a] = ok ? p.item_prep[(size_t)i * p.DP + a] : 0.f;
                acc_a[4 * c + j][a] = 0.f;
            }
            nb[4 * c + j] = ok ? p.item_prep[(size_t)i * p.DP + A] : 0.f;
            acc_b[4 * c + j] = 0.f;
        }
    float acc_t[A][8];     // table grads [a][set*4 + c*2 + {m,s}] (wave-uniform values, kept per lane)
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc_t[a][k] = 0.f;
    float s_log = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_nobs = 0.f;
    __syncthreads();

    auto load_row = [&](long long row, float4 (&x)[CH], uint32_t (&m)[CH], float (&ep)[A]) {
        const long long src = p.row_index ? p.row_index[row] : row;
#pragma unroll
        for (int a = 0; a < A; ++a) ep[a] = p.eps[row * A + a];
        const float4* rp = reinterpret_cast<const float4*>(p.response + src * p.resp_stride);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int q = lane + 64 * c;
            x[c] = float4{0.f, 0.f, 0.f, 0.f};
            m[c] = 0u;
            if (q < n4) {
                x[c] = rp[q];
                if constexpr (MK == 0) {
                    m[c] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride)[q];
                } else if constexpr (MK == 1) {
                    const longlong2* mp = reinterpret_cast<const longlong2*>(static_cast<const int64_t*>(p.mask) + src * p.mask_stride);
                    const longlong2 u = mp[2 * q], v = mp[2 * q + 1];
                    m[c] = (u.x != 0 ? 1u : 0u) | (u.y != 0 ? 0x100u : 0u) | (v.x != 0 ? 0x10000u : 0u) | (v.y != 0 ? 0x1000000u : 0u);
                } else {
                    m[c] = 0x01010101u;
                }
            }
        }
    };

    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long n_waves = (long long)gridDim.x * 4;
    // two row buffers: while a row is being processed, the loads of the next TWO rows are in flight
    float4 xa[CH], xb[CH];
    uint32_t ma[CH], mb[CH];
    float epa[A], epb[A];
    auto process = [&](float4 (&x)[CH], uint32_t (&m)[CH], float (&ep)[A], const long long row, const long long prow) {
        // ---- sweep 1: pack the row to one fp8 byte per cell (+1 correct / -1 wrong / 0 missing) and count
        //      observed / correct cells -> product of experts (models.py:596-629).  The raw row registers die
        //      here, so the NEXT row's HBM loads are issued into them and fly under this row's math.
        uint32_t cw[CH];
        int packed = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) cw[c] = pack_codes4(x[c], m[c], packed);
        float epsv[A];
#pragma unroll
        for (int a = 0; a < A; ++a) epsv[a] = ep[a];
        if (prow < p.B) load_row(prow, x, m, ep);
        __builtin_amdgcn_sched_barrier(0);
        const int cnt = lane63(wave_sum63(packed));
        const float n1 = (float)(cnt >> 16), nobs = (float)(cnt & 0xffff);
        const float n0 = nobs - n1, nmiss = (float)I - nobs;
        float th[A], amu[A], inv_lam[A], sig[A], lamv[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float lam = n0 * ctab[(0 * 2 + 0) * A + a] + n1 * ctab[(0 * 2 + 1) * A + a];
            if (p.missing_mode == 0) lam += nmiss * (1.0f / (1.0f + kPoeEps));
            const float s = n0 * ctab[(1 * 2 + 0) * A + a] + n1 * ctab[(1 * 2 + 1) * A + a];
            lamv[a] = lam;
            inv_lam[a] = 1.0f / lam;
            amu[a] = s * inv_lam[a];
            sig[a] = fast_rsq(lam);
            th[a] = amu[a] + sig[a] * epsv[a];
        }

        // ---- sweep 2: decode, masked Bernoulli log-lik, backward (all in registers) ----
        // The reference clamps the Bernoulli probability (utils.py:46-49 -> torch): log-lik value clamped at
        // logit +-kLogitLo, gradient exactly zero outside [-kLogitLo, kLogitHi].  |logit| > 15.9 is rare, so
        // the exact form is a wave-uniform slow path chosen per row.
        float gth[A];
#pragma unroll
        for (int a = 0; a < A; ++a) gth[a] = 0.f;
        float lg[NI];
        bool sat = false;
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            float l = nb[t];
#pragma unroll
            for (int a = 0; a < A; ++a) l = fmaf(na[t][a], th[a], l);
            lg[t] = l;
            sat |= fabsf(l) > kLoS;
        }
        const bool exact = __any(sat);
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            float w;
            if ((t & 3) == 0) w = code_to_f32<0>(cw[t >> 2]);
            else if ((t & 3) == 1) w = code_to_f32<1>(cw[t >> 2]);
            else if ((t & 3) == 2) w = code_to_f32<2>(cw[t >> 2]);
            else w = code_to_f32<3>(cw[t >> 2]);
            float lc = lg[t], wg = w;
            if (exact) {
                const float l2 = med3(lg[t], -kLoS, kHiS);
                lc = fminf(l2, kLoS);
                wg = (lg[t] == l2) ? w : 0.f;
            }
            const float eu = fast_exp2(-w * lc);
            const float tt = 1.0f + eu;
            s_log = fmaf(fabsf(w), fast_log2(tt), s_log);
            if constexpr (GRAD) {
                const float gl = wg * (eu * fast_rcp(tt));                  // d ll / d logit
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    gth[a] = fmaf(gl, na[t][a], gth[a]);                    // x log2e, removed below
                    acc_a[t][a] = fmaf(gl, th[a], acc_a[t][a]);             // = -d/d a_ia
                }
                acc_b[t] += gl;
            }
        }

        // ---- per-person epilogue (wave-uniform) ----
        float g0[A];
#pragma unroll
        for (int a = 0; a < A; ++a) g0[a] = GRAD ? wave_total(gth[a]) * kLn2 : 0.f;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const float alv = -kLn2 * fast_log2(lamv[a]);
            const float evar = inv_lam[a];
            if (lane == 0) {
                p.ability_mu[row * A + a] = amu[a];
                p.ability_logvar[row * A + a] = alv;
                p.ability[row * A + a] = th[a];
                s_kl += -0.5f * (1.0f + alv - amu[a] * amu[a] - evar);
                s_logq0 += -0.5f * kLog2Pi - 0.5f * alv - 0.5f * epsv[a] * epsv[a];
                s_logp += -0.5f * kLog2Pi - 0.5f * th[a] * th[a];
            }
            if constexpr (GRAD) {
                const float h = 0.5f * sig[a] * epsv[a];
                float gmu[2], glv[2];
                gmu[0] = g0[a];
                glv[0] = g0[a] * h;
                if (p.reg_mode == 0) {
                    gmu[1] = amu[a];
                    glv[1] = -0.5f * (1.0f - evar);
                } else {
                    gmu[1] = th[a];
                    glv[1] = th[a] * h - 0.5f;
                }
                const float nn[2] = {n0, n1};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float tau = ctab[(0 * 2 + c) * A + a], te = ctab[(2 * 2 + c) * A + a], mm = ctab[(3 * 2 + c) * A + a];
                    const float nl = nn[c] * inv_lam[a];
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        acc_t[a][st * 4 + c * 2 + 0] = fmaf(gmu[st] * nl, tau, acc_t[a][st * 4 + c * 2 + 0]);
                        const float g_tau = nl * (gmu[st] * (mm - amu[a]) - glv[st]);
                        acc_t[a][st * 4 + c * 2 + 1] = fmaf(-g_tau, te, acc_t[a][st * 4 + c * 2 + 1]);
                    }
                }
            }
        }
        if (lane == 0) s_nobs += nobs;
    };
    {
        const long long r0 = wave_id, r1 = wave_id + n_waves;
        if (r0 < p.B) load_row(r0, xa, ma, epa);
        if (r1 < p.B) load_row(r1, xb, mb, epb);
        for (long long row = r0; row < p.B; row += 2 * n_waves) {
            process(xa, ma, epa, row, row + 2 * n_waves);
            if (row + n_waves < p.B) process(xb, mb, epb, row + n_waves, row + 3 * n_waves);
        }
    }

    // ================= workgroup reduction -> partial record (same layout as the tiled kernel) ======
    float* out = p.partial + (size_t)blockIdx.x * p.lay.stride;
    {
        const float ll = -(kLn2 * wave_total(s_log));
        if (lane == 0) {
            red[wave][0] = ll; red[wave][1] = s_kl; red[wave][2] = s_logq0; red[wave][3] = s_logp;
            red[wave][4] = 0.f; red[wave][5] = s_nobs;
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int k = 0; k < 8; ++k) red[wave][8 + a * 8 + k] = acc_t[a][k];
        }
    }
    __syncthreads();
    if (tid < 8 + 8 * A) {
        const float t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (tid < 8) {
            out[tid] = (tid < 6) ? t : 0.f;
        } else if (GRAD) {
            const int a = (tid - 8) >> 3, k = (tid - 8) & 7;
            const int st = k >> 2, c = (k >> 1) & 1, ms = k & 1;
            out[p.lay.off_table + (st * 2 + c) * 2 * A + ms * A + a] = t;
        }
    }
    if constexpr (GRAD) {
        // item grads: sum the 4 waves' accumulators of the same item (same lane, same register)
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int i = 4 * (lane + 64 * (t >> 2)) + (t & 3);
#pragma unroll
            for (int d = 0; d <= A; ++d) {
                __syncthreads();
                red[wave][lane] = (d < A) ? -acc_a[t][d < A ? d : 0] : acc_b[t];
                __syncthreads();
                if (wave == 0 && i < I) {
                    const float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
                    if (IRT == 1) {
                        if (d == A) out[p.lay.off_item + i] = v;
                    } else {
                        out[p.lay.off_item + d * p.lay.i_pad + i] = v;
                    }
                }
            }
        }
    }
}

template <int A, int IRT, int CH, bool GRAD>
static hipError_t launch_row_mk(const ElboParams& p, int grid, hipStream_t s) {
    if (p.mask_dtype == 0) hipLaunchKernelGGL((row_kernel<A, IRT, CH, 0, GRAD>), dim3(grid), dim3(256), 0, s, p);
    else if (p.mask_dtype == 1) hipLaunchKernelGGL((row_kernel<A, IRT, CH, 1, GRAD>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((row_kernel<A, IRT, CH, 2, GRAD>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

template <int A, int IRT, bool GRAD>
static hipError_t launch_row_ch(const ElboParams& p, int grid, hipStream_t s) {
    if (p.I <= 256) return launch_row_mk<A, IRT, 1, GRAD>(p, grid, s);
    if (p.I <= 512) return launch_row_mk<A, IRT, 2, GRAD>(p, grid, s);
    return launch_row_mk<A, IRT, 4, GRAD>(p, grid, s);
}

// A in {1,2}, irt in {1,2}, I <= 1024, rows 16-byte aligned (vec_ok)
hipError_t launch_elbo_rows(const ElboParams& p, int irt, bool grad, int grid, hipStream_t s) {
    if (p.A == 1) {
        if (irt == 1) return grad ? launch_row_ch<1, 1, true>(p, grid, s) : launch_row_ch<1, 1, false>(p, grid, s);
        return grad ? launch_row_ch<1, 2, true>(p, grid, s) : launch_row_ch<1, 2, false>(p, grid, s);
    }
    if (irt == 1) return grad ? launch_row_ch<2, 1, true>(p, grid, s) : launch_row_ch<2, 1, false>(p, grid, s);
    return grad ? launch_row_ch<2, 2, true>(p, grid, s) : launch_row_ch<2, 2, false>(p, grid, s);
}

}  // namespace vibo
