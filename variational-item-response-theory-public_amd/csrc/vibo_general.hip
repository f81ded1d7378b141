// vibo_general.hip -- general fused ELBO kernel: one wave per person row.
//
// Covers every configuration of the hot path that the tiled MFMA kernel
// (vibo_elbo_kernel.hpp) does not specialise for: conditional posterior
// q(theta | responses, items) (models.py:664-710), planar flows on the ability
// sample (flows.py:21-66, models.py:342-348, 406-424), more than 1024 items.
// Same math, same outputs, same C ABI; simpler mapping:
//
//   lane l of a wave owns items l, l+64, ... of ONE person row;
//   pass 1  product of experts over the row (table gather per (code, item) when
//           conditional) -> wave reduce -> posterior mean / log-variance, sample,
//           planar flows (all lanes redundantly, per-person scalars);
//   pass 2  logit -> masked Bernoulli log-lik -> g; item gradients accumulate in an
//           LDS slab shared by the workgroup's waves (ds_add_f32), dLL/dtheta per lane;
//   pass 3  wave reduce dLL/dtheta -> flows backward -> backward through sample +
//           product of experts; conditional posterior: second sweep over the row
//           (L1/L2 resident) scattering table gradients.
//   Gradient buffers are zeroed by the launcher and accumulated with fp32 atomics,
//   so this path is NOT bitwise reproducible (the tiled kernel is).
#include <hip/hip_runtime.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_general.hpp"

namespace vibo {

constexpr int MF = VIBO_MAX_FLOWS;

__device__ __forceinline__ bool observed(const GeneralParams& p, long long src, int i) {
    if (p.mask_dtype == VIBO_MASK_U8) return static_cast<const uint8_t*>(p.mask)[src * p.mask_stride + i] != 0;
    if (p.mask_dtype == VIBO_MASK_I64) return static_cast<const int64_t*>(p.mask)[src * p.mask_stride + i] != 0;
    return true;
}

// MA: the widest ability_dim the instantiation holds in registers -- 8 (every fast path's limit) or 16 (VIBO_MAX_ABILITY_DIM_WIDE:
// ability_dim 9..16 runs here and only here)
template <int MA>
__global__ __launch_bounds__(256) void elbo_general_kernel(const GeneralParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    float* lds_item = reinterpret_cast<float*>(smem_g);       // [I][D] when p.item_in_lds
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int I = p.I, A = p.A, D = p.D;
    const bool grad = p.want_grad != 0;
    const float tau_prior = 1.0f / (1.0f + kPoeEps);

    if (grad && p.item_in_lds) {
        for (int k = threadIdx.x; k < I * D; k += blockDim.x) lds_item[k] = 0.f;
        __syncthreads();
    }

    // per-lane accumulators over all rows this wave processes
    float s_ll = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_ladj = 0.f, s_nobs = 0.f;
    float tacc[2][2][2 * MA];      // unconditional table grads [set][c][m|s]  (lane 0 only)
    float facc[2][MF][2 * MA + 1]; // flow grads [set][flow][uhat|w|b]         (lane 0 only)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2 * MA; ++j) tacc[s][c][j] = 0.f;
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < 2 * MA + 1; ++j) facc[s][f][j] = 0.f;
    }

    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long n_waves = (long long)gridDim.x * 4;
    for (long long row = wave_id; row < p.B; row += n_waves) {
        const long long src = p.row_index ? p.row_index[row] : row;
        const float* rp = p.response + src * p.resp_stride;

        // ---------------- pass 1: product of experts ----------------
        float lam[MA], smu[MA];
#pragma unroll
        for (int a = 0; a < MA; ++a) lam[a] = smu[a] = 0.f;
        float nobs = 0.f;
        for (int i = lane; i < I; i += 64) {
            const bool k = observed(p, src, i);
            const int c = (rp[i] == 1.0f) ? 1 : 0;
            const float* te = p.conditional ? p.table + ((size_t)c * I + i) * 2 * A : p.table + (size_t)c * 2 * A;
            if (k) nobs += 1.f;
#pragma unroll
            for (int a = 0; a < MA; ++a) {
                if (a < A) {
                    if (k) {
                        const float tau = 1.0f / (expf(te[A + a]) + kPoeEps);
                        lam[a] += tau;
                        smu[a] = fmaf(te[a], tau, smu[a]);
                    } else if (p.missing_mode == VIBO_MISSING_PRIOR) {
                        lam[a] += tau_prior;
                    }
                }
            }
        }
        float amu[MA], alv[MA], sig[MA], epsv[MA], th0[MA], th[MA], ilam[MA];
#pragma unroll
        for (int a = 0; a < MA; ++a) {
            amu[a] = alv[a] = sig[a] = epsv[a] = th0[a] = th[a] = ilam[a] = 0.f;
            if (a < A) {
                const float L = wave_total(lam[a]);
                const float S = wave_total(smu[a]);
                ilam[a] = 1.0f / L;
                amu[a] = S * ilam[a];
                alv[a] = logf(ilam[a]);
                sig[a] = sqrtf(ilam[a]);
                epsv[a] = p.eps[row * A + a];
                th0[a] = amu[a] + sig[a] * epsv[a];
                th[a] = th0[a];
            }
        }
        {
            const float tn = wave_total(nobs);
            if (lane == 0) s_nobs += tn;
        }

        // ---------------- planar flows (per person; every lane computes the same) ----------------
        float ft[MF], fpsi[MF];          // tanh(w.z+b), 1 + (1-t^2) w.uhat   per flow
        float zin[MF][MA];               // input of each flow
        float ladj = 0.f;
        for (int f = 0; f < p.n_flows; ++f) {
            const float* fp = p.flow + (size_t)f * (2 * A + 1);
            float aa = fp[2 * A], cwu = 0.f;
#pragma unroll
            for (int a = 0; a < MA; ++a)
                if (a < A) {
                    zin[f][a] = th[a];
                    aa = fmaf(th[a], fp[A + a], aa);
                    cwu = fmaf(fp[A + a], fp[a], cwu);
                }
            const float t = tanhf(aa);
            ft[f] = t;
            fpsi[f] = 1.0f + (1.0f - t * t) * cwu;
            ladj += logf(fabsf(fpsi[f]) + 1e-8f);
#pragma unroll
            for (int a = 0; a < MA; ++a)
                if (a < A) th[a] = fmaf(fp[a], t, th[a]);
        }

        // ---------------- pass 2: decode + log-lik + d/d logit ----------------
        float gth[MA];
#pragma unroll
        for (int a = 0; a < MA; ++a) gth[a] = 0.f;
        float ll = 0.f;
        for (int i = lane; i < I; i += 64) {
            const bool k = observed(p, src, i);
            if (!k) continue;
            const float x = (rp[i] == 1.0f) ? 1.f : 0.f;
            const float* it = p.item + (size_t)i * D;
            float l;
            if (p.irt == 1) {
                l = it[0];
#pragma unroll
                for (int a = 0; a < MA; ++a)
                    if (a < A) l += th[a];
            } else {
                l = it[A];
#pragma unroll
                for (int a = 0; a < MA; ++a)
                    if (a < A) l = fmaf(-it[a], th[a], l);
            }
            float gl, gguess = 0.f;
            if (p.irt != 3) {
                const float lc = fminf(fmaxf(l, -kLogitLo), kLogitLo);
                const bool live = (l >= -kLogitLo) && (l <= kLogitHi);
                const float e = expf(-fabsf(lc));
                ll += x * lc - fmaxf(lc, 0.f) - log1pf(e);
                const float r = 1.0f / (1.0f + e);
                const float sgm = (lc >= 0.f) ? r : e * r;
                gl = live ? (x - sgm) : 0.f;
            } else {
                const float guess = 1.0f / (1.0f + expf(-it[A + 1]));
                const float e = expf(-fabsf(l));
                const float r = 1.0f / (1.0f + e);
                const float sp = (l >= 0.f) ? r : e * r, sn = (l >= 0.f) ? e * r : r;
                const float pr = fmaf(1.0f - guess, sp, guess);
                const float qr = (1.0f - guess) * sn;
                const float pc = fminf(fmaxf(pr, kEps32), 1.0f - kEps32);
                const float qc = fminf(fmaxf(qr, kEps32), 1.0f - kEps32);
                ll += (x > 0.5f) ? logf(pc) : logf(qc);
                const float dll_dp = (pr == pc) ? ((x > 0.5f) ? 1.0f / pc : -1.0f / qc) : 0.f;
                const float common = dll_dp * (1.0f - guess) * sn;
                gl = common * sp;
                gguess = common * guess;
            }
            if (grad) {
                float* gi = p.item_in_lds ? lds_item + (size_t)i * D : p.grad_item + (size_t)i * D;
                if (p.irt == 1) {
                    atomicAdd(gi, gl);
#pragma unroll
                    for (int a = 0; a < MA; ++a)
                        if (a < A) gth[a] += gl;
                } else {
#pragma unroll
                    for (int a = 0; a < MA; ++a)
                        if (a < A) {
                            atomicAdd(gi + a, -gl * th[a]);
                            gth[a] = fmaf(gl, -it[a], gth[a]);
                        }
                    atomicAdd(gi + A, gl);
                    if (p.irt == 3) atomicAdd(gi + A + 1, gguess);
                }
            }
        }
        s_ll += ll;

        // ---------------- per-person heads ----------------
        float kl = 0.f, logq0 = 0.f, logp = 0.f;
#pragma unroll
        for (int a = 0; a < MA; ++a)
            if (a < A) {
                kl += -0.5f * (1.0f + alv[a] - amu[a] * amu[a] - ilam[a]);
                logq0 += -0.5f * kLog2Pi - 0.5f * alv[a] - 0.5f * epsv[a] * epsv[a];
                logp += -0.5f * kLog2Pi - 0.5f * th[a] * th[a];
            }
        if (lane == 0) {
            s_kl += kl; s_logq0 += logq0; s_logp += logp; s_ladj += ladj;
#pragma unroll
            for (int a = 0; a < MA; ++a)
                if (a < A) {
                    p.ability_mu[row * A + a] = amu[a];
                    p.ability_logvar[row * A + a] = alv[a];
                    p.ability[row * A + a] = th0[a];
                    if (p.ability_k) p.ability_k[row * A + a] = th[a];
                }
            if (p.ability_ladj) p.ability_ladj[row] = ladj;
        }
        if (!grad) continue;

        // ---------------- pass 3: backward ----------------
        float gz[2][MA];       // d head / d theta_K : set 0 = LL, set 1 = REG
#pragma unroll
        for (int a = 0; a < MA; ++a) {
            gz[0][a] = (a < A) ? wave_total(gth[a]) : 0.f;
            gz[1][a] = (a < A && p.reg_mode == VIBO_REG_SAMPLED) ? th[a] : 0.f;     // d(-log p(theta_K))
        }
        for (int f = p.n_flows - 1; f >= 0; --f) {
            const float* fp = p.flow + (size_t)f * (2 * A + 1);
            const float t = ft[f], omt = 1.0f - t * t;
            float cwu = 0.f;
#pragma unroll
            for (int a = 0; a < MA; ++a)
                if (a < A) cwu = fmaf(fp[A + a], fp[a], cwu);
            const float dl_dpsi_unit = ((fpsi[f] >= 0.f) ? 1.0f : -1.0f) / (fabsf(fpsi[f]) + 1e-8f);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float gla = (s == 1) ? -1.0f : 0.f;          // d REG / d ladj = -1
                const float dl_dpsi = gla * dl_dpsi_unit;
                float g_t = dl_dpsi * (-2.0f * t * cwu);
#pragma unroll
                for (int a = 0; a < MA; ++a)
                    if (a < A) g_t = fmaf(gz[s][a], fp[a], g_t);
                const float g_c = dl_dpsi * omt;
                const float g_a = g_t * omt;
                if (lane == 0) {
#pragma unroll
                    for (int a = 0; a < MA; ++a)
                        if (a < A) {
                            facc[s][f][a] += gz[s][a] * t + g_c * fp[A + a];            // d / d uhat
                            facc[s][f][A + a] += g_a * zin[f][a] + g_c * fp[a];         // d / d w
                        }
                    facc[s][f][2 * A] += g_a;                                           // d / d b
                }
#pragma unroll
                for (int a = 0; a < MA; ++a)
                    if (a < A) gz[s][a] = fmaf(g_a, fp[A + a], gz[s][a]);
            }
        }
        float gmu[2][MA], glv[2][MA];
#pragma unroll
        for (int a = 0; a < MA; ++a) {
            const float h = 0.5f * sig[a] * epsv[a];
            gmu[0][a] = gz[0][a];
            glv[0][a] = gz[0][a] * h;
            gmu[1][a] = gz[1][a];
            glv[1][a] = gz[1][a] * h;
            if (a < A) {
                if (p.reg_mode == VIBO_REG_KL) {
                    gmu[1][a] += amu[a];
                    glv[1][a] += -0.5f * (1.0f - ilam[a]);
                } else {
                    glv[1][a] += -0.5f;
                }
            }
        }
        // conditional posterior: sweep the row again, scatter into the [2][I][2A] table gradient
        if (p.conditional) {
            for (int i = lane; i < I; i += 64) {
                if (!observed(p, src, i)) continue;
                const int c = (rp[i] == 1.0f) ? 1 : 0;
                const float* te = p.table + ((size_t)c * I + i) * 2 * A;
#pragma unroll
                for (int a = 0; a < MA; ++a)
                    if (a < A) {
                        const float es = expf(te[A + a]);
                        const float tau = 1.0f / (es + kPoeEps);
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const float g_m = gmu[s][a] * ilam[a] * tau;
                            const float g_tau = (gmu[s][a] * (te[a] - amu[a]) - glv[s][a]) * ilam[a];
                            float* gt = p.grad_table + (size_t)s * 2 * I * 2 * A + ((size_t)c * I + i) * 2 * A;
                            atomicAdd(gt + a, g_m);
                            atomicAdd(gt + A + a, -g_tau * tau * tau * es);
                        }
                    }
            }
        }
        if (!p.conditional) {
            // unconditional: gradient depends on the row only through the counts of each code
            float n1 = 0.f, n0 = 0.f;
            for (int i = lane; i < I; i += 64)
                if (observed(p, src, i)) {
                    if (rp[i] == 1.0f) n1 += 1.f; else n0 += 1.f;
                }
            n1 = wave_total(n1);
            n0 = wave_total(n0);
            if (lane == 0) {
                const float nn[2] = {n0, n1};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float* te = p.table + (size_t)c * 2 * A;
#pragma unroll
                    for (int a = 0; a < MA; ++a)
                        if (a < A) {
                            const float es = expf(te[A + a]);
                            const float tau = 1.0f / (es + kPoeEps);
#pragma unroll
                            for (int s = 0; s < 2; ++s) {
                                const float g_m = gmu[s][a] * ilam[a] * tau * nn[c];
                                const float g_tau = (gmu[s][a] * (te[a] - amu[a]) - glv[s][a]) * ilam[a] * nn[c];
                                tacc[s][c][a] += g_m;
                                tacc[s][c][A + a] += -g_tau * tau * tau * es;
                            }
                        }
                }
            }
        }
    }

    // ---------------- flush ----------------
    const float tll = wave_total(s_ll);
    if (lane == 0) {
        atomicAdd(p.acc_scalars + 0, tll);
        atomicAdd(p.acc_scalars + 1, s_kl);
        atomicAdd(p.acc_scalars + 2, s_logq0);
        atomicAdd(p.acc_scalars + 3, s_logp);
        atomicAdd(p.acc_scalars + 4, s_ladj);
        atomicAdd(p.acc_scalars + 5, s_nobs);
        if (grad) {
            if (!p.conditional) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < 2 * MA; ++j)
                            if (j < 2 * A) atomicAdd(p.grad_table + ((size_t)s * 2 + c) * 2 * A + j, tacc[s][c][j]);
            }
            for (int f = 0; f < p.n_flows; ++f)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 2 * MA + 1; ++j)
                        if (j < 2 * A + 1)
                            atomicAdd(p.grad_flow + ((size_t)s * p.n_flows + f) * (2 * A + 1) + j, facc[s][f][j]);
        }
    }
    if (grad && p.item_in_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < I * D; k += blockDim.x) {
            const float v = lds_item[k];
            if (v != 0.f) atomicAdd(p.grad_item + k, v);
        }
    }
}

// scalars: acc[0..5] = ll, kl, logq0, logp, ladj, nobs  ->  out_scalars (VIBO_S_* order)
__global__ void general_scalars_kernel(const float* acc, float* out, int reg_mode) {
    if (threadIdx.x == 0) {
        const float ll = acc[0], kl = acc[1], logq0 = acc[2], logp = acc[3], ladj = acc[4];
        out[VIBO_S_LL] = ll;
        out[VIBO_S_REG] = (reg_mode == VIBO_REG_KL) ? kl : (logq0 - ladj - logp);
        out[VIBO_S_KL] = kl;
        out[VIBO_S_LOGQ0] = logq0;
        out[VIBO_S_LOGP] = logp;
        out[VIBO_S_LADJ] = ladj;
        out[VIBO_S_NOBS] = acc[5];
        out[VIBO_S_RESERVED] = 0.f;
    }
}

hipError_t launch_elbo_general(const GeneralParams& p, int num_cu, hipStream_t s) {
    const size_t item_bytes = (size_t)p.I * p.D * sizeof(float);
    GeneralParams q = p;
    q.item_in_lds = (p.want_grad && item_bytes <= 150 * 1024) ? 1 : 0;
    const size_t lds = q.item_in_lds ? item_bytes : 0;
    const bool wide = p.A > VIBO_MAX_ABILITY_DIM;
    static bool attr_set[2] = {false, false};
    if (lds > 48 * 1024 && !attr_set[wide]) {
        hipError_t e = hipFuncSetAttribute(wide ? reinterpret_cast<const void*>(elbo_general_kernel<VIBO_MAX_ABILITY_DIM_WIDE>)
                                                : reinterpret_cast<const void*>(elbo_general_kernel<VIBO_MAX_ABILITY_DIM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[wide] = true;
    }
    long long blocks = (p.B + 3) / 4;
    const long long cap = (long long)num_cu * (lds > 80 * 1024 ? 1 : lds > 40 * 1024 ? 2 : 4);
    if (blocks > cap) blocks = cap;
    if (wide) hipLaunchKernelGGL(elbo_general_kernel<VIBO_MAX_ABILITY_DIM_WIDE>, dim3((unsigned)blocks), dim3(256), lds, s, q);
    else hipLaunchKernelGGL(elbo_general_kernel<VIBO_MAX_ABILITY_DIM>, dim3((unsigned)blocks), dim3(256), lds, s, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(general_scalars_kernel, dim3(1), dim3(64), 0, s, p.acc_scalars, p.out_scalars, p.reg_mode);
    return hipGetLastError();
}

}  // namespace vibo
