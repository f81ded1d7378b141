// vibo_split_kernel.hpp -- fused ELBO forward+backward, items in lanes, a response row shared by the waves
// of a workgroup ("row-split" kernel).  Unconditional posterior, 1PL/2PL/3PL, optional planar flows,
// I <= 1024 (I % 4 == 0), any ability_dim <= 8.
//
// Measured on gfx950 (tools/ubench*): a wave issues at most one VALU instruction per ~4.5 cycles, plain fp32
// ops retire in ~2 cycles and v_pk_fma_f32 in ~4 (no FLOP gain, but two FMAs per issue slot);
// v_mfma_f32_16x16x4_f32 retires 1024 MACs in ~32 cycles = the VALU FMA rate and does not overlap with VALU
// work; v_readlane / v_permlane*_swap cost ~3 plain ops.  So the contractions stay on the VALU as packed
// FMAs, an item never leaves its lane, and the register problem of wide ability vectors is solved by
// splitting every response row over the nq <= 4 waves of a workgroup:
//   * wave q, lane l owns items 256q + 4l + {0..3} of EVERY row: item parameters and item-gradient
//     accumulators stay in registers for the whole kernel; d LL / d item needs no reduction at all;
//   * rows are walked in batches of 8.  Per batch and wave: 8 x (16 B response + 4 B mask) per lane arrive as
//     coalesced loads issued one batch ahead, are packed to 8 fp8 code words (LDS), and the 8 packed counts
//     are summed over the wave by ONE 8-value butterfly (v_permlane32_swap / v_permlane16_swap / DPP) that
//     leaves row r's total in lanes 8r..8r+7;
//   * after one workgroup barrier lane (r, d) of every wave forms the product of experts, the sample
//     theta[r][d] and the planar flows (wave-redundant, 64 lanes wide, dot products over d by DPP); theta
//     reaches the decode through a wave-private LDS row read back as broadcasts;
//   * decode / log-lik / backward per row are packed FMAs + 3 transcendentals per term (one log2 per 4 terms);
//     d LL / d theta partials are summed over the 64 lanes through an LDS transpose one row group behind the
//     math; a second barrier hands the nq partials to wave 0, whose lane (r, d) backpropagates through the
//     flows, the sample and the PoE into the table / flow gradient accumulators.
// Outputs use the same per-workgroup partial record as the other kernels (fixed order, bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_params.hpp"

namespace vibo {

typedef unsigned uint2v __attribute__((ext_vector_type(2)));
constexpr int kSplitRows = 8;             // rows per batch
constexpr int kMF = VIBO_MAX_FLOWS;

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
__device__ __forceinline__ int bfly8(const int (&v)[8], const int lane) {
    int w[4], u[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned a = (unsigned)v[k], b = (unsigned)v[k + 4];
        swap32(a, b);     // a = [v_k lanes 0-31 | v_k+4 lanes 0-31], b = [v_k lanes 32-63 | v_k+4 lanes 32-63]
        w[k] = (int)(a + b);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned a = (unsigned)w[k], b = (unsigned)w[k + 2];
        swap16(a, b);     // odd 16-lane rows of a <-> even rows of b
        u[k] = (int)(a + b);
    }
    const bool hi = (lane & 8) != 0;
    const int keep = hi ? u[1] : u[0], give = hi ? u[0] : u[1];
    int t = keep + dpp_i<0x128>(give);       // row_ror 8
    t += dpp_i<0x141>(t);                    // row_half_mirror
    t += dpp_i<0xb1>(t);                     // quad_perm [1,0,3,2]
    t += dpp_i<0x4e>(t);                     // quad_perm [2,3,0,1]
    return t;
}

// sum over the AT consecutive lanes that hold one person's ability dims (every lane of the group gets it)
template <int AT>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_f<0xb1>(v);                           // quad_perm [1,0,3,2]
    if constexpr (AT >= 4) v += dpp_f<0x4e>(v);    // quad_perm [2,3,0,1]
    if constexpr (AT >= 8) v += dpp_f<0x141>(v);   // row_half_mirror
    return v;
}

struct alignas(16) SplitWaveLds {
    float gtl[2][8][68];              // d LL/d theta partials of a row group, transposed read-back
    uint32_t codes[kSplitRows][64];   // this batch's fp8 code words
    float thl[64];                    // theta[r][d] of the batch
    float gthp[64];                   // 64-lane sums of d LL/d theta, slot r*AT+d
    int cntp[8];                      // packed counts per row (n1 << 16 | nobs) of this wave's items
    float red[8];
};
struct alignas(16) SplitCommonLds {
    float tred[8][64];
    float ctab[4 * 2 * 8];
    float fpar[kMF][2][8];            // planar flows: uhat | w per ability dim
    float fsc[kMF][2];                // b, w.uhat
};
struct alignas(16) SplitFlowAcc {
    float a[2][kMF][3][64];           // [set][flow][uhat | w | b][lane (r, d)]  (wave 0)
    float st[kMF][3][64];             // forward state of the batch for the backward: tanh, psi, flow input (wave 0)
};
inline size_t split_lds_bytes(int nq, bool flows) {
    return sizeof(SplitCommonLds) + (flows ? sizeof(SplitFlowAcc) : 0) + (size_t)nq * sizeof(SplitWaveLds);
}

// AT = template ability width (2, 4 or 8; runtime p.A <= AT); blockDim.x = 64 nq, nq = ceil(I / 256).
// NQT = nq as a compile-time constant with static LDS (4: the 768 < I <= 1024 shapes the benchmarks use), or 0 for
// a runtime nq with dynamic LDS.
// RM (row mode): 0 = fp32 responses + mask bytes, rows in order; 1 = the same through p.row_index (shuffled minibatch);
// 2 = 1-byte cell codes (VIBO_MASK_CODES, read through p.mask), with or without p.row_index.  Without the 40 row
// registers of the fp32 layout the width-2 kernels fit 3 waves per SIMD (a few spills; 1.11 -> 0.95 ms on 1M x 1k 2PL,
// 1.73 -> 1.38 ms 3PL); the fp32 layout at 3 waves spills ~100 registers (2.7 ms).
template <int AT, int IRT, bool GRAD, bool FLOWS, int NQT, int RM = 0>
__global__ __launch_bounds__(256, (RM == 2 && (AT <= 2 || (AT == 4 && IRT <= 2))) ? 3 : 2) void split_kernel(const ElboParams p) {
    constexpr bool CODES = RM == 2, GATHER = RM == 1;
    constexpr int R = kSplitRows;
    // rows per d LL/d theta reduction group: 8 / AT fills the 8-value reduction; 3PL at width 2 takes half of that
    // (its longer per-term math would otherwise keep 16 terms' temporaries live and spill)
    constexpr int RPB = (IRT == 3 && AT == 2) ? 2 : 8 / AT;
    constexpr int NVG = RPB * AT;                  // partial sums per group (<= 8)
    constexpr int G = R / RPB;
    constexpr int NE = R * AT;                     // (row, dim) lanes used by the per-person math
    constexpr int H = AT / 2;
    constexpr float kLoS = kLogitLo * kLog2e, kHiS = kLogitHi * kLog2e;
    SplitCommonLds* clp;
    SplitFlowAcc* fap = nullptr;
    SplitWaveLds* wls;
    if constexpr (NQT > 0) {
        __shared__ SplitCommonLds cl_static;
        __shared__ SplitWaveLds wls_static[NQT > 0 ? NQT : 1];
        clp = &cl_static;
        wls = wls_static;
        if constexpr (FLOWS) {
            __shared__ SplitFlowAcc fa_static;
            fap = &fa_static;
        }
    } else {
        extern __shared__ __attribute__((aligned(16))) unsigned char split_smem[];
        clp = reinterpret_cast<SplitCommonLds*>(split_smem);
        fap = reinterpret_cast<SplitFlowAcc*>(split_smem + sizeof(SplitCommonLds));
        wls = reinterpret_cast<SplitWaveLds*>(split_smem + sizeof(SplitCommonLds) + (FLOWS ? sizeof(SplitFlowAcc) : 0));
    }
    SplitCommonLds& cl = *clp;
    SplitFlowAcc& fa = *fap;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = NQT > 0 ? NQT : (int)(blockDim.x >> 6);
    SplitWaveLds& wl = wls[q];
    const int I = p.I, A = p.A;
    const int n4 = (I + 3) >> 2;                          // I % 4 != 0: rows are padded to 16 B by the caller's strides
    const int chunk = q * 64 + lane;                      // float4 chunk of the row this lane owns
    const bool chunk_ok = chunk < n4;
    // cells of the last chunk that lie past the row's end (they belong to the padding / the next columns)
    const uint32_t tail_mask = ((I & 3) && chunk == (I >> 2)) ? ((1u << (8 * (I & 3))) - 1u) : 0xFFFFFFFFu;

    auto put_ctab = [&](const float* table) {
        if (tid < 2 * AT) {
            const int c = tid / AT, a = tid % AT;
            float m = 0.f, s = 0.f;
            if (a < A) { m = table[c * 2 * A + a]; s = table[c * 2 * A + A + a]; }
            const float es = __expf(s);
            const float tau = 1.0f / (es + kPoeEps);
            cl.ctab[(0 * 2 + c) * AT + a] = tau;
            cl.ctab[(1 * 2 + c) * AT + a] = m * tau;
            cl.ctab[(2 * 2 + c) * AT + a] = tau * tau * es;
            cl.ctab[(3 * 2 + c) * AT + a] = m;
        }
    };
    put_ctab(p.table);
    if (p.step_tick && blockIdx.x == 0 && tid == 0) *p.step_tick += 1;
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
        if constexpr (GRAD) {
            for (int k = tid; k < 2 * kMF * 3 * 64; k += blockDim.x) (&fa.a[0][0][0][0])[k] = 0.f;
        }
    }

    // ---- this lane's 4 items, read from the caller's [I][D] item sample and brought to the kernel's form here (log2 units:
    //      na = -a_ia log2 e, or +log2 e for 1PL; nb = b_i log2 e; guess = sigmoid: what item_prep_kernel does for the fallback
    //      kernels, without its launch) ----
    float2v na2[4][H];
    float nb[4];
    float2v acc_a2[4][H];
    float acc_b[4];
    float gs[4], om[4], acc_g[4];       // 3PL: guess, 1 - guess, d / d guess-logit
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int il = 4 * chunk + j;                          // item of the panel
        const bool ok = chunk_ok && il < I;                    // (the last chunk of a padded row ends past the items)
        const float* ir = p.item_raw + (size_t)(p.item0 + (ok ? il : 0)) * p.D;
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            float v = 0.f;
            if (ok && a < A) v = IRT == 1 ? kLog2e : -ir[a] * kLog2e;          // models.py:731 / 744,759
            na2[j][a >> 1][a & 1] = v;
            acc_a2[j][a >> 1][a & 1] = 0.f;
        }
        nb[j] = ok ? ir[IRT == 1 ? 0 : A] * kLog2e : 0.f;
        acc_b[j] = 0.f;
        acc_g[j] = 0.f;
        float gv = 0.f;
        if (IRT == 3 && ok) gv = 1.0f / (1.0f + expf(-ir[A + 1]));            // models.py:758
        gs[j] = gv;
        om[j] = (IRT == 3 && ok) ? 1.0f - gv : (IRT == 3 && chunk_ok ? 0.f : 1.f);
    }
    // lane e = (er, ed): person er of the batch, ability dim ed
    const int er = (lane / AT) & (R - 1), ed = lane % AT;
    const bool e_ok = lane < NE && ed < A;
    float acc_t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc_t[k] = 0.f;
    float s_log = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_nobs = 0.f, s_ladj = 0.f;
    int unobs = 0;                        // unobserved cells of this lane so far (1PL/2PL log-lik correction)
    __syncthreads();
    const float tau0 = cl.ctab[(0 * 2 + 0) * AT + ed], tau1 = cl.ctab[(0 * 2 + 1) * AT + ed];
    const float mt0 = cl.ctab[(1 * 2 + 0) * AT + ed], mt1 = cl.ctab[(1 * 2 + 1) * AT + ed];

    const long long n_batches = ((long long)p.B + R - 1) / R;
    float4 x[CODES ? 1 : R];
    uint32_t m[R];
    float epn = 0.f;
    auto load_batch = [&](const long long bt) {
        const long long row0 = bt * R;
        if constexpr (CODES) {
            // gathered rows: all 8 (wave-uniform, < 2^31) row indices first, parked in scalar registers, then the row
            // loads back to back (an index load in front of every row load makes each row wait for the previous one)
            int src[R];
#pragma unroll
            for (int r = 0; r < R; ++r) src[r] = (int)row0 + r;
            if (p.row_index) {
#pragma unroll
                for (int r = 0; r < R; ++r) src[r] = row0 + r < p.B ? (int)p.row_index[row0 + r] : 0;
#pragma unroll
                for (int r = 0; r < R; ++r) src[r] = __builtin_amdgcn_readfirstlane(src[r]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                m[r] = kAllMissing4;
                if (row0 + r < p.B && chunk_ok)
                    m[r] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + (long long)src[r] * p.mask_stride + p.item0)[chunk];
            }
        } else if constexpr (GATHER) {
            // the 8 row indices (wave-uniform, < 2^31) first, parked in scalar registers
            int src[R];
#pragma unroll
            for (int r = 0; r < R; ++r) src[r] = row0 + r < p.B ? (int)p.row_index[row0 + r] : 0;
#pragma unroll
            for (int r = 0; r < R; ++r) src[r] = __builtin_amdgcn_readfirstlane(src[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                x[r] = float4{0.f, 0.f, 0.f, 0.f};
                m[r] = 0u;
                if (row0 + r < p.B && chunk_ok) {
                    x[r] = reinterpret_cast<const float4*>(p.response + (long long)src[r] * p.resp_stride + p.item0)[chunk];
                    if (p.mask_dtype == 0)
                        m[r] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + (long long)src[r] * p.mask_stride + p.item0)[chunk];
                    else
                        m[r] = 0x01010101u;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long long row = row0 + r;
                x[r] = float4{0.f, 0.f, 0.f, 0.f};
                m[r] = 0u;
                if (row < p.B && chunk_ok) {
                    const long long src = row;          // (rows through p.row_index take the RM = 1 instantiation)
                    x[r] = reinterpret_cast<const float4*>(p.response + src * p.resp_stride + p.item0)[chunk];
                    if (p.mask_dtype == 0)
                        m[r] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride + p.item0)[chunk];
                    else
                        m[r] = 0x01010101u;
                }
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
            if constexpr (CODES) wl.codes[r][lane] = pack_cell_codes4(m[r], tail_mask, pk[r]);
            else wl.codes[r][lane] = pack_codes4(x[r], m[r] & tail_mask, pk[r]);   // mask at use: the loads stay in flight
        }
        const float eps_c = epn;
        if (bt + gridDim.x < n_batches) load_batch(bt + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int tot = bfly8(pk, lane);
            if ((lane & 7) == 0) wl.cntp[lane >> 3] = tot;
            // 1PL/2PL: every cell of this lane without an observation (missing, padding, rows past the end) contributes
            // exactly log2(1 + 2^0) = 1 to the lane's running log-lik sum below: count them (integers, exact) and
            // take them out of the lane's own sum at the end, so that the sums only ever hold real contributions
            if constexpr (IRT != 3) {
                int obs8 = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) obs8 += pk[r];
                unobs += 4 * R - (obs8 & 0xffff);
            }
        }
        __syncthreads();

        // ---- product of experts + reparameterised sample for (person er, dim ed)  (models.py:596-629) ----
        const bool live = e_ok && (row0 + er) < p.B;
        int cnt = 0;
        if (p.row_cnt) {                    // panel mode: counts of the whole row from row_count_kernel
            cnt = live ? p.row_cnt[row0 + er] : 0;
        } else {
#pragma unroll
            for (int w = 0; w < (NQT > 0 ? NQT : 4); ++w)
                if (w < nq) cnt += wls[w].cntp[er];
        }
        const float n1 = (float)(cnt >> 16);
        float nobs = (float)(cnt & 0xffff);
        const float n0 = nobs - n1;
        float lam = n0 * tau0 + n1 * tau1, smu = n0 * mt0 + n1 * mt1;
        if (p.pre_stats) {                  // conditional posterior: per-(code, item) experts summed by cond_pre_kernel
            lam = 0.f; smu = 0.f; nobs = 0.f;
            if (live) {
                for (int pn = 0; pn < p.pre_panels; ++pn) {
                    const float* st = p.pre_stats + ((size_t)pn * p.B + (row0 + er)) * (2 * A + 1);
                    lam += st[ed]; smu += st[A + ed]; nobs += st[2 * A];
                }
            }
        }
        if (p.given_post) {                 // caller-supplied posterior (given_pre_kernel's statements)
            lam = 0.f; smu = 0.f; nobs = (float)p.I_total;
            if (live) {
                const float* po = p.given_post + (size_t)(row0 + er) * 2 * A;
                lam = expf(-po[A + ed]);
                smu = po[ed] * lam;
            }
        }
        const float nmiss = (float)p.I_total - nobs;
        if (p.missing_mode == 0) lam += nmiss * (1.0f / (1.0f + kPoeEps));
        if (!live) lam = 1.0f;              // rows past the end / padded dims: keep the arithmetic finite
        const float inv_lam = 1.0f / lam;
        const float amu = smu * inv_lam;
        const float sig = fast_rsq(lam);
        const float th0 = live ? amu + sig * eps_c : 0.f;
        float thv = th0;
        // planar flows on the sample (flows.py:21-66, models.py:342-348): z <- z + uhat tanh(w.z + b)
        float ladj = 0.f;
        if constexpr (FLOWS) {
#pragma unroll
            for (int f = 0; f < kMF; ++f) {
                if (f < p.n_flows) {
                    const float ud = cl.fpar[f][0][ed], wd = cl.fpar[f][1][ed];
                    const float aa = group_sum<AT>(thv * wd) + cl.fsc[f][0];
                    // tanh(x) = 1 - 2 / (1 + e^2x), |x| clamped so that e^2x stays finite (tanh(+-15) = +-1 in fp32)
                    const float t = 1.0f - 2.0f * fast_rcp(1.0f + fast_exp2((2.0f * kLog2e) * med3(aa, -15.f, 15.f)));
                    const float psi = 1.0f + (1.0f - t * t) * cl.fsc[f][1];
                    if (GRAD && q == 0) {
                        fa.st[f][0][lane] = t;
                        fa.st[f][1][lane] = psi;
                        fa.st[f][2][lane] = thv;
                    }
                    ladj += kLn2 * fast_log2(fabsf(psi) + 1e-8f);
                    thv = fmaf(ud, t, thv);
                }
            }
        }
        if (q == 0 && live && p.primary) {
            const long long o = (row0 + er) * A + ed;
            const float alv = -kLn2 * fast_log2(lam);
            p.ability_mu[o] = amu;
            p.ability_logvar[o] = alv;
            p.ability[o] = th0;
            s_kl += -0.5f * (1.0f + alv - amu * amu - inv_lam);
            s_logq0 += -0.5f * kLog2Pi - 0.5f * alv - 0.5f * eps_c * eps_c;
            s_logp += -0.5f * kLog2Pi - 0.5f * thv * thv;
            if (ed == 0) s_nobs += nobs;
            if constexpr (FLOWS) {
                p.ability_k[o] = thv;
                if (ed == 0) {
                    p.ability_ladj[row0 + er] = ladj;
                    s_ladj += ladj;
                }
            }
        }

        // ---- decode, masked Bernoulli log-lik, backward: 4 items per lane ----
        // theta goes through a wave-private LDS row and comes back as broadcast reads.  1PL/2PL: the reference
        // clamps the Bernoulli probability (utils.py:46-49 -> torch): log-lik value clamped at logit +-kLogitLo
        // (done always, one v_med3), gradient exactly zero outside [-kLogitLo, kLogitHi] (rare: fix-up pass).
        wl.thl[lane] = thv;
        // row group = RPB rows (one d LL/d theta reduction).  Software pipeline over groups: the LDS reads of
        // group g+1 (codes, theta) and the transposed partials of group g-1 are issued before group g's math.
        struct GroupIn {
            uint32_t cw[RPB];
            float2v th2[RPB][H];
        };
        auto fetch_group = [&](const int g, GroupIn& gi) {
#pragma unroll
            for (int rr = 0; rr < RPB; ++rr) {
                const int r = g * RPB + rr;
                gi.cw[rr] = wl.codes[r][lane];
                if constexpr (AT >= 4) {
#pragma unroll
                    for (int a = 0; a < AT; a += 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(&wl.thl[r * AT + a]);
                        gi.th2[rr][a / 2] = float2v{t4.x, t4.y};
                        gi.th2[rr][a / 2 + 1] = float2v{t4.z, t4.w};
                    }
                } else {
                    gi.th2[rr][0] = *reinterpret_cast<const float2v*>(&wl.thl[r * AT]);
                }
            }
        };
        auto finish_group = [&](const int g, const float4 ru, const float4 rv) {
            float t = ((ru.x + ru.y) + (ru.z + ru.w)) + ((rv.x + rv.y) + (rv.z + rv.w));
            t += dpp_f<0xb1>(t);                     // quad_perm [1,0,3,2]
            t += dpp_f<0x4e>(t);                     // quad_perm [2,3,0,1]
            t += dpp_f<0x141>(t);                    // row_half_mirror
            if ((lane & 7) == 0 && (lane >> 3) < NVG) wl.gthp[g * NVG + (lane >> 3)] = t;
        };
        auto do_group = [&](const int g, const GroupIn& gi, GroupIn& nxt) {
            if (g + 1 < G) fetch_group(g + 1, nxt);
            float4 ru = float4{0.f, 0.f, 0.f, 0.f}, rv = ru;
            if constexpr (GRAD) {
                if (g > 0) {
                    // transposed read-back of group g-1's 8 x 64 partials: lane (k = l >> 3, s = l & 7) sums 8
                    // lanes of value k, then 3 DPP steps finish the 64-lane sum
                    const float* src = &wl.gtl[(g - 1) & 1][lane >> 3][(lane & 7) * 8];
                    ru = *reinterpret_cast<const float4*>(src);
                    rv = *reinterpret_cast<const float4*>(src + 4);
                }
            }
            float2v gth2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) gth2[k] = float2v{0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < RPB; ++rr) {
                const float2v(&th2)[H] = gi.th2[rr];
                float lg[4];
                float lmax = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float2v l2 = float2v{nb[t], 0.f};
#pragma unroll
                    for (int h = 0; h < H; ++h) l2 = na2[t][h] * th2[h] + l2;
                    lg[t] = l2[0] + l2[1];
                    if constexpr (IRT != 3) lmax = fmaxf(lmax, fabsf(lg[t]));
                }
                const float2v w01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)gi.cw[rr], false);
                const float2v w23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)gi.cw[rr], true);
                const float w[4] = {w01[0], w01[1], w23[0], w23[1]};
                float prod = 1.0f;
                float gls[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float gl = 0.f;
                    if constexpr (IRT != 3) {
                        const float lc = med3(lg[t], -kLoS, kLoS);
                        const float eu = fast_exp2(-w[t] * lc);           // exactly 1 for a missing cell (w = 0)
                        const float tt = 1.0f + eu;
                        prod *= tt;             // <= (1 + 2^23)^4: one log2 per 4 terms; the 2s of missing cells
                                                // are taken out per batch (s_log correction above)
                        if constexpr (GRAD) gl = w[t] * (eu * fast_rcp(tt));   // d ll / d logit
                    } else {
                        // 3PL: p = guess + (1 - guess) sigmoid(l)  (models.py:758-765), probability clamp of
                        // torch's Bernoulli on p itself
                        const float l = lg[t];
                        const float e = fast_exp2(-fabsf(l));
                        const float rr_ = fast_rcp(1.0f + e);
                        const float er_ = e * rr_;
                        const float sp = (l >= 0.f) ? rr_ : er_;       // sigmoid(l)
                        const float sn = (l >= 0.f) ? er_ : rr_;       // sigmoid(-l)
                        const float pr = fmaf(om[t], sp, gs[t]);       // P(correct)
                        const float qr = om[t] * sn;                   // P(wrong)
                        const float pc = med3(pr, kEps32, 1.0f - kEps32);
                        const float arg = (w[t] > 0.f) ? pc : med3(qr, kEps32, 1.0f - kEps32);
                        prod *= (w[t] != 0.f) ? arg : 1.0f;            // >= eps32^4: no underflow
                        if constexpr (GRAD) {
                            const float wlv = (pr == pc) ? w[t] : 0.f;                  // clamp kills the gradient
                            const float common = wlv * fast_rcp(arg) * om[t] * sn;     // (x/p-(1-x)/(1-p)) (1-g) sig(-l)
                            gl = common * sp;                                          // * d p / d logit
                            acc_g[t] = fmaf(common, gs[t], acc_g[t]);                  // * d p / d guess-logit
                        }
                    }
                    if constexpr (GRAD) {
                        gls[t] = gl;
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            gth2[rr * H + h] = na2[t][h] * gl + gth2[rr * H + h];      // x log2e, removed below
                            if (IRT != 1) acc_a2[t][h] = th2[h] * gl + acc_a2[t][h];   // = -d/d a_ia
                        }
                        acc_b[t] += gl;
                    }
                }
                s_log += fast_log2(prod);
                if constexpr (GRAD && IRT != 3) {
                    if (__any(lmax > kLoS)) {
                        // rare: the reference's gradient is exactly zero outside [-kLogitLo, kLogitHi]; take the
                        // contributions of those cells back out
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float gl = (lg[t] < -kLoS || lg[t] > kHiS) ? -gls[t] : 0.f;
#pragma unroll
                            for (int h = 0; h < H; ++h) {
                                gth2[rr * H + h] = na2[t][h] * gl + gth2[rr * H + h];
                                if (IRT != 1) acc_a2[t][h] = th2[h] * gl + acc_a2[t][h];
                            }
                            acc_b[t] += gl;
                        }
                    }
                }
            }
            if constexpr (GRAD) {
                if (g > 0) finish_group(g - 1, ru, rv);
                // 64-lane sums of the 8 partials go through an LDS transpose, one group behind the math
#pragma unroll
                for (int k = 0; k < 8; ++k) wl.gtl[g & 1][k][lane] = gth2[k >> 1][k & 1];
            }
        };
        {
            GroupIn ga, gb;
            fetch_group(0, ga);
#pragma unroll 1
            for (int g = 0; g < G; g += 2) {
                do_group(g, ga, gb);
                if (g + 1 < G) do_group(g + 1, gb, ga);
            }
        }
        if constexpr (GRAD) {
            const float* src = &wl.gtl[(G - 1) & 1][lane >> 3][(lane & 7) * 8];
            finish_group(G - 1, *reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4));
            __syncthreads();
            // ---- wave 0, lane (er, ed): backward through the flows, the sample and the PoE ----
            if (q == 0) {
                float g0 = 0.f;
#pragma unroll
                for (int w = 0; w < (NQT > 0 ? NQT : 4); ++w)
                    if (w < nq) g0 += wls[w].gthp[lane & (NE - 1)];
                float gz0 = live ? g0 * kLn2 : 0.f;                                    // d LL  / d theta_K
                // the backward is linear in d LL/d theta, so every panel backpropagates its own partial sum; the
                // regulariser's own terms (set 1) belong to the primary launch only
                const bool reg_on = live && p.primary;
                float gz1 = (reg_on && p.reg_mode != 0) ? thv : 0.f;                   // d REG / d theta_K (-log p)
                if constexpr (FLOWS) {
#pragma unroll
                    for (int f = kMF - 1; f >= 0; --f) {
                        if (f < p.n_flows) {
                            const float ud = cl.fpar[f][0][ed], wd = cl.fpar[f][1][ed];
                            const float cwu = cl.fsc[f][1];
                            const float t = fa.st[f][0][lane], psi = fa.st[f][1][lane], zin = fa.st[f][2][lane];
                            const float omt = 1.0f - t * t;
                            const float unit = ((psi >= 0.f) ? 1.0f : -1.0f) / (fabsf(psi) + 1e-8f);
                            const float lv = live ? 1.0f : 0.f, lv1 = reg_on ? 1.0f : 0.f;
                            // set 0 (LL): no log-det term
                            {
                                const float g_t = group_sum<AT>(gz0 * ud);
                                const float g_a = g_t * omt;
                                fa.a[0][f][0][lane] += gz0 * t;
                                fa.a[0][f][1][lane] += lv * g_a * zin;
                                fa.a[0][f][2][lane] += lv * g_a;
                                gz0 = fmaf(g_a, wd, gz0) * lv;
                            }
                            // set 1 (REG = log q0 - sum log|det| - log p): d REG / d ladj = -1
                            {
                                const float dl_dpsi = -unit;
                                const float g_t = dl_dpsi * (-2.0f * t * cwu) + group_sum<AT>(gz1 * ud);
                                const float g_c = dl_dpsi * omt;
                                const float g_a = g_t * omt;
                                fa.a[1][f][0][lane] += lv1 * (gz1 * t + g_c * wd);
                                fa.a[1][f][1][lane] += lv1 * (g_a * zin + g_c * ud);
                                fa.a[1][f][2][lane] += lv1 * g_a;
                                gz1 = fmaf(g_a, wd, gz1) * lv1;
                            }
                        }
                    }
                }
                const float h = 0.5f * sig * eps_c;
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
                if (p.given_grad) {
                    if (live) {
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            float* gg = p.given_grad + ((size_t)st * p.B + (row0 + er)) * 2 * A;
                            gg[ed] = gmu[st];
                            gg[A + ed] = glv[st];
                        }
                    }
                }
                if (p.post_coef) {
                    if (live) {
                        float* pc = p.post_coef + (size_t)(row0 + er) * 4 * A;
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            pc[(st * 2 + 0) * A + ed] = gmu[st] * inv_lam;
                            pc[(st * 2 + 1) * A + ed] = -(gmu[st] * amu + glv[st]) * inv_lam;
                        }
                    }
                }
                const float nn[2] = {n0, n1};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float tau = c ? tau1 : tau0;
                    const float te = cl.ctab[(2 * 2 + c) * AT + ed], mm = cl.ctab[(3 * 2 + c) * AT + ed];
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
        const float ll = (IRT == 3 ? kLn2 : -kLn2) * wave_total(s_log - (float)unobs);
        const float t_kl = wave_total(s_kl), t_q0 = wave_total(s_logq0), t_lp = wave_total(s_logp);
        const float t_no = wave_total(s_nobs), t_la = wave_total(s_ladj);
        if (lane == 0) {
            wl.red[0] = ll; wl.red[1] = t_kl; wl.red[2] = t_q0; wl.red[3] = t_lp; wl.red[4] = t_la; wl.red[5] = t_no;
            wl.red[6] = 0.f; wl.red[7] = 0.f;
        }
        if (q == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) cl.tred[k][lane] = acc_t[k];
        }
    }
    __syncthreads();
    if (tid < 8) {
        float t = 0.f;
        for (int w = 0; w < nq; ++w) t += wls[w].red[tid];
        out[tid] = (tid < 6) ? t : 0.f;
    }
    if constexpr (GRAD) {
        if (q == nq - 1 && lane < 8 * A) {        // last wave: lane = a * 8 + k
            const int a = lane >> 3, k = lane & 7;
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) t += cl.tred[k][r * AT + a];
            const int st = k >> 2, c = (k >> 1) & 1, ms = k & 1;
            out[p.lay.off_table + (st * 2 + c) * 2 * A + ms * A + a] = t;
        }
        if constexpr (FLOWS) {
            const int per = 2 * A + 1;
            for (int idx = tid; idx < 2 * p.n_flows * per; idx += blockDim.x) {
                const int s = idx / (p.n_flows * per), f = (idx / per) % p.n_flows, j = idx % per;
                const int kind = j < A ? 0 : j < 2 * A ? 1 : 2;
                const int a = kind == 0 ? j : kind == 1 ? j - A : 0;
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) t += fa.a[s][f][kind][r * AT + a];
                out[p.lay.off_flow + idx] = t;
            }
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
                            float4{-acc_a2[0][a >> 1][a & 1], -acc_a2[1][a >> 1][a & 1], -acc_a2[2][a >> 1][a & 1],
                                   -acc_a2[3][a >> 1][a & 1]};
                *reinterpret_cast<float4*>(oi + (size_t)A * p.lay.i_pad) = float4{acc_b[0], acc_b[1], acc_b[2], acc_b[3]};
                if (IRT == 3)
                    *reinterpret_cast<float4*>(oi + (size_t)(A + 1) * p.lay.i_pad) =
                        float4{acc_g[0], acc_g[1], acc_g[2], acc_g[3]};
            }
        }
    }
}

template <int AT, int IRT, bool GRAD, int RM>
static hipError_t launch_split_flows(const ElboParams& p, int nq, int grid, hipStream_t s) {
    const bool flows = p.n_flows > 0;
    const size_t lds = split_lds_bytes(nq, flows);
    if (nq == 4) {
        if (flows) hipLaunchKernelGGL((split_kernel<AT, IRT, GRAD, true, 4, RM>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((split_kernel<AT, IRT, GRAD, false, 4, RM>), dim3(grid), dim3(256), 0, s, p);
    } else {
        if (flows) hipLaunchKernelGGL((split_kernel<AT, IRT, GRAD, true, 0, RM>), dim3(grid), dim3(64 * nq), lds, s, p);
        else hipLaunchKernelGGL((split_kernel<AT, IRT, GRAD, false, 0, RM>), dim3(grid), dim3(64 * nq), lds, s, p);
    }
    return hipGetLastError();
}

template <int AT, int RM = 0>
static hipError_t launch_split_at(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s) {
    if (irt == 1) return grad ? launch_split_flows<AT, 1, true, RM>(p, nq, grid, s) : launch_split_flows<AT, 1, false, RM>(p, nq, grid, s);
    if (irt == 2) return grad ? launch_split_flows<AT, 2, true, RM>(p, nq, grid, s) : launch_split_flows<AT, 2, false, RM>(p, nq, grid, s);
    return grad ? launch_split_flows<AT, 3, true, RM>(p, nq, grid, s) : launch_split_flows<AT, 3, false, RM>(p, nq, grid, s);
}

}  // namespace vibo
