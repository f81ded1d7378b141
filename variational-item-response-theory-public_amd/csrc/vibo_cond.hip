// vibo_cond.hip -- conditional posterior q(theta | responses, items) (models.py:664-710) around the row-split kernel.
//
// With --conditional-posterior the encoder table has one (mu, logvar) row per (response code, item):
// table[2][I][2A].  The product of experts of a person is then a gather over the row, and the table gradient a
// scatter over (code, item).  Both keep the row-split kernel's mapping (an item never leaves its lane):
//   cond_pre_kernel   lane holds tau = 1/(exp(logvar)+eps) and mu*tau of its 4 items for both codes; per row it sums
//                     the experts the codes select, 8-row butterflies + LDS finish the row sums
//                     -> pre_stats[panel][B][2A+1] = lam | s | nobs
//   split_kernel      (vibo_split_kernel.hpp) takes lam, s from pre_stats instead of the 2-row table; its per-person
//                     backward writes P1 = gmu/lam, P2 = -(gmu amu + glv)/lam per head -> post_coef[panel][B][2][2][A]
//   cond_post_kernel  lane accumulates S1 = sum_p [code] P1, S2 = sum_p [code] P2 per (head, code, dim) of its items
//                     in registers; at the end  d/d mu = S1 tau,  d/d logvar = -(S1 mu + S2) tau^2 exp(logvar)
//   cond_finalize_kernel  fixed-order fp64 sum of the per-workgroup records -> grad_table[2 heads][2][I][2A]
// Three passes over the response matrix (5 B/term each) instead of one; every kernel streams rows exactly like the
// row-split kernel (16-byte loads one batch ahead).  A launch covers 4 ability dims (8 x 4 accumulators per item in
// cond_post); wider posteriors (ability_dim 5..8) run cond_pre / cond_post twice, dims [0,4) and [4,A): deterministic, no atomics.
#include <hip/hip_runtime.h>
#include "../../include/vibo_hip.h"
#include "vibo_cond.hpp"
#include "vibo_cond_finalize.hpp"
#include <string.h>
#include "vibo_device.hpp"
#include "vibo_split_kernel.hpp"

namespace vibo {

// 8 values per lane -> lane l returns the 64-lane sum of value (l >> 3)   (float twin of bfly8(int))
__device__ __forceinline__ float bfly8f(const float (&v)[8], const int lane) {
    float w[4], u[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned a = __builtin_bit_cast(unsigned, v[k]), b = __builtin_bit_cast(unsigned, v[k + 4]);
        swap32(a, b);
        w[k] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned a = __builtin_bit_cast(unsigned, w[k]), b = __builtin_bit_cast(unsigned, w[k + 2]);
        swap16(a, b);
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

constexpr int kCR = 8;   // rows per batch

// cells of a lane's chunk that lie inside the row (I % 4 != 0 with 16-byte padded row strides); applied where the
// mask word is consumed, so that the loads stay in flight
__device__ __forceinline__ uint32_t chunk_tail_mask(const CondParams& p, const int chunk) {
    return ((p.I & 3) && chunk == (p.I >> 2)) ? ((1u << (8 * (p.I & 3))) - 1u) : 0xFFFFFFFFu;
}

// CODES: the rows are 1-byte cell codes (VIBO_MASK_CODES through p.mask): no fp32 row registers
template <bool CODES>
struct RowBatch {
    float4 x[CODES ? 1 : kCR];
    uint32_t m[kCR];
};
template <bool CODES>
__device__ __forceinline__ void load_rows(const CondParams& p, const long long bt, const int chunk, const bool chunk_ok,
                                          RowBatch<CODES>& rb) {
    const long long row0 = bt * kCR;
#pragma unroll
    for (int r = 0; r < kCR; ++r) {
        const long long row = row0 + r;
        if constexpr (!CODES) rb.x[r] = float4{0.f, 0.f, 0.f, 0.f};
        rb.m[r] = CODES ? kAllMissing4 : 0u;
        if (row < p.B && chunk_ok) {
            const long long src = p.row_index ? p.row_index[row] : row;
            if constexpr (!CODES) rb.x[r] = nt_load4(reinterpret_cast<const float4*>(p.response + src * p.resp_stride + p.item0) + chunk);
            if (CODES || p.mask_dtype == 0)
                rb.m[r] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride + p.item0)[chunk];
            else
                rb.m[r] = 0x01010101u;
        }
    }
}
// The word a pass keeps per row and lane: the 4 cell codes as they are (CODES) or the 4 fp8 codes of an fp32 row.
template <bool CODES>
__device__ __forceinline__ uint32_t row_word(const RowBatch<CODES>& rb, const int r, const uint32_t tail_mask, int& nobs) {
    if constexpr (CODES) {
        return rb.m[r];
    } else {
        int pk = 0;
        const uint32_t cw = pack_codes4(rb.x[r], rb.m[r] & tail_mask, pk);
        nobs = pk & 0xffff;
        return cw;
    }
}
// [correct] / [wrong] indicators (exactly 0.0 / 1.0) of the word's 4 cells.  On cell codes they come straight from the code
// bits (observed = !bit 1, correct = bit 0: one v_cvt_f32_ubyte per value); from fp8 codes (+1 / -1 / 0) through a clamp --
// v_med3, which unlike fmaxf() needs no canonicalising v_max in front of it (that doubled the instruction count here).
template <bool CODES>
__device__ __forceinline__ void word_indicators(const uint32_t wd, const uint32_t tail_mask, float (&wp)[4], float (&wn)[4], int& nobs) {
    if constexpr (CODES) {
        const uint32_t ob = ~(wd >> 1) & (tail_mask & 0x01010101u);
        const uint32_t p1 = wd & ob, p0 = ob ^ p1;
        // (spelled out: hipcc shifts the word first and converts byte 0 otherwise)
        asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(wp[0]) : "v"(p1));
        asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(wp[1]) : "v"(p1));
        asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(wp[2]) : "v"(p1));
        asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(wp[3]) : "v"(p1));
        asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(wn[0]) : "v"(p0));
        asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(wn[1]) : "v"(p0));
        asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(wn[2]) : "v"(p0));
        asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(wn[3]) : "v"(p0));
        nobs = __builtin_popcount(ob);
    } else {
        const float2v w01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wd, false);
        const float2v w23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wd, true);
        const float w[4] = {w01[0], w01[1], w23[0], w23[1]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wp[j] = __builtin_amdgcn_fmed3f(w[j], 0.f, 1.f);
            wn[j] = __builtin_amdgcn_fmed3f(-w[j], 0.f, 1.f);
        }
    }
}

// ---------------------------------------------------------------------------
template <int AT, bool CODES>
__global__ __launch_bounds__(256, CODES ? 3 : 2) void cond_pre_kernel(const CondParams p_in) {
    constexpr int NV = 2 * AT + 1;
    __shared__ float part[4][kCR][NV];
    // all panels of a wide row in one launch (CondParams::panel_count): this workgroup's panel, its share of the row batches
    CondParams p = p_in;
    long long bt0 = blockIdx.x, bstep = gridDim.x;
    if (p_in.panel_count > 1) {
        const int panel = (int)(blockIdx.x % (unsigned)p_in.panel_count);
        bt0 = blockIdx.x / (unsigned)p_in.panel_count;
        bstep = gridDim.x / (unsigned)p_in.panel_count;
        p.item0 = panel * 1024;
        p.I = p.I_total - p.item0 < 1024 ? p.I_total - p.item0 : 1024;
        p.pre_out = p_in.pre_out + (size_t)panel * p.B * (2 * p.A + 1);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = blockDim.x >> 6;
    const int A = p.A;
    const int chunk = q * 64 + lane;
    const bool chunk_ok = chunk < ((p.I + 3) >> 2);
    const uint32_t tail_mask = chunk_tail_mask(p, chunk);
    float tau[4][2][AT], mt[4][2][AT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // (the last chunk of a row whose item count is not a multiple of 4 covers items past the table's end: those
            // cells carry code 0, but 0 x (whatever lies behind the table) must not be formed -- it may be Inf / NaN)
            const int item = p.item0 + 4 * chunk + j;
            const bool item_ok = chunk_ok && item < p.I_total;
            const float* te = p.table + ((size_t)c * p.I_total + (item_ok ? item : 0)) * 2 * A;
#pragma unroll
            for (int a = 0; a < AT; ++a) {
                float t = 0.f, mm = 0.f;
                if (item_ok && p.a0 + a < A) {
                    t = 1.0f / (expf(te[A + p.a0 + a]) + kPoeEps);       // utils.py:105-113
                    mm = te[p.a0 + a] * t;
                }
                tau[j][c][a] = t;
                mt[j][c][a] = mm;
            }
        }
    const long long n_batches = ((long long)p.B + kCR - 1) / kCR;
    RowBatch<CODES> rb;
    long long bt = bt0;
    if (bt < n_batches) load_rows(p, bt, chunk, chunk_ok, rb);
    for (; bt < n_batches; bt += bstep) {
        const long long row0 = bt * kCR;
        float v[NV][kCR];
#pragma unroll
        for (int r = 0; r < kCR; ++r) {
            int nobs = 0;
            const uint32_t cw = row_word(rb, r, tail_mask, nobs);
            if constexpr (!CODES) {          // leave the row behind as cell codes for the passes that follow (1 B/cell instead of 5)
                if (p.codes_out && chunk_ok && row0 + r < p.B)
                    nt_store1(reinterpret_cast<uint32_t*>(p.codes_out + (row0 + r) * p.codes_stride + p.item0) + chunk,
                              cell_codes4(rb.x[r], rb.m[r] & tail_mask));
            }
            float wp[4], wn[4];                                              // [correct], [wrong]
            word_indicators<CODES>(cw, tail_mask, wp, wn, nobs);
#pragma unroll
            for (int a = 0; a < AT; ++a) { v[a][r] = 0.f; v[AT + a][r] = 0.f; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int a = 0; a < AT; ++a) {
                    v[a][r] = fmaf(wp[j], tau[j][1][a], fmaf(wn[j], tau[j][0][a], v[a][r]));
                    v[AT + a][r] = fmaf(wp[j], mt[j][1][a], fmaf(wn[j], mt[j][0][a], v[AT + a][r]));
                }
            }
            v[2 * AT][r] = (float)nobs;
        }
        if (bt + bstep < n_batches) load_rows(p, bt + bstep, chunk, chunk_ok, rb);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const float t = bfly8f(v[k], lane);            // lane l: row l >> 3
            if ((lane & 7) == 0) part[q][lane >> 3][k] = t;
        }
        __syncthreads();
        for (int e = tid; e < kCR * NV; e += blockDim.x) {     // (one-wave workgroups have fewer threads than outputs)
            const int r = e / NV, k = e % NV;
            float t = 0.f;
            for (int w = 0; w < nq; ++w) t += part[w][r][k];
            const long long row = row0 + r;
            // [lam 0..A) | s 0..A) | nobs]: drop the padded dims
            const int a = p.a0 + (k < AT ? k : k - AT);
            if (row < p.B) {
                if (k == 2 * AT) p.pre_out[row * (2 * A + 1) + 2 * A] = t;
                else if (a < A) p.pre_out[row * (2 * A + 1) + (k < AT ? a : A + a)] = t;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
template <int AT, bool CODES>
__global__ __launch_bounds__(256, AT <= 2 ? (CODES ? 3 : 2) : 1) void cond_post_kernel(const CondParams p) {
    constexpr int NC = 4 * AT;                       // coefficients per person: [head][P1|P2][dim]
    __shared__ __attribute__((aligned(16))) float cbuf[4][kCR][NC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = p.A;
    const int chunk = q * 64 + lane;
    const bool chunk_ok = chunk < ((p.I + 3) >> 2);
    const uint32_t tail_mask = chunk_tail_mask(p, chunk);
    float S[4][2][NC];                               // [item][code][head, P1|P2, dim]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < NC; ++k) S[j][c][k] = 0.f;
    const long long n_batches = ((long long)p.B + kCR - 1) / kCR;
    RowBatch<CODES> rb;
    long long bt = blockIdx.x;
    if (bt < n_batches) load_rows(p, bt, chunk, chunk_ok, rb);
    for (; bt < n_batches; bt += gridDim.x) {
        const long long row0 = bt * kCR;
        uint32_t cw[kCR];
#pragma unroll
        for (int r = 0; r < kCR; ++r) {
            int nobs = 0;
            cw[r] = row_word(rb, r, tail_mask, nobs);
        }
        if (bt + gridDim.x < n_batches) load_rows(p, bt + gridDim.x, chunk, chunk_ok, rb);
        // this batch's coefficients (sum over the panels' shares), wave-private copy
        for (int e = lane; e < kCR * NC; e += 64) {
            const int r = e / NC, k = e % NC, a = p.a0 + k % AT, hk = k / AT;
            float t = 0.f;
            if (row0 + r < p.B && a < A)
                for (int pn = 0; pn < p.coef_panels; ++pn)
                    t += p.coef_in[((size_t)pn * p.B + (row0 + r)) * 4 * A + hk * A + a];
            cbuf[q][r][k] = t;
        }
#pragma unroll
        for (int r = 0; r < kCR; ++r) {
            float wp[4], wn[4];
            int nobs = 0;
            word_indicators<CODES>(cw[r], tail_mask, wp, wn, nobs);
            float C[NC];
#pragma unroll
            for (int k = 0; k < NC; k += 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(&cbuf[q][r][k]);
                C[k] = t4.x; C[k + 1] = t4.y; C[k + 2] = t4.z; C[k + 3] = t4.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    S[j][1][k] = fmaf(wp[j], C[k], S[j][1][k]);
                    S[j][0][k] = fmaf(wn[j], C[k], S[j][0][k]);
                }
            }
        }
    }
    // d / d table of this workgroup's share:  record [head][code][mu dims | logvar dims][1024 items]
    if (chunk_ok) {
        float* out = p.partial + (size_t)blockIdx.x * p.rec_stride + 4 * chunk;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int al = 0; al < AT; ++al) {
                const int a = p.a0 + al;
                if (a >= A) continue;
                float tauv[4], muv[4], esv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int item = p.item0 + 4 * chunk + j;          // (past the row's end: read item 0, the result is never used)
                    const float* te = p.table + ((size_t)c * p.I_total + (item < p.I_total ? item : 0)) * 2 * A;
                    esv[j] = expf(te[A + a]);
                    tauv[j] = 1.0f / (esv[j] + kPoeEps);
                    muv[j] = te[a];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float gm[4], gl[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float s1 = S[j][c][(h * 2 + 0) * AT + al], s2 = S[j][c][(h * 2 + 1) * AT + al];
                        gm[j] = s1 * tauv[j];
                        gl[j] = -(s1 * muv[j] + s2) * tauv[j] * tauv[j] * esv[j];
                    }
                    *reinterpret_cast<float4*>(out + (size_t)((h * 2 + c) * 2 * A + a) * 1024) = float4{gm[0], gm[1], gm[2], gm[3]};
                    *reinterpret_cast<float4*>(out + (size_t)((h * 2 + c) * 2 * A + A + a) * 1024) = float4{gl[0], gl[1], gl[2], gl[3]};
                }
            }
    }
}

// grad_table[head][code][i][j2] = sum over the panel's workgroup records (fp64, fixed order): vibo_cond_finalize.hpp
__global__ __launch_bounds__(1024) void cond_finalize_kernel(const CondFinTail t) {
    __shared__ __attribute__((aligned(16))) double smem[1024];
    cond_finalize_body(t, blockIdx.x, blockIdx.y, smem);
}

hipError_t launch_cond_pre(const CondParams& p, int at, int nq, int grid, hipStream_t s) {
    const bool codes = p.mask_dtype == 3;      // VIBO_MASK_CODES
    if (at <= 1) {
        if (codes) hipLaunchKernelGGL((cond_pre_kernel<1, true>), dim3(grid), dim3(64 * nq), 0, s, p);
        else hipLaunchKernelGGL((cond_pre_kernel<1, false>), dim3(grid), dim3(64 * nq), 0, s, p);
    } else if (at <= 2) {
        if (codes) hipLaunchKernelGGL((cond_pre_kernel<2, true>), dim3(grid), dim3(64 * nq), 0, s, p);
        else hipLaunchKernelGGL((cond_pre_kernel<2, false>), dim3(grid), dim3(64 * nq), 0, s, p);
    } else {
        if (codes) hipLaunchKernelGGL((cond_pre_kernel<4, true>), dim3(grid), dim3(64 * nq), 0, s, p);
        else hipLaunchKernelGGL((cond_pre_kernel<4, false>), dim3(grid), dim3(64 * nq), 0, s, p);
    }
    return hipGetLastError();
}
hipError_t launch_cond_post(const CondParams& p, int at, int nq, int grid, hipStream_t s) {
    const bool codes = p.mask_dtype == 3;
    if (at <= 1) {
        if (codes) hipLaunchKernelGGL((cond_post_kernel<1, true>), dim3(grid), dim3(64 * nq), 0, s, p);
        else hipLaunchKernelGGL((cond_post_kernel<1, false>), dim3(grid), dim3(64 * nq), 0, s, p);
    } else if (at <= 2) {
        if (codes) hipLaunchKernelGGL((cond_post_kernel<2, true>), dim3(grid), dim3(64 * nq), 0, s, p);
        else hipLaunchKernelGGL((cond_post_kernel<2, false>), dim3(grid), dim3(64 * nq), 0, s, p);
    } else {
        if (codes) hipLaunchKernelGGL((cond_post_kernel<4, true>), dim3(grid), dim3(64 * nq), 0, s, p);
        else hipLaunchKernelGGL((cond_post_kernel<4, false>), dim3(grid), dim3(64 * nq), 0, s, p);
    }
    return hipGetLastError();
}
hipError_t launch_cond_finalize(const float* partial, float* grad_table, int I, int A, int panels, int bpp, int rec_stride,
                                hipStream_t s, CondFinTail* defer) {
    CondFinTail t;
    memset(&t, 0, sizeof(t));
    t.kind = 1; t.gx = (I + 63) / 64; t.gy = 2 * 2 * 2 * A; t.rec = partial; t.grad_table = grad_table; t.I = I; t.A = A;
    t.bpp = bpp; t.rec_stride = rec_stride;
    (void)panels;
    if (defer) { *defer = t; return hipSuccess; }        // (rides in the ELBO finalize launch)
    hipLaunchKernelGGL(cond_finalize_kernel, dim3(t.gx, t.gy), dim3(1024), 0, s, t);
    return hipGetLastError();
}

}  // namespace vibo
