// vibo_decoder.hip -- the per-term MLP decoders of --generative-model link | deep | residual (models.py:769-919) as one
// fused forward + backward kernel on the matrix pipe.
//
// All three decoders end in the same per-(person, item) network  64 -> ELU -> [64 x 64] -> ELU -> 64 -> 1:
//   z1[p,i,:] = U[i,:] + V[p,:] + w1 * l[p,i]        (deep / residual: U = W1a . mlp_item_feat(item_i),
//                                                      V = W1b . mlp_ability(theta_p) + b1, w1 = 0;
//                                                      link: U = 0, V = b1, w1 = first layer, l = the IRT logit)
//   o[p,i]    = w3 . ELU(W2 . ELU(z1) + b2) + b3 + resid * l[p,i]      (residual: resid = 1, models.py:900-913)
//   p         = sigmoid(o)  or  guess_i + (1 - guess_i) sigmoid(o)      (3PL link / residual)
//   ll        = masked Bernoulli log-likelihood (utils.py:46-49, torch's probs clamp)
// The per-item / per-person halves (U, V, the IRT logit l and everything upstream of them) are a few small dense layers
// that stay in PyTorch; this kernel does the O(B I 64^2) part: per term one 64x64 mat-vec forward, one backward, and
// the outer product for d W2 -- 24.6 kFLOP per term, the only genuinely dense contraction of the model family.
//
// Matrix-pipe formulation (v_mfma_f32_16x16x32_f16, operands split x = hi + lo in f16, hi.hi + hi.lo + lo.hi with fp32
// accumulation: fp32-grade products).  A wave owns 16 items and walks its persons two at a time (32 terms):
//   forward   Z2^T[n, t] = W2[n, :] . h1[t, :]     A = W2 (LDS image), B = h1: lane (t = lane & 15, g = lane >> 4)
//             holds h1[t, k] for k in K(g) = {16 kt + 4 g + j}: exactly what the lane computed element-wise
//   backward  dH1^T[k, t] = W2[:, k] . dz2[t, :]   A = W2^T (LDS image), B = dz2: the D layout of the forward product
//             (lane (t, n = 16 nt + 4 g + j)) IS the B layout of this one for the K order n in K(g): no shuffle, and
//             its D layout returns d h1 to the lane that holds h1 / z1 of the same (t, k)
//   d W2[n,k] = sum_t dz2[t, n] h1[t, k]           terms along K: both factors go through a wave-private LDS image
//             (rows = terms) and come back transposed by ds_read_b64_tr_b16; accumulated in registers for the whole launch
// Element-wise work (2 x 64 ELU per term, the f16 splits) runs on the VALU between the MFMAs of the same wave.
// Outputs are per-workgroup / per-wave partial records (fixed order: bitwise reproducible); the host sums them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"

namespace vibo {

typedef _Float16 dh8 __attribute__((ext_vector_type(8)));
typedef _Float16 dh4 __attribute__((ext_vector_type(4)));
typedef _Float16 dh2 __attribute__((ext_vector_type(2)));
typedef float df4 __attribute__((ext_vector_type(4)));
typedef short ds4 __attribute__((ext_vector_type(4)));

constexpr int kH = 64;             // hidden width of the per-term network (models.py:777, 824: hidden_dim = 64)
constexpr int kXRow = 72;          // halfs per row of the transposition image (64 + 8 of padding: 144-byte rows)

struct DecParams {
    const float* response; const uint8_t* mask;
    long long resp_stride, mask_stride;
    const float* U; const float* V; const float* L; const float* guess; const float* w1;
    const float* W2; const float* b2; const float* w3; const float* b3;
    float resid;
    int B, I, ppc;                 // persons per chunk (blockIdx.y)
    float* ll_part; float* dU_part; float* dV_part; float* dL; float* dguess_part; float* dW2_part; float* dvec_part;
    float* prob_out;
};

struct alignas(16) DecLds {
    _Float16 Fh[4][2][64][8], Fl[4][2][64][8];    // forward A operands: W2[16 nt + m][K(g) of step s], hi | lo
    _Float16 Gh[4][2][64][8], Gl[4][2][64][8];    // backward A operands: W2[K(g) of step s][16 kt + m]
    _Float16 X[4][32 * kXRow];                    // per wave: 32 terms x 64 values, the transposition image
    float b2[kH], w3[kH], w1[kH];
};

__device__ __forceinline__ df4 dmfma(const dh8 a, const dh8 b, const df4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ dh4 dtr16(const _Float16* p) {
    return __builtin_bit_cast(dh4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ds4 __attribute__((address_space(3)))*)p));
}
__device__ __forceinline__ dh2 dpk(float a, float b) { return __builtin_bit_cast(dh2, __builtin_amdgcn_cvt_pkrtz(a, b)); }
// 8 floats -> f16 hi and lo pieces (x = hi + lo to 2^-22)
// (hi = round-toward-zero pair; lo = f16(x - hi) by two mixed-precision fmas that write the two halves of one register)
__device__ __forceinline__ void split8(const float* x, dh8& hi, dh8& lo) {
    uint32_t hw[4], lw[4];
    float xv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = x[e];
#pragma unroll
    for (int e = 0; e < 8; e += 2) hw[e >> 1] = __builtin_bit_cast(uint32_t, dpk(x[e], x[e + 1]));
    lo_pieces4(hw, xv, lw);      // (one asm block ending in the wait states an MFMA consumer needs: vibo_device.hpp)
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const dh2 h = __builtin_bit_cast(dh2, hw[e >> 1]), l = __builtin_bit_cast(dh2, lw[e >> 1]);
        hi[e] = h[0]; hi[e + 1] = h[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}
__device__ __forceinline__ float elu(float z) { return z > 0.f ? z : fast_exp2(z * kLog2e) - 1.0f; }

// HASL: the network sees the IRT logit l (link: through w1; residual: added to the output).
template <bool GRAD, bool HASL>
__global__ __launch_bounds__(256, 1) void decoder_kernel(const DecParams p) {
    __shared__ DecLds sm;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    // ---- weight images (once per workgroup) ----
    for (int idx = tid; idx < 4 * 2 * 64; idx += 256) {
        const int nt = idx >> 7, s = (idx >> 6) & 1, l = idx & 63;
        const int m = l & 15, gg = l >> 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 16 * (2 * s + (e >> 2)) + 4 * gg + (e & 3);       // K(g) order of step s
            const float wf = p.W2[(16 * nt + m) * kH + kk];                    // forward: row n = 16 nt + m
            const float wb = p.W2[kk * kH + 16 * nt + m];                      // backward: contraction over n = kk, row k = 16 nt + m
            const _Float16 fh = (_Float16)dpk(wf, 0.f)[0], bh = (_Float16)dpk(wb, 0.f)[0];
            sm.Fh[nt][s][l][e] = fh; sm.Fl[nt][s][l][e] = dpk(wf - (float)fh, 0.f)[0];
            sm.Gh[nt][s][l][e] = bh; sm.Gl[nt][s][l][e] = dpk(wb - (float)bh, 0.f)[0];
        }
    }
    if (tid < kH) {
        sm.b2[tid] = p.b2[tid]; sm.w3[tid] = p.w3[tid];
        sm.w1[tid] = (HASL && p.w1) ? p.w1[tid] : 0.f;
    }
    __syncthreads();
    const float b3 = p.b3[0];
    const int item = 64 * (int)blockIdx.x + 16 * w + i16;
    const bool item_ok = item < p.I;
    const int it = item_ok ? item : 0;
    // per-lane constants: its 16 hidden units  k(kt, j) = 16 kt + 4 g + j  (index 4 kt + j below)
    float u[16], b2v[16], w3v[16], w1v[16];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const float4 t = p.U ? *reinterpret_cast<const float4*>(p.U + (size_t)it * kH + 16 * kt + 4 * g) : float4{0.f, 0.f, 0.f, 0.f};
        u[4 * kt] = t.x; u[4 * kt + 1] = t.y; u[4 * kt + 2] = t.z; u[4 * kt + 3] = t.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b2v[4 * kt + j] = sm.b2[16 * kt + 4 * g + j];
            w3v[4 * kt + j] = sm.w3[16 * kt + 4 * g + j];
            w1v[4 * kt + j] = sm.w1[16 * kt + 4 * g + j];
        }
    }
    const float guess = (p.guess && item_ok) ? p.guess[item] : 0.f;
    const bool has_guess = p.guess != nullptr;

    df4 accW[4][4];                      // d W2 tile (nt, kt): rows n = 16 nt + 4 g + j, col k = 16 kt + i16
    float dU[16], db2[16], dw3[16], dw1[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) { dU[a] = 0.f; db2[a] = 0.f; dw3[a] = 0.f; dw1[a] = 0.f; accW[a >> 2][a & 3] = df4{0.f, 0.f, 0.f, 0.f}; }
    float db3 = 0.f, dgs = 0.f, llsum = 0.f;
    _Float16* X = &sm.X[w][0];
    const int xw = i16 * kXRow + 4 * g;                         // write: row (16 r +) i16, halfs 16 nt + 4 g ..
    const int xr = (8 * g + (i16 >> 2)) * kXRow + 4 * (i16 & 3);   // transposed read: rows 8 g + (4 h +) (i16 >> 2), cols (16 nt +) 4 (i16 & 3)

    const long long p_begin = (long long)blockIdx.y * p.ppc;
    const long long p_end = p_begin + p.ppc < p.B ? p_begin + p.ppc : p.B;
    // the pair's inputs are loaded one iteration ahead (one wave per SIMD: nobody else hides the load latency)
    struct PersonIn { float4 v[4]; float x, l; unsigned m; };
    PersonIn cur[2], nxt[2];
    auto load_pair = [&](const long long pp, PersonIn (&d)[2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const long long pr = pp + r;
            const long long prc = pr < p_end ? pr : p_end - 1;
            d[r].x = p.response[prc * p.resp_stride + it];
            d[r].m = p.mask ? (unsigned)p.mask[prc * p.mask_stride + it] : 1u;
            d[r].l = 0.f;
            if constexpr (HASL) d[r].l = p.L[prc * (long long)p.I + it];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) d[r].v[kt] = *reinterpret_cast<const float4*>(p.V + (size_t)prc * kH + 16 * kt + 4 * g);
        }
    };
    if (p_begin < p_end) load_pair(p_begin, cur);
#pragma unroll 1
    for (long long pp = p_begin; pp < p_end; pp += 2) {
        load_pair(pp + 2, nxt);
        dh8 h1h[2][2], h1l[2][2], dzh[2][2], dzl[2][2];        // [person of the pair][K step]
        float h1f[2][16], lgt[2];
        float dov[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const long long pr = pp + r;
            const bool ok = pr < p_end && item_ok;
            const float x = cur[r].x;
            const bool obs = ok && cur[r].m != 0;
            const float l = cur[r].l;
            lgt[r] = l;
            // ---- layer 1 (element-wise in the operand layout) ----
            float z1[16];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float4 v = cur[r].v[kt];
                z1[4 * kt] = u[4 * kt] + v.x; z1[4 * kt + 1] = u[4 * kt + 1] + v.y;
                z1[4 * kt + 2] = u[4 * kt + 2] + v.z; z1[4 * kt + 3] = u[4 * kt + 3] + v.w;
            }
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                if constexpr (HASL) z1[a] = fmaf(w1v[a], l, z1[a]);
                h1f[r][a] = elu(z1[a]);
            }
            split8(&h1f[r][0], h1h[r][0], h1l[r][0]);
            split8(&h1f[r][8], h1h[r][1], h1l[r][1]);
            // ---- layer 2: Z2^T = W2 . h1^T ----
            float h2[16];
            float opart = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                df4 acc = df4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const dh8 ah = *reinterpret_cast<const dh8*>(&sm.Fh[nt][s][lane][0]);
                    const dh8 al = *reinterpret_cast<const dh8*>(&sm.Fl[nt][s][lane][0]);
                    acc = dmfma(ah, h1h[r][s], acc);
                    acc = dmfma(ah, h1l[r][s], acc);
                    acc = dmfma(al, h1h[r][s], acc);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float z2 = acc[j] + b2v[4 * nt + j];
                    h2[4 * nt + j] = elu(z2);
                    opart = fmaf(w3v[4 * nt + j], h2[4 * nt + j], opart);
                }
            }
            opart = xor16_add(opart);
            opart = xor32_add(opart);
            float o = opart + b3;
            if constexpr (HASL) o = fmaf(p.resid, l, o);
            // ---- link + masked Bernoulli log-likelihood (utils.py:46-49) ----
            // P and 1 - P formed separately (1 - P = (1 - guess) / (1 + e^o) has no cancellation when P -> 1); torch clamps
            // the probability to [eps32, 1 - eps32] and its gradient is zero outside (utils.py:46-49)
            const float eo = expf(-fabsf(o));
            const float big = 1.0f / (1.0f + eo), small = eo * big;
            const float sg = o >= 0.f ? big : small, sq = o >= 0.f ? small : big;       // sigmoid(o), 1 - sigmoid(o)
            const float pr_ = has_guess ? guess + (1.0f - guess) * sg : sg;
            const float qr_ = has_guess ? (1.0f - guess) * sq : sq;
            const bool inside = pr_ >= kEps32 && pr_ <= 1.0f - kEps32;      // (torch's clamp passes the gradient at the bounds themselves)
            const float pc = inside ? pr_ : fminf(fmaxf(pr_, kEps32), 1.0f - kEps32);
            const float qc = inside ? qr_ : 1.0f - pc;
            float ll = 0.f, dldp = 0.f;
            if (obs) {
                const bool one = x > 0.5f;
                // (value floor: the reference's fp32 1 - P is quantised at eps32, so its log-likelihood of a confidently wrong cell
                //  stops at log eps32 = -15.94 while its gradient lives on until P rounds to 1: tests/golden/saturation.npz)
                ll = logf(fmaxf(one ? pc : qc, kEps32));
                if (inside) dldp = one ? 1.0f / pc : -1.0f / qc;
            }
            if (g == 0) llsum += ll;
            if (p.prob_out && ok && g == 0) p.prob_out[pr * (long long)p.I + item] = pr_;
            const float dsg = dldp * (has_guess ? (1.0f - guess) : 1.0f);      // d ll / d sigmoid
            const float d_o = dsg * sg * sq;
            dov[r] = d_o;
            if constexpr (GRAD) {
                if (g == 0) { dgs += dldp * sq; db3 += d_o; }
                float dz2[16];
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    const float d = d_o * w3v[a] * (h2[a] > 0.f ? 1.0f : h2[a] + 1.0f);       // ELU' = 1 | e^z = h + 1
                    dz2[a] = d;
                    db2[a] += d;
                    dw3[a] = fmaf(d_o, h2[a], dw3[a]);
                }
                split8(&dz2[0], dzh[r][0], dzl[r][0]);
                split8(&dz2[8], dzh[r][1], dzl[r][1]);
            }
        }
        if constexpr (!GRAD) { cur[0] = nxt[0]; cur[1] = nxt[1]; continue; }
        // ---- backward through layer 2: dH1^T = W2^T . dz2^T, then the ELU of layer 1 ----
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const long long pr = pp + r;
            float dz1[16];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                df4 acc = df4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const dh8 ah = *reinterpret_cast<const dh8*>(&sm.Gh[kt][s][lane][0]);
                    const dh8 al = *reinterpret_cast<const dh8*>(&sm.Gl[kt][s][lane][0]);
                    acc = dmfma(ah, dzh[r][s], acc);
                    acc = dmfma(ah, dzl[r][s], acc);
                    acc = dmfma(al, dzh[r][s], acc);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int a = 4 * kt + j;
                    dz1[a] = acc[j] * (h1f[r][a] > 0.f ? 1.0f : h1f[r][a] + 1.0f);
                    dU[a] += dz1[a];
                }
            }
            if constexpr (HASL) {
                float dl = 0.f;
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    dl = fmaf(w1v[a], dz1[a], dl);
                    dw1[a] = fmaf(dz1[a], lgt[r], dw1[a]);
                }
                dl = xor16_add(dl);
                dl = xor32_add(dl);
                dl = fmaf(p.resid, dov[r], dl);
                if (g == 0 && item_ok && pr < p_end) p.dL[pr * (long long)p.I + item] = dl;
            }
            // d V[p, k] = sum over this wave's 16 items (lanes of a row) -> the wave's partial row
            if (p.dV_part) {
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    float v = dz1[a];
                    v += dpp_f<0xb1>(v);         // quad_perm [1,0,3,2]
                    v += dpp_f<0x4e>(v);         // quad_perm [2,3,0,1]
                    v += dpp_f<0x141>(v);        // row_half_mirror
                    v += dpp_f<0x140>(v);        // row_mirror
                    // (kept scalar: left to the SLP vectoriser the sixteen chains became v_pk_add_f32 pairs, which cannot take a
                    //  DPP operand -- two zero moves + two v_mov_dpp + one packed add per pair and step, 320 instructions of the
                    //  loop's 1 680, instead of two v_add_f32_dpp)
                    asm volatile("" : "+v"(v));
                    dz1[a] = v;
                }
                if (i16 == 0 && pr < p_end) {
                    float* dst = p.dV_part + (((size_t)blockIdx.x * 4 + w) * p.B + pr) * kH;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        *reinterpret_cast<float4*>(dst + 16 * kt + 4 * g) = float4{dz1[4 * kt], dz1[4 * kt + 1], dz1[4 * kt + 2], dz1[4 * kt + 3]};
                }
            }
        }
        // ---- d W2 += dz2^T . h1 over the pair's 32 terms (K = terms): operands through the transposition image ----
        dh8 Ah[4], Al[4], Bh[4], Bl[4];
        auto through_image = [&](const dh8 (&src)[2][2], dh8 (&dst)[4]) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const dh8 v = src[r][nt >> 1];
                    const int e0 = 4 * (nt & 1);
                    *reinterpret_cast<dh4*>(X + 16 * r * kXRow + xw + 16 * nt) = dh4{v[e0], v[e0 + 1], v[e0 + 2], v[e0 + 3]};
                }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const dh4 a = dtr16(X + xr + 16 * nt), b = dtr16(X + xr + 4 * kXRow + 16 * nt);
                dst[nt] = dh8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            }
        };
        through_image(dzh, Ah);
        through_image(dzl, Al);
        through_image(h1h, Bh);
        through_image(h1l, Bl);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                df4 acc = accW[nt][kt];
                acc = dmfma(Ah[nt], Bh[kt], acc);
                acc = dmfma(Ah[nt], Bl[kt], acc);
                acc = dmfma(Al[nt], Bh[kt], acc);
                accW[nt][kt] = acc;
            }
        cur[0] = nxt[0]; cur[1] = nxt[1];
    }

    // ================= partial records =================
    const int wave_id = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * 4 + w;
    {
        const float t = wave_total(llsum);
        if (lane == 0) p.ll_part[wave_id] = t;
    }
    if constexpr (GRAD) {
        float* dW = p.dW2_part + (size_t)wave_id * kH * kH;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) dW[(16 * nt + 4 * g + j) * kH + 16 * kt + i16] = accW[nt][kt][j];
        // vectors: sums over the wave's 16 items
        float* dv = p.dvec_part + (size_t)wave_id * 4 * kH;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            float v0 = db2[a], v1 = dw3[a], v2 = dw1[a];
            v0 += dpp_f<0xb1>(v0); v0 += dpp_f<0x4e>(v0); v0 += dpp_f<0x141>(v0); v0 += dpp_f<0x140>(v0);
            v1 += dpp_f<0xb1>(v1); v1 += dpp_f<0x4e>(v1); v1 += dpp_f<0x141>(v1); v1 += dpp_f<0x140>(v1);
            v2 += dpp_f<0xb1>(v2); v2 += dpp_f<0x4e>(v2); v2 += dpp_f<0x141>(v2); v2 += dpp_f<0x140>(v2);
            if (i16 == 0) {
                const int k = 16 * (a >> 2) + 4 * g + (a & 3);
                dv[0 * kH + k] = v0;
                dv[1 * kH + k] = v1;
                dv[2 * kH + k] = v2;
            }
        }
        {
            const float t = wave_total(db3);
            dv[3 * kH + lane] = lane == 0 ? t : 0.f;          // (the rest of the row is zero: outputs are fully overwritten)
        }
        if (item_ok) {
            if (p.dU_part) {
                float* du = p.dU_part + ((size_t)blockIdx.y * p.I + item) * kH;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    *reinterpret_cast<float4*>(du + 16 * kt + 4 * g) = float4{dU[4 * kt], dU[4 * kt + 1], dU[4 * kt + 2], dU[4 * kt + 3]};
            }
            if (p.dguess_part && g == 0) p.dguess_part[(size_t)blockIdx.y * p.I + item] = dgs;
        }
    }
}

}  // namespace vibo

using namespace vibo;

static int decoder_check(const vibo_decoder_desc* d) {
    if (!d) return -1;
    if (d->num_person < 1 || d->num_item < 1) return -2;
    if (d->hidden_dim != kH) return -6;
    if (d->person_chunks < 1 || d->person_chunks > d->num_person) return -3;
    return 0;
}

extern "C" int vibo_decoder_person_chunks(int num_person, int num_item) {
    if (num_person < 1 || num_item < 1) return 0;
    int dev = 0, n = 0;
    const int cus = (hipGetDevice(&dev) == hipSuccess &&
                     hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    const int n_ib = (num_item + 63) / 64;
    long long chunks = ((long long)cus * 2 + n_ib - 1) / n_ib;         // about two workgroups per CU in flight
    const long long most = ((long long)num_person + 1) / 2;            // at least one person pair per chunk
    if (chunks > most) chunks = most;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}

extern "C" int vibo_decoder_fwd_bwd(const vibo_decoder_desc* d, const float* response, const uint8_t* mask,
                                    const float* U, const float* V, const float* L, const float* guess, const float* w1,
                                    const float* W2, const float* b2, const float* w3, const float* b3,
                                    float* ll_part, float* dU_part, float* dV_part, float* dL, float* dguess_part,
                                    float* dW2_part, float* dvec_part, float* prob_out, void* stream) {
    const int rc = decoder_check(d);
    if (rc) return rc;
    if (!response || !V || !W2 || !b2 || !w3 || !b3 || !ll_part) return -5;
    const bool hasl = L != nullptr;
    if ((w1 || d->resid != 0.f) && !hasl) return -5;
    if (d->want_grad && (!dW2_part || !dvec_part || !dV_part || (hasl && !dL) || (U && !dU_part) || (guess && !dguess_part))) return -5;
    if ((((uintptr_t)V) | ((uintptr_t)U)) & 15) return -4;
    DecParams p;
    memset(&p, 0, sizeof(p));
    p.response = response; p.mask = mask; p.resp_stride = d->response_row_stride; p.mask_stride = d->mask_row_stride;
    p.U = U; p.V = V; p.L = L; p.guess = guess; p.w1 = w1; p.W2 = W2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.resid = d->resid;
    p.B = d->num_person; p.I = d->num_item;
    p.ppc = (int)((((long long)d->num_person + d->person_chunks - 1) / d->person_chunks + 1) & ~1ll);    // even: persons go in pairs
    p.ll_part = ll_part; p.dU_part = dU_part; p.dV_part = dV_part; p.dL = dL; p.dguess_part = dguess_part;
    p.dW2_part = dW2_part; p.dvec_part = dvec_part; p.prob_out = prob_out;
    const dim3 grid((unsigned)((d->num_item + 63) / 64), (unsigned)d->person_chunks), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (d->want_grad) {
        if (hasl) hipLaunchKernelGGL((decoder_kernel<true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((decoder_kernel<true, false>), grid, block, 0, s, p);
    } else {
        if (hasl) hipLaunchKernelGGL((decoder_kernel<false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((decoder_kernel<false, false>), grid, block, 0, s, p);
    }
    return (int)hipGetLastError();
}
