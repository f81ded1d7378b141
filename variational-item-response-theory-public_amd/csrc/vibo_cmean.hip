// vibo_cmean.hip -- the two dense contractions of the VIBO encoder over a person's observed cells, on the matrix pipe.
//
// With --conditional-posterior (models.py:664-710) every per-term quantity of the encoder depends on the response code AND the
// item, X[c, i, :], and a person needs its sum over the observed cells; the transposed sum is the gradient:
//
//     S[p, :]     = sum_i [cell (p, i) observed] X[code_pi, i, :]        = onehot(codes) [B, 2I] x X [2I, N]
//     dX[c, i, :] = sum_p [code_pi == c] G[p, :]                          = onehot(codes)^T [2I, B] x G [B, N]
//
// Two users:
//   * --ability-merge mean (_forward_mean :631-650): X = elu(mlp1([c, item_i])) in R^64, G = the upstream gradient of the
//     row sums  (N = 64; exported as vibo_code_table_sum_forward / _backward);
//   * --ability-merge product (the product of experts of the conditional posterior, models.py:664-710 + utils.py:105-113):
//     X = [tau | mu tau] with tau = 1/(exp(logvar) + eps) (N = 2 A <= 16: the precision and the precision-weighted mean of a
//     person's experts, plus the observed count), G = the ELBO kernel's per-person coefficients [head][P1 | P2][dim]
//     (N = 4 A <= 32), from which d/d mu = S1 tau, d/d logvar = -(S1 mu + S2) tau^2 exp(logvar)  (launch_cond_pre_mfma /
//     launch_cond_post_mfma, called by vibo_capi.hip around the row-split kernel; round 2 ran these as VALU kernels with one
//     register accumulator per (item, code, coefficient): two launches each at ability_dim 5..8, 4.8 ms at 1M x 1k A = 8).
//
// Both run on v_mfma_f32_16x16x32_bf16 straight from the 1-byte cell codes (0 wrong / 1 right / 2 missing).  The one-hot
// operand is exact (entries 2.0 = bf16 0x4000, a byte permute away from the code bits); the dense operand goes in as THREE
// bf16 pieces hi + mid + lo of 0.5 x (truncation: 8 + 8 + 8 significant bits, |x - pieces| <= 2^-24 |x|) with fp32
// accumulation.  bf16 has fp32's exponent range, so no operand rescaling and no range fault: an Inf / NaN input stays
// Inf / NaN in the output (x - hi is NaN then).
//
// Operand layouts (16x16x32: A lane (row = lane & 15, g = lane >> 4) holds k = 8 g .. 8 g + 7; B lane (col = lane & 15, g)
// likewise; D lane (col = lane & 15, g) holds rows 4 g .. 4 g + 3):
//   * K order of the forward, per super-step S of 64 items: K-step j in 0..3, lane group g, kk in 0..7 <-> item
//     64 S + 16 g + 4 j + (kk >> 1), code kk & 1 -- so that a lane's four K-steps are the four dwords of ONE 16-byte load of
//     its person's code row (a wave reads 16 rows x 64 contiguous bytes per load);
//   * the dense operands are pre-arranged once per call into "images" whose 16-byte lane pieces are contiguous per wave load;
//   * the backward's one-hot^T operand comes out of an LDS tile [64 persons][128 (item, code) columns] by
//     ds_read_b64_tr_b16 (the 4 x 16 block of a 16-lane group is 4 persons x 16 columns; a lane receives its column's 4 persons).
// NT = N-tiles of 16 output columns (1, 2 or 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_cond.hpp"
#include "vibo_cond_finalize.hpp"
#include <string.h>

namespace vibo {

typedef __bf16 cm_bf8 __attribute__((ext_vector_type(8)));
typedef float cm_f32x4 __attribute__((ext_vector_type(4)));
typedef short cm_s4 __attribute__((ext_vector_type(4)));
typedef unsigned int cm_u4 __attribute__((ext_vector_type(4)));
constexpr int kCmH = 64;
constexpr int kCmNP = 3;                // bf16 pieces per fp32 value

// 8 fp32 values (x 0.5: the one-hot entries are 2.0) -> three bf16 piece vectors, element kk in half kk
__device__ __forceinline__ void cm_split3(const float (&v)[8], uint4& hi, uint4& mid, uint4& lo) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const float x = 0.5f * v[kk];
        h[kk] = __float_as_uint(x) & 0xffff0000u;
        const float r = x - __uint_as_float(h[kk]);
        m[kk] = __float_as_uint(r) & 0xffff0000u;
        const float r2 = r - __uint_as_float(m[kk]);
        l[kk] = __float_as_uint(r2) & 0xffff0000u;
    }
    hi = uint4{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3], (h[4] >> 16) | h[5], (h[6] >> 16) | h[7]};
    mid = uint4{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3], (m[4] >> 16) | m[5], (m[6] >> 16) | m[7]};
    lo = uint4{(l[0] >> 16) | l[1], (l[2] >> 16) | l[3], (l[4] >> 16) | l[5], (l[6] >> 16) | l[7]};
}
// 4 cell codes (one dword) -> the 8 one-hot bf16 [c0 == 0, c0 == 1, c1 == 0, c1 == 1, ...] with entries 2.0 (0x4000).
// The code bytes themselves are the byte selectors of v_perm_b32 (selector 0..3 = bytes of the second source): two table
// look-ups give the 0x40 high bytes of the "is 0" / "is 1" halfs, four more permutes lay them out.
__device__ __forceinline__ uint4 cm_onehot(const uint32_t w) {
    const uint32_t t0 = __builtin_amdgcn_perm(0u, 0x00000040u, w);      // bytes 0x40 where the cell is a 0   (code 2 -> 0x00)
    const uint32_t t1 = __builtin_amdgcn_perm(0u, 0x00004000u, w);      // ... where it is a 1
    // cell b -> dword [0x00, t0.b, 0x00, t1.b]  (perm: S0 = t1 -> bytes 4..7, S1 = t0 -> bytes 0..3, 0x0c = constant 0)
    return uint4{__builtin_amdgcn_perm(t1, t0, 0x040c000cu), __builtin_amdgcn_perm(t1, t0, 0x050c010cu),
                 __builtin_amdgcn_perm(t1, t0, 0x060c020cu), __builtin_amdgcn_perm(t1, t0, 0x070c030cu)};
}
__device__ __forceinline__ cm_f32x4 cm_mfma(const uint4 a, const uint4 b, const cm_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cm_bf8, a), __builtin_bit_cast(cm_bf8, b), c, 0, 0, 0);
}

// ---- images -----------------------------------------------------------------------------------------------------------
// forward B operand: img[(((S * 4 + j) * NT + nt) * 3 + piece) * 64 + lane] = 8 pieces over kk of X[kk & 1][item][16 nt + (lane & 15)]
// COND: X is built from the encoder table [2][I][2A] (mu | logvar): columns [0, A) tau, [A, 2A) mu tau (utils.py:105-113)
template <bool COND>
__global__ __launch_bounds__(256) void cm_table_image_kernel(const float* __restrict__ src, uint4* __restrict__ img, int I, int nS, int NT,
                                                             int ncols) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nS * 4 * NT * 64) return;
    const int lane = t & 63, nt = (t >> 6) % NT, j = ((t >> 6) / NT) & 3, S = (t >> 6) / (NT * 4);
    const int n = 16 * nt + (lane & 15), g = lane >> 4;
    float v[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int item = 64 * S + 16 * g + 4 * j + (kk >> 1), c = kk & 1;
        float x = 0.f;
        if (item < I && n < ncols) {
            if constexpr (COND) {                     // ncols = 2 A, or 2 A + 1 with a column of ones (-> the observed count)
                const int A = ncols >> 1, a = n < A ? n : n - A;
                if (n == 2 * A) x = 1.0f;
                else {
                    const float* te = src + ((size_t)c * I + item) * 2 * A;
                    const float tau = 1.0f / (expf(te[A + a]) + kPoeEps);
                    x = n < A ? tau : te[a] * tau;
                }
            } else {
                x = src[((size_t)c * I + item) * ncols + n];
            }
        }
        v[kk] = x;
    }
    uint4 hi, mid, lo;
    cm_split3(v, hi, mid, lo);
    const size_t o = ((size_t)((S * 4 + j) * NT + nt) * kCmNP) * 64 + lane;
    img[o] = hi;
    img[o + 64] = mid;
    img[o + 128] = lo;
}
// backward B operand: gimg[((c32 * NT + nt) * 3 + piece) * 64 + lane] = 8 pieces over kk of G[32 c32 + 8 g + kk][16 nt + (lane & 15)]
// (G row stride = ncols; columns >= ncols are zero)
__global__ __launch_bounds__(256) void cm_grad_image_kernel(const float* __restrict__ G, uint4* __restrict__ gimg, long long B, long long n32,
                                                            int NT, int ncols) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n32 * NT * 64) return;
    const int lane = (int)(t & 63), nt = (int)((t >> 6) % NT);
    const long long c32 = (t >> 6) / NT;
    const int n = 16 * nt + (lane & 15), g = lane >> 4;
    float v[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const long long p = 32 * c32 + 8 * g + kk;
        v[kk] = (p < B && n < ncols) ? G[p * ncols + n] : 0.f;
    }
    uint4 hi, mid, lo;
    cm_split3(v, hi, mid, lo);
    const size_t o = ((size_t)(c32 * NT + nt) * kCmNP) * 64 + lane;
    gimg[o] = hi;
    gimg[o + 64] = mid;
    gimg[o + 128] = lo;
}

// the packed form (cm_backward_body's PK): gimg[c32 * 64 + lane] = piece (lane & 15) / ncols of column (lane & 15) % ncols (3 ncols <= 16)
__global__ __launch_bounds__(256) void cm_grad_image_packed_kernel(const float* __restrict__ G, uint4* __restrict__ gimg, long long B, long long n32,
                                                                   int ncols) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n32 * 64) return;
    const int lane = (int)(t & 63);
    const long long c32 = t >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int piece = n / ncols, col = n % ncols;
    float v[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const long long p = 32 * c32 + 8 * g + kk;
        v[kk] = (p < B && piece < kCmNP) ? G[p * ncols + col] : 0.f;
    }
    uint4 hi, mid, lo;
    cm_split3(v, hi, mid, lo);
    gimg[(size_t)c32 * 64 + lane] = piece == 0 ? hi : piece == 1 ? mid : lo;
}

// the 16 code bytes at items [i0, i0 + 16) of the row at `rp`.  AL: rows 16-byte aligned (one load), else 4-byte aligned (four).
// cm_load_full: the 16 bytes lie inside the row (every step below I / 64: no bounds logic anywhere near the pipelined loads);
// cm_load_tail: the row's last, partial step -- dword by dword (rows are whole dwords), bytes past the row's end read as missing (2)
template <bool AL>
__device__ __forceinline__ uint4 cm_load_full(const uint8_t* __restrict__ rp, int i0) {
    if constexpr (AL) return *reinterpret_cast<const uint4*>(rp + i0);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(rp + i0);
    return uint4{q[0], q[1], q[2], q[3]};
}
// cm_load_padded: the same partial step when the rows are padded to whole 64-byte steps (the library's own code rows, and
// pack_cell_codes' from 256 items): one full load, the bytes past the row's end forced to missing
template <bool AL>
__device__ __forceinline__ uint4 cm_load_padded(const uint8_t* __restrict__ rp, int I, int i0) {
    uint4 w = cm_load_full<AL>(rp, i0);
    uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int left = I - (i0 + 4 * k);
        const uint32_t keep = left >= 4 ? 0xffffffffu : left <= 0 ? 0u : (1u << (8 * left)) - 1u;
        d[k] = (d[k] & keep) | (0x02020202u & ~keep);
    }
    return uint4{d[0], d[1], d[2], d[3]};
}
__device__ __forceinline__ uint4 cm_load_tail(const uint8_t* __restrict__ rp, int I, int i0) {
    uint32_t d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int left = I - (i0 + 4 * k);
        uint32_t v = 0x02020202u;
        if (left > 0) v = *reinterpret_cast<const uint32_t*>(rp + i0 + 4 * k);
        if (left < 4) {
            const uint32_t keep = left <= 0 ? 0u : (1u << (8 * left)) - 1u;
            v = (v & keep) | (0x02020202u & ~keep);
        }
        d[k] = v;
    }
    return uint4{d[0], d[1], d[2], d[3]};
}
// row of person p (clamped to the last person: what a wave reads for persons that do not exist never reaches an output --
// the forward does not store their rows, the backward multiplies them by zero coefficients)
__device__ __forceinline__ const uint8_t* cm_row(const uint8_t* __restrict__ codes, long long stride, const int64_t* __restrict__ row_index,
                                                 long long B, long long p) {
    p = p < B ? p : B - 1;
    return codes + (row_index ? row_index[p] : p) * stride;
}
__device__ __forceinline__ int cm_observed(const uint4 w) {
    return __popc(~(w.x >> 1) & 0x01010101u) + __popc(~(w.y >> 1) & 0x01010101u) + __popc(~(w.z >> 1) & 0x01010101u) +
           __popc(~(w.w >> 1) & 0x01010101u);
}
template <int J>
__device__ __forceinline__ uint32_t cm_dword(const uint4 w) { return J == 0 ? w.x : J == 1 ? w.y : J == 2 ? w.z : w.w; }

// ---- forward: S[p][:] = sum_i onehot . X ; a wave owns 64 persons (4 M-tiles) x all 16 NT columns ------------------------
// out[p * out_stride + n] for n < ncols; COUNT: out[p * out_stride + ncols] = observed cells of the row.
// A workgroup is 4 compute waves + 1 producer wave.  Vector-memory loads return in order, so a wave that waits for a
// (short, L2-resident) dense-operand load also waits for every code-row load it issued before it -- which would expose the
// code rows' whole HBM latency at every step.  Hence the split: the producer wave streams the dense operand into LDS in 12 KB
// chunks (4 / NT K-steps x NT N-tiles x 3 pieces, double-buffered, one barrier per chunk); the only vector-memory loads a
// compute wave ever waits on are its own code rows, issued two 64-item steps ahead (three register sets, rotated by name).
// (one N-tile: 4 waves per SIMD = 3 workgroups per CU; the register cap costs 4 spilled registers and wins 10 %: 261 -> 235 us per GB of codes)
template <int NT, bool COUNT, bool AL>
__global__ __launch_bounds__(320, NT == 1 ? 4 : 1) void cm_forward_kernel(const uint8_t* __restrict__ codes, long long stride, const int64_t* __restrict__ row_index,
                                                         long long B, int I, int nS, const uint4* __restrict__ img, float* __restrict__ out,
                                                         int out_stride, int ncols) {
    constexpr int JC = 4 / NT;                      // K-steps per chunk
    __shared__ uint4 bimg[2][768];
    __shared__ uint4 wtail[4][4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nQ = nS * NT;
    if (wv == 4) {                                   // ---- producer wave
        const cm_u4* src = reinterpret_cast<const cm_u4*>(img) + lane;
        cm_u4* lds = reinterpret_cast<cm_u4*>(&bimg[0][0]) + lane;
        cm_u4 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11;
#define CM_CHUNK_LOAD(np) { r0 = (np)[0]; r1 = (np)[64]; r2 = (np)[128]; r3 = (np)[192]; r4 = (np)[256]; r5 = (np)[320]; \
                            r6 = (np)[384]; r7 = (np)[448]; r8 = (np)[512]; r9 = (np)[576]; r10 = (np)[640]; r11 = (np)[704]; }
#define CM_CHUNK_STORE(dp) { (dp)[0] = r0; (dp)[64] = r1; (dp)[128] = r2; (dp)[192] = r3; (dp)[256] = r4; (dp)[320] = r5; \
                             (dp)[384] = r6; (dp)[448] = r7; (dp)[512] = r8; (dp)[576] = r9; (dp)[640] = r10; (dp)[704] = r11; }
        CM_CHUNK_LOAD(src)
        CM_CHUNK_STORE(lds)
        CM_CHUNK_LOAD(src + (size_t)(1 < nQ ? 1 : 0) * 768)
        __syncthreads();
        for (int q = 0; q < nQ; ++q) {
            cm_u4* dst = lds + ((q + 1) & 1) * 768;
            CM_CHUNK_STORE(dst)                                           // chunk q + 1 (fetched while chunk q - 1 was being used)
            const cm_u4* np = src + (size_t)(q + 2 < nQ ? q + 2 : nQ - 1) * 768;
            CM_CHUNK_LOAD(np)
            __syncthreads();
        }
#undef CM_CHUNK_LOAD
#undef CM_CHUNK_STORE
        return;
    }
    const int m = lane & 15, g = lane >> 4;
    const long long p0 = ((long long)blockIdx.x * 4 + wv) * 64;
    // Loads: lane l fetches piece l & 3 of row l >> 2 of the M-tile (4 adjacent lanes = 64 contiguous bytes: one cache access
    // instead of four); a lane permutation then hands lane (m, g) the operand layout's piece g of row m.
    const int lrow = lane >> 2, lpiece = lane & 3;
    const int perm_addr = 4 * (4 * m + g);
    const uint8_t* rp[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) rp[mt] = cm_row(codes, stride, row_index, B, p0 + 16 * mt + lrow);
    cm_f32x4 acc[4][NT];
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = cm_f32x4{0.f, 0.f, 0.f, 0.f};
    const int nFull = I >> 6;                        // steps whose 64 items all exist; one partial step may follow
    uint4 w0[4], w1[4], w2[4];
    // code rows of step S (clamped to the last whole step: the final prefetches re-read it)
    auto fetch = [&](uint4 (&w)[4], const int S) {
        const int i0 = 64 * (S < nFull ? S : nFull - 1) + 16 * lpiece;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) w[mt] = cm_load_full<AL>(rp[mt], i0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // the partial last step's code words wait in LDS (wave-private: 16 registers less through the whole loop)
    if (nS > nFull) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) wtail[wv][mt][lane] = cm_load_tail(rp[mt], I, 64 * nFull + 16 * lpiece);
    }
    if (nFull > 0) {
        fetch(w0, 0);
        fetch(w1, 1);
    }
    __syncthreads();
    // one 64-item step on the code words w: NT chunks of the dense operand
    auto step = [&](const int S, const uint4 (&wl)[4]) {
        uint4 w[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            w[mt] = uint4{(uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr, (int)wl[mt].x), (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr, (int)wl[mt].y),
                          (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr, (int)wl[mt].z), (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr, (int)wl[mt].w)};
#pragma unroll
        for (int jc = 0; jc < NT; ++jc) {
            const int cur = (S * NT + jc) & 1;
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                uint4 a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int j = jc * JC + jj;
                    a[mt] = cm_onehot(j == 0 ? w[mt].x : j == 1 ? w[mt].y : j == 2 ? w[mt].z : w[mt].w);
                }
                const uint4* bp = &bimg[cur][(jj * NT) * kCmNP * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const uint4 b0 = bp[(nt * kCmNP) * 64], b1 = bp[(nt * kCmNP + 1) * 64], b2 = bp[(nt * kCmNP + 2) * 64];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[mt][nt] = cm_mfma(a[mt], b2, acc[mt][nt]);        // smallest pieces first
                        acc[mt][nt] = cm_mfma(a[mt], b1, acc[mt][nt]);
                        acc[mt][nt] = cm_mfma(a[mt], b0, acc[mt][nt]);
                    }
                }
            }
            __syncthreads();
        }
        if constexpr (COUNT) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) cnt[mt] += cm_observed(w[mt]);
        }
    };
    int S = 0;
    for (; S + 3 <= nFull; S += 3) {
        fetch(w2, S + 2); step(S, w0);
        fetch(w0, S + 3); step(S + 1, w1);
        fetch(w1, S + 4); step(S + 2, w2);
    }
    if (S < nFull) {
        step(S, w0);
        if (S + 1 < nFull) step(S + 1, w1);
    }
    if (nS > nFull) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) w0[mt] = wtail[wv][mt][lane];
        step(nFull, w0);
    }
    // D: lane (col n = m, g) holds rows 4 g + jj of the M-tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const long long p = p0 + 16 * mt + 4 * g + jj;
            if (p < B) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (16 * nt + m < ncols) out[p * out_stride + 16 * nt + m] = acc[mt][nt][jj];
            }
        }
        if constexpr (COUNT) {
            int t = cnt[mt];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            const long long p = p0 + 16 * mt + m;
            if (g == 0 && p < B) out[p * out_stride + ncols] = (float)t;
        }
    }
}

// ---- the same forward straight from fp32 rows (+ u8 mask): count-and-emit inside the contraction ---------------------------
// The conditional posterior on fp32 rows used to run a pure re-pack stream (row_count_kernel: 5 B in, 1 B out per cell, 1.26 ms
// per 1M x 1k) in front of the forward contraction (0.29 ms).  Here the compute waves load the fp32 cells themselves, turn each
// 16-cell piece into the 16 code bytes the contraction wants (cell_codes4), leave them behind as the row's Format P codes for the
// passes that follow (minibatch order) and go on as cm_forward_kernel<1> does.  One N-tile only (2A + 1 <= 16 columns; 8 dims
// count with the code words).  A compute wave owns 32 persons (two M-tiles) and keeps three steps of raw cells in flight
// (3 x 40 registers: at five waves per workgroup a wave has 256).
// (Round 6: two steps in flight at 136 registers = TWO workgroups per CU measured the same 1.38 ms per 1M x 1k x 8 dims as three steps
//  and one workgroup (1.36): the pass is not latency-bound per CU.  It moves 5 GB in + 1 GB out at 4.4 TB/s where the VALU
//  count-and-emit pass -- whole rows per wave -- reaches 5.4: a wave here touches 16 rows x 256 bytes per step.)
struct CmRawPiece {        // the 16 cells at items [i0, i0 + 16) of one row
    float4 x[4];
    uint4 m;
};
// Lane layout of the raw cells (round 6).  A lane's four float4 used to be 64 contiguous bytes of its row: per load instruction the
// four lanes of a row then touched four different 64-byte blocks -- 64 blocks per wave instruction, a quarter of each used, and
// the address path of the texture unit handles one 64-byte block per quad and cycle (cm_forward_fp32 streamed 3.6 TB/s where the
// VALU count-and-emit pass of the same bytes runs 5.4).  Now load k of lane (row, piece) reads the 16 bytes at 64 k + 16 piece of
// the step's 256-byte row segment: the four lanes of a row share ONE contiguous 64-byte block per instruction.  The lane then
// holds cells 16 k + 4 piece .. + 3 (k = 0..3) instead of 16 piece .. 16 piece + 15: the four "answered right" byte words are
// transposed inside the quad (two DPP exchange stages) back to the contiguous layout -- where the mask bytes (one 16-byte
// load, already contiguous per row), the emitted code row and the operand permutation expect them.
__device__ __forceinline__ uint32_t cm_answer4(const float4 x) {         // byte b = byte 3 of cell b (0x3F: 1.0, 0x00: 0.0)
    const uint32_t x0 = __builtin_bit_cast(uint32_t, x.x), x1 = __builtin_bit_cast(uint32_t, x.y);
    const uint32_t x2 = __builtin_bit_cast(uint32_t, x.z), x3 = __builtin_bit_cast(uint32_t, x.w);
    return __builtin_amdgcn_perm(x1, x0, 0x0c0c0703u) | __builtin_amdgcn_perm(x3, x2, 0x07030c0cu);
}
// 4 x 4 transpose of dwords inside every quad of lanes: lane p, register k  <->  lane k, register p
__device__ __forceinline__ void cm_quad_transpose4(uint32_t (&a)[4], const bool bit1, const bool bit0) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {                    // 2 x 2 blocks across lanes p ^ 2
        const uint32_t send = bit1 ? a[k] : a[k + 2];
        const uint32_t recv = (uint32_t)dpp_i<0x4e>((int)send);       // quad_perm [2,3,0,1]
        a[k] = bit1 ? recv : a[k];
        a[k + 2] = bit1 ? a[k + 2] : recv;
    }
#pragma unroll
    for (int k = 0; k < 4; k += 2) {                 // inside the blocks across lanes p ^ 1
        const uint32_t send = bit0 ? a[k] : a[k + 1];
        const uint32_t recv = (uint32_t)dpp_i<0xb1>((int)send);       // quad_perm [1,0,3,2]
        a[k] = bit0 ? recv : a[k];
        a[k + 1] = bit0 ? a[k + 1] : recv;
    }
}
// -> the lane's 16 CONTIGUOUS cell codes (items 16 piece .. 16 piece + 15 of the step; 0 wrong / 1 right / 2 missing: cell_codes4)
__device__ __forceinline__ uint4 cm_codes_of(const CmRawPiece& r, const bool bit1, const bool bit0) {
    uint32_t hb[4] = {cm_answer4(r.x[0]), cm_answer4(r.x[1]), cm_answer4(r.x[2]), cm_answer4(r.x[3])};
    cm_quad_transpose4(hb, bit1, bit0);
    const uint32_t mm[4] = {r.m.x, r.m.y, r.m.z, r.m.w};
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = (hb[k] & mm[k] & 0x01010101u) | ((mm[k] ^ 0x01010101u) << 1);
    return uint4{c[0], c[1], c[2], c[3]};
}
constexpr int kCmF32MT = 2;            // M-tiles (16 persons) per compute wave of cm_forward_fp32_kernel
template <bool COUNT, bool MAL /* mask rows 16-byte aligned */>
__global__ __launch_bounds__(320, 1) void cm_forward_fp32_kernel(const float* __restrict__ response, const uint8_t* __restrict__ mask,
                                                                 long long resp_stride, long long mask_stride,
                                                                 const int64_t* __restrict__ row_index, long long B, int I, int nS,
                                                                 const uint4* __restrict__ img, float* __restrict__ out, int out_stride,
                                                                 int ncols, uint8_t* __restrict__ codes_out, long long codes_stride) {
    __shared__ uint4 bimg[2][768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wv == 4) {                                   // ---- producer wave (as in cm_forward_kernel<1>: one 12 KB chunk per step)
        const cm_u4* src = reinterpret_cast<const cm_u4*>(img) + lane;
        cm_u4* lds = reinterpret_cast<cm_u4*>(&bimg[0][0]) + lane;
        cm_u4 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11;
#define CM_CHUNK_LOAD(np) { r0 = (np)[0]; r1 = (np)[64]; r2 = (np)[128]; r3 = (np)[192]; r4 = (np)[256]; r5 = (np)[320]; \
                            r6 = (np)[384]; r7 = (np)[448]; r8 = (np)[512]; r9 = (np)[576]; r10 = (np)[640]; r11 = (np)[704]; }
#define CM_CHUNK_STORE(dp) { (dp)[0] = r0; (dp)[64] = r1; (dp)[128] = r2; (dp)[192] = r3; (dp)[256] = r4; (dp)[320] = r5; \
                             (dp)[384] = r6; (dp)[448] = r7; (dp)[512] = r8; (dp)[576] = r9; (dp)[640] = r10; (dp)[704] = r11; }
        CM_CHUNK_LOAD(src)
        CM_CHUNK_STORE(lds)
        CM_CHUNK_LOAD(src + (size_t)(1 < nS ? 1 : 0) * 768)
        __syncthreads();
        for (int q = 0; q < nS; ++q) {
            cm_u4* dst = lds + ((q + 1) & 1) * 768;
            CM_CHUNK_STORE(dst)
            const cm_u4* np = src + (size_t)(q + 2 < nS ? q + 2 : nS - 1) * 768;
            CM_CHUNK_LOAD(np)
            __syncthreads();
        }
#undef CM_CHUNK_LOAD
#undef CM_CHUNK_STORE
        return;
    }
    const int m = lane & 15, g = lane >> 4;
    constexpr int MT = kCmF32MT;
    const long long p0 = ((long long)blockIdx.x * 4 + wv) * (16 * MT);
    const int lrow = lane >> 2, lpiece = lane & 3;
    const int perm_addr = 4 * (4 * m + g);
    const float* rpx[MT];
    const uint8_t* rpm[MT];
    uint8_t* rpc[MT];                                 // nullptr: the row does not exist (nothing is stored for it)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long p = p0 + 16 * mt + lrow;
        const long long pc = p < B ? p : B - 1;
        const long long src = row_index ? row_index[pc] : pc;
        rpx[mt] = response + src * resp_stride;
        rpm[mt] = mask ? mask + src * mask_stride : nullptr;
        rpc[mt] = (codes_out && p < B) ? codes_out + p * codes_stride : nullptr;
    }
    cm_f32x4 acc[MT];
    int cnt[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cnt[mt] = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = cm_f32x4{0.f, 0.f, 0.f, 0.f};
    const int nFull = I >> 6;                        // steps whose 64 items all exist; one partial step may follow
    auto fetch = [&](CmRawPiece (&w)[MT], const int S) {
        const int i0 = 64 * (S < nFull ? S : nFull - 1) + 16 * lpiece;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // (load k: the 16 bytes at 64 k + 16 piece of the step's 256-byte row segment -- see cm_codes_of)
            const float4* xp = reinterpret_cast<const float4*>(rpx[mt] + (i0 - 12 * lpiece));
            w[mt].x[0] = xp[0]; w[mt].x[1] = xp[4]; w[mt].x[2] = xp[8]; w[mt].x[3] = xp[12];
            if (mask) {
                if constexpr (MAL) w[mt].m = *reinterpret_cast<const uint4*>(rpm[mt] + i0);
                else {
                    const uint32_t* q = reinterpret_cast<const uint32_t*>(rpm[mt] + i0);
                    w[mt].m = uint4{q[0], q[1], q[2], q[3]};
                }
            } else {
                w[mt].m = uint4{0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // one 64-item step on the code words wl (this lane's pieces: stored as the rows' codes, then permuted into the operand layout)
    auto step = [&](const int S, const uint4 (&wl)[MT]) {
        const int i0 = 64 * S + 16 * lpiece;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (rpc[mt]) *reinterpret_cast<uint4*>(rpc[mt] + i0) = wl[mt];
        uint4 w[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            w[mt] = uint4{(uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr, (int)wl[mt].x), (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr, (int)wl[mt].y),
                          (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr, (int)wl[mt].z), (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr, (int)wl[mt].w)};
        const int cur = S & 1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            uint4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = cm_onehot(jj == 0 ? w[mt].x : jj == 1 ? w[mt].y : jj == 2 ? w[mt].z : w[mt].w);
            const uint4* bp = &bimg[cur][jj * kCmNP * 64 + lane];
            const uint4 b0 = bp[0], b1 = bp[64], b2 = bp[128];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = cm_mfma(a[mt], b2, acc[mt]);        // smallest pieces first
                acc[mt] = cm_mfma(a[mt], b1, acc[mt]);
                acc[mt] = cm_mfma(a[mt], b0, acc[mt]);
            }
        }
        __syncthreads();
        if constexpr (COUNT) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) cnt[mt] += cm_observed(w[mt]);
        }
    };
    // three steps of raw cells in flight (3 x 40 registers); a set is converted when its step comes up
    CmRawPiece rA[MT], rB[MT], rC[MT];
    if (nFull > 0) {
        fetch(rA, 0);
        fetch(rB, 1);
    }
    __syncthreads();
    auto run = [&](const int S, const CmRawPiece (&raw)[MT]) {
        uint4 wl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wl[mt] = cm_codes_of(raw[mt], (lane & 2) != 0, (lane & 1) != 0);
        step(S, wl);
    };
    int S = 0;
    for (; S + 3 <= nFull; S += 3) {
        fetch(rC, S + 2); run(S, rA);
        fetch(rA, S + 3); run(S + 1, rB);
        fetch(rB, S + 4); run(S + 2, rC);
    }
    if (S < nFull) {
        run(S, rA);
        if (S + 1 < nFull) run(S + 1, rB);
    }
    if (nS > nFull) {                                // the row's last, partial step: 4 cells at a time, cells past the row's end missing
        uint4 wl[MT];
        const int i0 = 64 * nFull + 16 * lpiece;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            uint32_t d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int left = I - (i0 + 4 * k);
                uint32_t v = kAllMissing4;
                if (left > 0) {                      // (rows are whole 4-cell chunks: the chunk is inside the row's stride)
                    const float4 x = *reinterpret_cast<const float4*>(rpx[mt] + i0 + 4 * k);
                    uint32_t mm = mask ? *reinterpret_cast<const uint32_t*>(rpm[mt] + i0 + 4 * k) : 0x01010101u;
                    if (left < 4) mm &= (1u << (8 * left)) - 1u;
                    v = cell_codes4(x, mm);
                }
                d[k] = v;
            }
            wl[mt] = uint4{d[0], d[1], d[2], d[3]};
        }
        step(nFull, wl);
    }
    // D: lane (col n = m, g) holds rows 4 g + jj of the M-tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const long long p = p0 + 16 * mt + 4 * g + jj;
            if (p < B && m < ncols) out[p * out_stride + m] = acc[mt][jj];
        }
        if constexpr (COUNT) {
            int t = cnt[mt];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            const long long p = p0 + 16 * mt + m;
            if (g == 0 && p < B) out[p * out_stride + ncols] = (float)t;
        }
    }
}

// ---- backward: dX = onehot^T G ; a wave (= workgroup) owns a stripe of 64 items x a range of persons x NT N-tiles -------
// One-hot tile [64 persons][128 (item, code) columns] of bf16, 256-byte rows, XOR-swizzled so that neither the 16-byte writes
// of a code row's pieces nor the transposed 8-byte reads of a 16-lane group (4 persons x 32 bytes) share banks:
//   byte offset in the row = ((unit ^ swz(person)) << 5) | ((half ^ person bit 1) << 4) | byte,   unit = 32-byte unit (16 columns),
//   swz(person) = (person & 3) | (person bit 3) << 2
// Loads and tile writes: lane l handles piece l & 3 (16 bytes = 16 items) of row l >> 2 of each 16-person M-tile -- 4 adjacent
// lanes fetch 64 contiguous bytes (one cache access instead of four).
// Per 64-person step the coefficient operands are fetched first and the code rows of the step after next behind them (loads
// return in order: what is waited for must have been issued before what may stay in flight).  NT is 1, 2 or 4 (the 64 hidden
// units of --ability-merge mean); nt0 = NT blockIdx.y of ntot.
__device__ __forceinline__ uint2 cm_lds_tr(const char* p) {
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((cm_s4 __attribute__((address_space(3)))*)p));
}
__device__ __forceinline__ int cm_tile_swz(int person) { return (person & 3) | (((person >> 3) & 1) << 2); }
// WV = waves per workgroup sharing the tile (1, or 2: each wave builds two of the four M-tiles and owns four of the eight column
// tiles -- half the accumulators and code-row registers per wave, 8 KB of LDS per wave: 4 waves per SIMD instead of 2.5)
// PK (round 6): the three bf16 pieces of the coefficients sit in three column groups of ONE operand (columns piece * ncols + col; the
// conditional posterior at ability_dim 1 has 4 coefficient columns: 12 of 16) -- one MFMA per (column tile, K-chunk) instead of three,
// the finalize adds the groups (cm_cond_finalize_body, CondFinTail::packed_cols).
template <int NT, bool AL, int TAIL /* 0: whole step, 1: partial step of padded rows, 2: partial step */, bool GATHER, int WV, int PS /* persons per step: 64 or 32 */, bool PK = false>
__device__ __forceinline__ void cm_backward_body(const uint8_t* __restrict__ codes, long long stride, const int64_t* __restrict__ row_index,
                                                 long long B, int I, const uint4* __restrict__ gimg, float* __restrict__ rec, int nR,
                                                 long long per_r, char* tile, int S, int r, int ntot, int nt0) {
    constexpr int MTW = (PS / 16) / WV, TW = 8 / WV;   // M-tiles built / column tiles owned per wave
    constexpr int NC = PS / 32;                        // 32-person K-chunks per step
    const int lane = threadIdx.x & 63;
    const int wv = WV > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int m = lane & 15, g = lane >> 4;
    const int lrow = lane >> 2, lpiece = lane & 3;
    const int i0 = 64 * S + 16 * lpiece;
    const long long pa = (long long)r * per_r, pb = pa + per_r < B ? pa + per_r : B;        // per_r is a multiple of 64
    cm_f32x4 acc[TW][NT];
#pragma unroll
    for (int T = 0; T < TW; ++T)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[T][nt] = cm_f32x4{0.f, 0.f, 0.f, 0.f};
    // code rows of the 64 persons from p (persons past the last one read the last one's row: their coefficients are zero)
    auto fetch = [&](uint4 (&w)[MTW], const long long p) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            long long q = p + 16 * (MTW * wv + mt) + lrow;
            q = q < B ? q : B - 1;
            if constexpr (GATHER) q = row_index[q];
            const uint8_t* rp = codes + q * stride;
            if constexpr (TAIL == 0) w[mt] = cm_load_full<AL>(rp, i0);
            else if constexpr (TAIL == 1) w[mt] = cm_load_padded<AL>(rp, I, i0);
            else w[mt] = cm_load_tail(rp, I, i0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // write side: lane l of M-tile mt owns person 16 mt + (l >> 2), columns 32 (l & 3) .. + 31 = units 2 (l & 3), + 1 (4 x 16 bytes)
    int wofs[MTW][4];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int person = 16 * (MTW * wv + mt) + lrow;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wofs[mt][j] = person * 256 + ((((2 * lpiece + (j >> 1)) ^ cm_tile_swz(person)) << 5) | ((((j & 1) ^ (person >> 1)) & 1) << 4));
    }
    // read side: lane i of a 16-lane group addresses person 8 g + (i >> 2) (+ 4 for the second read, + 32 c), 8-byte piece i & 3 of
    // the unit; the unit index T is XORed in per read
    int rofs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int person = 8 * g + (m >> 2) + 4 * h;
        rofs[h] = person * 256 + ((cm_tile_swz(person) << 5) | (((((m & 3) >> 1) ^ (person >> 1)) & 1) << 4) | ((m & 1) << 3));
    }
    auto step = [&](const long long p0, const uint4 (&w)[MTW], uint4 (&wnext)[MTW]) {
        const long long c32 = p0 >> 5;
        constexpr bool kBoth = NT <= 2 || NC == 1;   // both 32-person halves' coefficients in registers at once
        constexpr int NPC = PK ? 1 : kCmNP;        // operands per coefficient tile
        uint4 bg[kBoth ? NC : 1][NT][NPC];
#pragma unroll
        for (int c = 0; c < (kBoth ? NC : 1); ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const uint4* gp = gimg + ((size_t)((c32 + c) * ntot + nt0 + nt) * NPC) * 64 + lane;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) bg[c][nt][pc] = gp[64 * pc];
            }
        __builtin_amdgcn_sched_barrier(0);
        fetch(wnext, p0 + 2 * PS);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            *reinterpret_cast<uint4*>(tile + wofs[mt][0]) = cm_onehot(w[mt].x);
            *reinterpret_cast<uint4*>(tile + wofs[mt][1]) = cm_onehot(w[mt].y);
            *reinterpret_cast<uint4*>(tile + wofs[mt][2]) = cm_onehot(w[mt].z);
            *reinterpret_cast<uint4*>(tile + wofs[mt][3]) = cm_onehot(w[mt].w);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if constexpr (!kBoth) {                  // (4 N-tiles: a step is ~3000 matrix-pipe cycles, the reload hides under the first half)
                if (c == 1) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const uint4* gp = gimg + ((size_t)((c32 + 1) * ntot + nt0 + nt) * NPC) * 64 + lane;
#pragma unroll
                        for (int pc = 0; pc < NPC; ++pc) bg[0][nt][pc] = gp[64 * pc];
                    }
                }
            }
            const int cb = kBoth ? c : 0;
#pragma unroll
            for (int T = 0; T < TW; ++T) {
                // A operand: row = column 16 (TW wv + T) + m of the tile, k = persons 32 c + 8 g + kk
                const int tq = (TW * wv + T) << 5;
                const uint2 a0 = cm_lds_tr(tile + 32 * c * 256 + (rofs[0] ^ tq));
                const uint2 a1 = cm_lds_tr(tile + 32 * c * 256 + (rofs[1] ^ tq));
                const uint4 a = uint4{a0.x, a0.y, a1.x, a1.y};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (PK) {
                        acc[T][nt] = cm_mfma(a, bg[cb][nt][0], acc[T][nt]);
                    } else {
                        acc[T][nt] = cm_mfma(a, bg[cb][nt][2], acc[T][nt]);
                        acc[T][nt] = cm_mfma(a, bg[cb][nt][1], acc[T][nt]);
                        acc[T][nt] = cm_mfma(a, bg[cb][nt][0], acc[T][nt]);
                    }
                }
            }
        }
        __syncthreads();
    };
    uint4 w0[MTW], w1[MTW], w2[MTW];
    fetch(w0, pa);
    fetch(w1, pa + PS);
    long long p0 = pa;
    for (; p0 + 2 * PS < pb; p0 += 3 * PS) {
        step(p0, w0, w2);
        step(p0 + PS, w1, w0);
        step(p0 + 2 * PS, w2, w1);
    }
    if (p0 < pb) {
        step(p0, w0, w2);
        if (p0 + PS < pb) step(p0 + PS, w1, w0);
    }
    // record [128 columns = 32 g' + 8 j + kk][16 ntot]; D: lane (col n = m, g) holds rows 4 g + jj of tile T
    float* out = rec + (size_t)blockIdx.x * 128 * (16 * ntot);
#pragma unroll
    for (int T = 0; T < TW; ++T)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                out[(size_t)(16 * (TW * wv + T) + 4 * g + jj) * (16 * ntot) + 16 * (nt0 + nt) + m] = acc[T][nt][jj];
}
#define CM_BACKWARD_KERNEL_BODY                                                                                                              \
    __shared__ __attribute__((aligned(256))) char tile[PS * 256];                                                                            \
    const int S = blockIdx.x / nR, r = blockIdx.x % nR;                                                                                      \
    const int nt0 = NT * blockIdx.y;                                                                                                         \
    if (64 * S + 64 <= I) cm_backward_body<NT, AL, 0, GATHER, WV, PS, PK>(codes, stride, row_index, B, I, gimg, rec, nR, per_r, tile, S, r, ntot, nt0);  \
    else if (64 * S + 64 <= stride) cm_backward_body<NT, AL, 1, GATHER, WV, PS, PK>(codes, stride, row_index, B, I, gimg, rec, nR, per_r, tile, S, r, ntot, nt0); \
    else cm_backward_body<NT, AL, 2, GATHER, WV, PS, PK>(codes, stride, row_index, B, I, gimg, rec, nR, per_r, tile, S, r, ntot, nt0);
// 4 N-tiles: 128 accumulator + 48 coefficient + 48 code-row registers -- one wave per SIMD with the whole register file
template <bool AL, bool GATHER>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void cm_backward_wide_kernel(
    const uint8_t* __restrict__ codes, long long stride, const int64_t* __restrict__ row_index, long long B, int I, const uint4* __restrict__ gimg,
    float* __restrict__ rec, int nR, long long per_r, int ntot) {
    constexpr int NT = 4, WV = 1, PS = 64;
    constexpr bool PK = false;
    CM_BACKWARD_KERNEL_BODY
}
template <int NT, bool AL, bool GATHER, bool PK = false>
__global__ __launch_bounds__(64) void cm_backward_kernel(const uint8_t* __restrict__ codes, long long stride, const int64_t* __restrict__ row_index,
                                                             long long B, int I, const uint4* __restrict__ gimg, float* __restrict__ rec, int nR,
                                                             long long per_r, int ntot) {
    constexpr int PS = 64;
    constexpr int WV = 1;       // (WV = 2, two waves sharing the tile at 4 waves per SIMD, measured slower: 336 -> 384 us at one N-tile)
    CM_BACKWARD_KERNEL_BODY
}
// record column of (item, code): (g', j, kk) <-> item 64 S + 16 g' + 4 j + (kk >> 1), code kk & 1
__device__ __forceinline__ int cm_rec_column(int item, int c) {
    const int il = item & 63;
    return 32 * (il >> 4) + 8 * ((il >> 2) & 3) + (((il & 3) << 1) | c);
}
// dX[c][item][n] = sum over the person ranges of the records, fixed order
__global__ __launch_bounds__(256) void cm_backward_reduce_kernel(const float* __restrict__ rec, float* __restrict__ dX, int I, int nR, int N) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * I * N) return;
    const int n = t % N, item = (t / N) % I, c = t / (N * I);
    const float* rp = rec + ((size_t)(item >> 6) * nR * 128 + cm_rec_column(item, c)) * N + n;
    const size_t rs = (size_t)128 * N;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = 0;
    for (; r + 8 <= nR; r += 8) {                    // 8 loads in flight per thread
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += rp[(size_t)(r + u) * rs];
    }
    for (; r < nR; ++r) a[0] += rp[(size_t)r * rs];
    dX[t] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
// conditional posterior: S1, S2 per (head, code, item, dim) from the records -> grad_table: vibo_cond_finalize.hpp
__global__ __launch_bounds__(1024) void cm_cond_finalize_kernel(const CondFinTail t) {
    __shared__ __attribute__((aligned(16))) float smem[2048];
    cm_cond_finalize_body(t, blockIdx.x, blockIdx.y, smem);
}

// ~2048 workgroups: nS item stripes x nR person ranges of a multiple of 64 persons; nR a multiple of 8 where the persons
// allow it, so that the stripes of one person range (block index S nR + r) share an XCD's L2 for the coefficient image
static int cm_ranges(long long B, int nS, long long* per_r) {
    int nR = 2048 / (nS > 0 ? nS : 1);
    nR = (nR + 7) / 8 * 8;
    if (nR < 8) nR = 8;
    long long per = (B + nR - 1) / nR;
    per = (per + 63) / 64 * 64;
    if (per < 64) per = 64;
    *per_r = per;
    return (int)((B + per - 1) / per);
}
static size_t cm_up(size_t b) { return (b + 255) & ~(size_t)255; }
static size_t cm_timg_bytes(int nS, int NT) { return cm_up((size_t)nS * 4 * NT * kCmNP * 64 * 16); }
static size_t cm_gimg_bytes(long long B, int NT) { return cm_up((size_t)((B + 63) / 64 * 2) * NT * kCmNP * 64 * 16); }
static size_t cm_rec_bytes(long long B, int nS, int NT) {
    long long per_r;
    const int nR = cm_ranges(B, nS, &per_r);
    return cm_up((size_t)nS * nR * 128 * 16 * NT * 4);
}

template <int NT, bool COUNT>
static void cm_launch_forward_nt(bool al, dim3 grid, hipStream_t s, const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I,
                                 int nS, const uint4* img, float* out, int out_stride, int ncols) {
    if (al) hipLaunchKernelGGL((cm_forward_kernel<NT, COUNT, true>), grid, dim3(320), 0, s, codes, stride, row_index, B, I, nS, img, out, out_stride, ncols);
    else hipLaunchKernelGGL((cm_forward_kernel<NT, COUNT, false>), grid, dim3(320), 0, s, codes, stride, row_index, B, I, nS, img, out, out_stride, ncols);
}
template <bool COUNT>
static hipError_t cm_launch_forward(int NT, const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I, int nS,
                                    const uint4* img, float* out, int out_stride, int ncols, hipStream_t s) {
    const bool al = stride % 16 == 0 && ((uintptr_t)codes & 15) == 0;
    const dim3 grid((unsigned)((B + 255) / 256));
    if (NT == 1) cm_launch_forward_nt<1, COUNT>(al, grid, s, codes, stride, row_index, B, I, nS, img, out, out_stride, ncols);
    else if (NT == 2) cm_launch_forward_nt<2, COUNT>(al, grid, s, codes, stride, row_index, B, I, nS, img, out, out_stride, ncols);
    else cm_launch_forward_nt<4, COUNT>(al, grid, s, codes, stride, row_index, B, I, nS, img, out, out_stride, ncols);
    return hipGetLastError();
}
static void cm_launch_backward_wide(bool al, dim3 grid, hipStream_t s, const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I,
                                    const uint4* gimg, float* rec, int nR, long long per_r, int ntot) {
    if (row_index) {
        if (al) hipLaunchKernelGGL((cm_backward_wide_kernel<true, true>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
        else hipLaunchKernelGGL((cm_backward_wide_kernel<false, true>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
    } else {
        if (al) hipLaunchKernelGGL((cm_backward_wide_kernel<true, false>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
        else hipLaunchKernelGGL((cm_backward_wide_kernel<false, false>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
    }
}
static void cm_launch_backward_packed(bool al, dim3 grid, hipStream_t s, const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I,
                                      const uint4* gimg, float* rec, int nR, long long per_r) {
    if (row_index) {
        if (al) hipLaunchKernelGGL((cm_backward_kernel<1, true, true, true>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, 1);
        else hipLaunchKernelGGL((cm_backward_kernel<1, false, true, true>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, 1);
    } else {
        if (al) hipLaunchKernelGGL((cm_backward_kernel<1, true, false, true>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, 1);
        else hipLaunchKernelGGL((cm_backward_kernel<1, false, false, true>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, 1);
    }
}
template <int NT>
static void cm_launch_backward_nt(bool al, dim3 grid, hipStream_t s, const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I,
                                  const uint4* gimg, float* rec, int nR, long long per_r, int ntot) {
    if (row_index) {
        if (al) hipLaunchKernelGGL((cm_backward_kernel<NT, true, true>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
        else hipLaunchKernelGGL((cm_backward_kernel<NT, false, true>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
    } else {
        if (al) hipLaunchKernelGGL((cm_backward_kernel<NT, true, false>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
        else hipLaunchKernelGGL((cm_backward_kernel<NT, false, false>), grid, dim3(64), 0, s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
    }
}
// ntot N-tiles in all (1, 2 or a multiple of 4: blockIdx.y slices of 4)
static hipError_t cm_launch_backward(int ntot, const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I, int nS,
                                     const uint4* gimg, float* rec, hipStream_t s, int* nR_out, bool packed = false) {
    long long per_r;
    const int nR = cm_ranges(B, nS, &per_r);
    *nR_out = nR;
    const bool al = stride % 16 == 0 && ((uintptr_t)codes & 15) == 0;
    if (packed) cm_launch_backward_packed(al, dim3((unsigned)(nS * nR)), s, codes, stride, row_index, B, I, gimg, rec, nR, per_r);
    else if (ntot == 1) cm_launch_backward_nt<1>(al, dim3((unsigned)(nS * nR)), s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
    else if (ntot == 2) cm_launch_backward_nt<2>(al, dim3((unsigned)(nS * nR)), s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
    else cm_launch_backward_wide(al, dim3((unsigned)(nS * nR), (unsigned)(ntot / 4)), s, codes, stride, row_index, B, I, gimg, rec, nR, per_r, ntot);
    return hipGetLastError();
}

// ---- the conditional posterior's two passes (vibo_cond.hpp) -------------------------------------------------------------
size_t cond_mfma_scratch_bytes(long long B, int I, int A) {
    const int nS = (I + 63) / 64;
    const int NTb = (4 * A + 15) / 16;
    return cm_timg_bytes(nS, 1) + cm_gimg_bytes(B, NTb) + cm_rec_bytes(B, nS, NTb);
}
// pre[B][2A + 1] = lam | s | nobs of every person from its code row (all items in one launch)
hipError_t launch_cond_pre_mfma(const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I, int A,
                                const float* table, float* pre, void* scratch, hipStream_t s) {
    const int nS = (I + 63) / 64;
    uint4* img = static_cast<uint4*>(scratch);
    // the observed count: a column of ones while the 16-column tile has room for it (exact: a sum of 1.0s), the code words'
    // population count at 8 ability dims
    const int ncols = 2 * A < 16 ? 2 * A + 1 : 2 * A;
    hipLaunchKernelGGL((cm_table_image_kernel<true>), dim3((nS * 4 * 64 + 255) / 256), dim3(256), 0, s, table, img, I, nS, 1, ncols);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ncols > 2 * A) return cm_launch_forward<false>(1, codes, stride, row_index, B, I, nS, img, pre, 2 * A + 1, ncols, s);
    return cm_launch_forward<true>(1, codes, stride, row_index, B, I, nS, img, pre, 2 * A + 1, 2 * A, s);
}
// the same from fp32 rows (+ u8 mask or none): the rows' cell codes are left in `codes_out` (minibatch order, rows of
// `codes_stride` bytes = whole 64-byte steps) for the passes that follow
hipError_t launch_cond_pre_mfma_fp32(const float* response, const void* mask, long long resp_stride, long long mask_stride,
                                     const int64_t* row_index, long long B, int I, int A, const float* table, float* pre,
                                     uint8_t* codes_out, long long codes_stride, void* scratch, hipStream_t s) {
    const int nS = (I + 63) / 64;
    uint4* img = static_cast<uint4*>(scratch);
    const int ncols = 2 * A < 16 ? 2 * A + 1 : 2 * A;
    hipLaunchKernelGGL((cm_table_image_kernel<true>), dim3((nS * 4 * 64 + 255) / 256), dim3(256), 0, s, table, img, I, nS, 1, ncols);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint8_t* mk = static_cast<const uint8_t*>(mask);
    const bool mal = !mk || (mask_stride % 16 == 0 && ((uintptr_t)mk & 15) == 0);
    const dim3 grid((unsigned)((B + 64 * kCmF32MT - 1) / (64 * kCmF32MT)));
    const bool count = ncols == 2 * A;               // 8 dims: no room for the ones column, the code words are counted
#define CM_FWD32(C, M) hipLaunchKernelGGL((cm_forward_fp32_kernel<C, M>), grid, dim3(320), 0, s, response, mk, resp_stride, mask_stride, row_index, B, I, \
                                          nS, (const uint4*)img, pre, 2 * A + 1, count ? 2 * A : ncols, codes_out, codes_stride)
    if (count) { if (mal) CM_FWD32(true, true); else CM_FWD32(true, false); }
    else { if (mal) CM_FWD32(false, true); else CM_FWD32(false, false); }
#undef CM_FWD32
    return hipGetLastError();
}
// grad_table[2 heads][2][I][2A] from the per-person coefficients coef[B][4A] = [head][P1 | P2][dim]
hipError_t launch_cond_post_mfma(const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I, int A,
                                 const float* table, const float* coef, float* grad_table, void* scratch, hipStream_t s,
                                 CondFinTail* defer) {
    const int nS = (I + 63) / 64;
    const int NT = (4 * A + 15) / 16;
    char* base = static_cast<char*>(scratch) + cm_timg_bytes(nS, 1);
    uint4* gimg = reinterpret_cast<uint4*>(base);
    float* rec = reinterpret_cast<float*>(base + cm_gimg_bytes(B, NT));
    const long long n32 = (B + 63) / 64 * 2;
    const bool packed = 3 * 4 * A <= 16;             // ability_dim 1: the three pieces of the 4 coefficient columns in one operand
    if (packed) hipLaunchKernelGGL(cm_grad_image_packed_kernel, dim3((unsigned)((n32 * 64 + 255) / 256)), dim3(256), 0, s, coef, gimg, B, n32, 4 * A);
    else hipLaunchKernelGGL(cm_grad_image_kernel, dim3((unsigned)((n32 * NT * 64 + 255) / 256)), dim3(256), 0, s, coef, gimg, B, n32, NT, 4 * A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int nR = 0;
    e = cm_launch_backward(NT, codes, stride, row_index, B, I, nS, gimg, rec, s, &nR, packed);
    if (e != hipSuccess) return e;
    CondFinTail t;
    memset(&t, 0, sizeof(t));
    t.packed_cols = packed ? 4 * A : 0;
    t.kind = 2; t.gx = nS; t.gy = 4; t.rec = rec; t.table = table; t.grad_table = grad_table; t.I = I; t.A = A; t.nR = nR; t.N = 16 * NT;
    if (defer) { *defer = t; return hipSuccess; }        // (rides in the ELBO finalize launch)
    hipLaunchKernelGGL(cm_cond_finalize_kernel, dim3(nS, 4), dim3(1024), 0, s, t);
    return hipGetLastError();
}

}  // namespace vibo

using namespace vibo;

extern "C" {

// bytes of scratch the two calls need (images + partial records), 256-byte aligned
size_t vibo_code_table_scratch_bytes(int64_t num_person, int num_item, int hidden_dim) {
    if (num_person < 1 || num_item < 1 || hidden_dim != kCmH) return 0;
    const int nS = (num_item + 63) / 64;
    return cm_timg_bytes(nS, 4) + cm_gimg_bytes(num_person, 4) + cm_rec_bytes(num_person, nS, 4) + 256;
}

/* S [B][64] = sum over the observed cells of feature[code][item][:]   (feature = elu(mlp1([c, item_i])), [2][I][64]) */
int vibo_code_table_sum_forward(int64_t num_person, int num_item, int hidden_dim, const uint8_t* codes, int64_t codes_row_stride,
                                const float* feature, float* out_sum, void* scratch, size_t scratch_bytes, void* stream) {
    if (num_person < 1 || num_item < 1) return -3;
    if (hidden_dim != kCmH) return -6;
    if (!codes || !feature || !out_sum || !scratch) return -5;
    if (codes_row_stride % 4 != 0 || ((uintptr_t)codes & 3) || ((uintptr_t)scratch & 255)) return -8;
    if (scratch_bytes < vibo_code_table_scratch_bytes(num_person, num_item, hidden_dim)) return -7;
    const int nS = (num_item + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    uint4* img = static_cast<uint4*>(scratch);
    hipLaunchKernelGGL((cm_table_image_kernel<false>), dim3((nS * 4 * 4 * 64 + 255) / 256), dim3(256), 0, s, feature, img, num_item, nS, 4, kCmH);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return (int)cm_launch_forward<false>(4, codes, (long long)codes_row_stride, nullptr, (long long)num_person, num_item, nS, img, out_sum, kCmH,
                                         kCmH, s);
}

/* d feature [2][I][64] = sum over the persons of [code == c] grad_sum[p][:] */
int vibo_code_table_sum_backward(int64_t num_person, int num_item, int hidden_dim, const uint8_t* codes, int64_t codes_row_stride,
                                 const float* grad_sum, float* grad_feature, void* scratch, size_t scratch_bytes, void* stream) {
    if (num_person < 1 || num_item < 1) return -3;
    if (hidden_dim != kCmH) return -6;
    if (!codes || !grad_sum || !grad_feature || !scratch) return -5;
    if (codes_row_stride % 4 != 0 || ((uintptr_t)codes & 3) || ((uintptr_t)scratch & 255)) return -8;
    if (scratch_bytes < vibo_code_table_scratch_bytes(num_person, num_item, hidden_dim)) return -7;
    const int nS = (num_item + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    char* base = static_cast<char*>(scratch) + cm_timg_bytes(nS, 4);
    uint4* gimg = reinterpret_cast<uint4*>(base);
    float* rec = reinterpret_cast<float*>(base + cm_gimg_bytes(num_person, 4));
    const long long n32 = (num_person + 63) / 64 * 2;
    hipLaunchKernelGGL(cm_grad_image_kernel, dim3((unsigned)((n32 * 4 * 64 + 255) / 256)), dim3(256), 0, s, grad_sum, gimg, (long long)num_person, n32,
                       4, kCmH);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    int nR = 0;
    e = cm_launch_backward(4, codes, (long long)codes_row_stride, nullptr, (long long)num_person, num_item, nS, gimg, rec, s, &nR);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cm_backward_reduce_kernel, dim3((2 * num_item * kCmH + 255) / 256), dim3(256), 0, s, (const float*)rec, grad_feature, num_item,
                       nR, kCmH);
    return (int)hipGetLastError();
}

}  // extern "C"
