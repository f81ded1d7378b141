// vibo_msplit_kernel.hpp -- fused ELBO forward+backward with the three ability-wide contractions on the matrix
// pipe ("matrix row-split" kernel).  Unconditional / conditional / caller-supplied posterior, 1PL/2PL/3PL,
// any ability_dim <= 8, I <= 1024 per launch (panels beyond), rows chunkable in 4 cells.
//
// Why: the row-split kernel (vibo_split_kernel.hpp) is VALU-issue-bound at ability_dim 8 -- 24 of its ~36 VALU
// instructions per term are the packed FMAs of  logit = theta . a,  d LL/d theta = g . a,  d LL/d a = g^T theta
// -- while the matrix pipe idles (profiles/r01_split_kernel_pmc_a8.txt: SQ_INSTS_MFMA = 0).  Measured on gfx950
// (tools/ubench/ubench4.hip, ubench5.hip): v_mfma_f32_16x16x32_f16 issues every ~20 cycles and costs a VALU stream
// only ~9 cycles of issue, f16 subnormal inputs are honoured, and a 2-piece f16 split x = hi + lo (round toward
// zero) leaves |x - hi - lo| <= 2.4e-7 |x|, i.e. fp32-grade.  So the contractions move to f16 MFMAs on hi/lo pieces
// with fp32 accumulation:
//   * logit  [32 persons x 16 items, K = 32]: K = {theta_hi.a_hi, theta_hi.a_lo, theta_lo.a_hi, 1.(b_hi,b_mid,b_lo)}
//     -> one MFMA per 16 persons x 16 items; D layout: lane (item i16 = lane & 15, g = lane >> 4) holds persons
//     4 g + j (+ 16 per M-tile) of item i16 -- exactly the cells the lane loaded from HBM (float4 = 4 item tiles);
//   * d LL/d a [16 items x 16 cols, K = 32 persons]: A = g (hi, then lo) straight from the D layout of the logits,
//     B = [theta_hi | theta_lo] -> two MFMAs per tile, accumulated in registers for the whole kernel (no reduction);
//   * d LL/d theta [16 persons x 16 cols, K = 32 items]: needs g with persons in lanes: the hi/lo pieces go through a
//     wave-private LDS image written as the d LL/d a operand registers (ds_write_b64) and read back transposed by
//     ds_read_b64_tr_b16 (conflict-free XOR-swizzled 8-byte pieces), B = [a_hi | a_lo].
// What stays on the VALU per term: clamp, exp2, 1 + e, rcp, one fma for g, product for the shared log2, the f16
// split (~11 plain + 2.25 transcendental instructions instead of ~36).
//
// Geometry: a workgroup of nw = ceil(I / 128) waves shares a batch of 32 response rows; wave q owns items
// [128 q, 128 q + 128) as two "u-steps" of 64 items.  Per u-step and lane: 8 persons x one float4 (+ 4 mask bytes),
// a wave-load instruction reads 4 rows x 256 contiguous bytes.  Loads run one u-step ahead of the math; the next
// batch's cells are packed to fp8 codes (16 registers) as they arrive.  Two workgroup barriers per batch: counts +
// d LL/d theta shares out, (person, dim) lanes do backward(batch) and forward(batch + 1), theta operands back.
// Outputs use the per-workgroup partial record of the other kernels (fixed order, bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_params.hpp"

namespace vibo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

constexpr int kMsRows = 32;       // rows per batch
constexpr int kMsSpan = 128;      // items per wave

constexpr int kMsItemRow = 24;    // halfs per item of the operand image: na_hi[8] | na_lo[8] | nb pieces[3] | 0...
constexpr int kMsItemLane = 4 * kMsItemRow + 8;   // halfs per float4 chunk (4 items) + 16 B of padding (208 B: conflict-free b128 reads)
struct alignas(16) MsWaveLds {
    _Float16 tr[2][2][64 * 16];   // [hi | lo][M-tile][row = 16 t + i16][16 persons]: g pieces of a u-step
    _Float16 img[2][16 * kMsItemLane];   // [u-step][chunk i16][item t][24]: MFMA operands of this wave's 128 items
    float gth[2][kMsRows][8];     // [batch parity] this wave's share of d LL/d theta of the batch (log2 units)
    int cnt[kMsRows];             // packed counts (n1 << 16 | nobs) of this wave's items
    float red[8];
};
struct alignas(16) MsCommonLds {
    _Float16 thA[2][kMsRows][8];  // theta hi | lo, [person][dim]: A operand of the logit MFMA
    _Float16 thT[16][kMsRows];    // [dim (hi) | 8 + dim (lo)][slot]: B operand of the d LL/d a MFMA
    float st[2][5][256];          // [batch parity] forward state of the (person, dim) pairs, kept for the backward
    float ctab[4 * 2 * 8];
    float tred[8][8][8];          // [wave][k][dim] table-gradient sums of the wave's (person, dim) lanes
};
inline size_t msplit_lds_bytes(int nw) { return sizeof(MsCommonLds) + (size_t)nw * sizeof(MsWaveLds); }

__device__ __forceinline__ half2v pkrtz(float a, float b) {
    return __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// x = hi + lo, both f16 (round toward zero): |x - hi - lo| <= 2^-22 |x|
__device__ __forceinline__ void split16(float x, _Float16& hi, _Float16& lo) {
    const half2v h = pkrtz(x, 0.f);
    hi = h[0];
    lo = pkrtz(x - (float)h[0], 0.f)[0];
}
// buffer resource over `bytes` bytes from `base` (wave-uniform); loads past the end return zeros
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ms_rsrc(const void* base, long long bytes) {
    const unsigned n = bytes <= 0 ? 0u : bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)n, 0x00020000);
}
__device__ __forceinline__ f32x4 mfma16(const half8 a, const half8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ half4v lds_tr16(const _Float16* p) {
    return __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)p));
}
__device__ __forceinline__ half8 cat8(const half4v a, const half4v b) {
    return half8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ half8 cat8(const half2v a, const half2v b, const half2v c, const half2v d) {
    return half8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// IRT: 1/2/3PL.  GRAD: also gradients.  RM (row mode): 0 = fp32 responses + mask bytes, rows in order; 1 = the same
// through p.row_index; 2 = 1-byte cell codes (VIBO_MASK_CODES, through p.mask), with or without p.row_index.
// blockDim.x = 64 nw, dynamic LDS = msplit_lds_bytes(nw).
template <int IRT, bool GRAD, int RM>
__global__ __launch_bounds__(512, 2) void msplit_kernel(const ElboParams p) {
    constexpr bool CODES = RM == 2;
    constexpr int R = kMsRows;
    constexpr float kLoS = kLogitLo * kLog2e, kHiS = kLogitHi * kLog2e;
    extern __shared__ __attribute__((aligned(16))) unsigned char ms_smem[];
    MsCommonLds& cl = *reinterpret_cast<MsCommonLds*>(ms_smem);
    MsWaveLds* wls = reinterpret_cast<MsWaveLds*>(ms_smem + sizeof(MsCommonLds));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (int)(blockDim.x >> 6);
    MsWaveLds& wl = wls[q];
    const int I = p.I, A = p.A;
    const int n4 = (I + 3) >> 2;
    const int i16 = lane & 15, g = lane >> 4;

    if (tid < 16) {
        const int c = tid >> 3, a = tid & 7;
        float m = 0.f, s = 0.f;
        if (a < A) { m = p.table[c * 2 * A + a]; s = p.table[c * 2 * A + A + a]; }
        const float es = __expf(s);
        const float tau = 1.0f / (es + kPoeEps);
        cl.ctab[(0 * 2 + c) * 8 + a] = tau;
        cl.ctab[(1 * 2 + c) * 8 + a] = m * tau;
        cl.ctab[(2 * 2 + c) * 8 + a] = tau * tau * es;
        cl.ctab[(3 * 2 + c) * 8 + a] = m;
    }

    // ---- item operands of this wave's 128 items (prepped rows, log2 units: [na_0..na_7, nb, guess, 1 - guess]) go to
    //      a wave-private LDS image as f16 hi/lo pieces; the MFMA operands are read from there when needed:
    //   logit MFMA, tile (u, t): lane (item 4 (32 q + 16 u + i16) + t, g) reads 16 B: g 0/2 = na hi, 1 = na lo, 3 = bias pieces
    //   d LL/d theta MFMA, (u, kt): lane (col i16, g) gets k = 8 g + kk <-> item (chunk 4 g + (kk & 3), t = 2 kt + (kk >> 2)) by two
    //   transposed reads of the [na_hi | na_lo] rows
    float gs[2][4], om[2][4];           // 3PL: guess, 1 - guess of the lane's items
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int xl = 64 * h + lane;                                   // item of the wave, one per lane and pass
        const int il = kMsSpan * q + xl;
        const bool ok = il < I;
        const float* ir = p.item_prep + (size_t)(p.item0 + (ok ? il : 0)) * p.DP;
        _Float16* dst = &wl.img[h][0] + ((xl & 63) >> 2) * kMsItemLane + (xl & 3) * kMsItemRow;
        half8 hi8, lo8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            _Float16 hi, lo;
            split16(ok ? ir[kk] : 0.f, hi, lo);
            hi8[kk] = hi; lo8[kk] = lo;
        }
        const float nb = ok ? ir[8] : 0.f;
        _Float16 b0, b1, b2, b3;
        split16(nb, b0, b1);
        split16(nb - (float)b0 - (float)b1, b2, b3);
        *reinterpret_cast<half8*>(dst) = hi8;
        *reinterpret_cast<half8*>(dst + 8) = lo8;
        *reinterpret_cast<half8*>(dst + 16) = half8{b0, b1, b2, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int il = kMsSpan * q + 64 * u + 4 * i16 + t;
            const bool ok = IRT == 3 && il < I;
            const float* ir = p.item_prep + (size_t)(p.item0 + (ok ? il : 0)) * p.DP;
            gs[u][t] = ok ? ir[9] : 0.f;
            om[u][t] = ok ? ir[10] : 1.f;
        }
    // per-lane offsets (halfs) into the operand image
    const int b1ofs = i16 * kMsItemLane + (g == 1 ? 8 : g == 3 ? 16 : 0);
    const int b3ofs = (4 * g + (i16 >> 2)) * kMsItemLane + 4 * (i16 & 3);
    f32x4 acc_ga[2][4];                 // d LL/d a: [16 items of tile (u, t)][theta_hi cols | theta_lo cols]
    float acc_b[2][4], acc_g[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc_ga[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_b[u][t] = 0.f;
            acc_g[u][t] = 0.f;
        }
    f32x4 acc_gt[2];                    // d LL/d theta of the batch: [16 persons of M-tile][a_hi cols | a_lo cols]
    acc_gt[0] = acc_gt[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (lane < 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) cl.tred[q][k][lane] = 0.f;
    }
    float s_log = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_nobs = 0.f;
    int unobs = 0;
    __syncthreads();
    const int ed = lane & 7;

    // LDS image offsets (halfs): producer row 16 t + i16, piece g; consumer rows 32 kt + 4 g + (i16 >> 2) (+ 16), piece i16 & 3
    const int wofs = 16 * i16 + 4 * (g ^ (i16 >> 2));
    const int rofs = 64 * g + 16 * (i16 >> 2) + 4 * ((i16 & 3) ^ g);

    const long long n_batches = ((long long)p.B + R - 1) / R;
    float4 x[CODES ? 1 : 8];
    uint32_t m[8];
    int ridx[8];
    auto fetch_idx = [&](const long long bt) {
        if constexpr (RM != 0) {
            if (!p.row_index) return;
            const long long row0 = bt * R;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long long row = row0 + 4 * g + (k & 3) + 16 * (k >> 2);
                const long long rc = min(row, (long long)p.B - 1);
                ridx[k] = bt < n_batches ? (int)p.row_index[rc] : 0;
            }
        }
    };
    // In-order rows go through buffer loads: one resource per batch (scalar registers) whose record limit ends at the last
    // row of the matrix (rows past the end read as zeros), ONE per-lane offset shared by the 8 rows of a lane and the row
    // step as a scalar offset -- no per-row address registers.  Gathered rows compute their addresses at the load.  Chunks
    // past the row's end read its last chunk; both cases are masked when the cells are packed.
    auto load_ustep = [&](const long long bt, const int u) {
        const long long row0 = bt * R;
        const long long left = (long long)p.B - row0;
        if (left <= 0) return;
        const int c = min(32 * q + 16 * u + i16, n4 - 1);
        const long long nrow = left < R ? left : R;
        bool linear = RM == 0;
        if constexpr (RM == 2) linear = p.row_index == nullptr;
        if (linear) {
            const unsigned mvo = (unsigned)(4 * g * (int)p.mask_stride + 4 * c);
            const __amdgpu_buffer_rsrc_t mrs = ms_rsrc(static_cast<const uint8_t*>(p.mask) + row0 * p.mask_stride + p.item0,
                                                       (nrow - 1) * p.mask_stride + 4 * n4);
            if constexpr (CODES) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    m[k] = __builtin_amdgcn_raw_buffer_load_b32(mrs, mvo, ((k & 3) + 16 * (k >> 2)) * (int)p.mask_stride, 0);
            } else {
                const unsigned rvo = (unsigned)(16 * g * (int)p.resp_stride + 16 * c);
                const __amdgpu_buffer_rsrc_t rrs = ms_rsrc(p.response + row0 * p.resp_stride + p.item0,
                                                           ((nrow - 1) * p.resp_stride + 4 * n4) * 4);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rrs, rvo, ((k & 3) + 16 * (k >> 2)) * 4 * (int)p.resp_stride, 0);
                    x[k] = __builtin_bit_cast(float4, v);
                    if (p.mask_dtype == 0)
                        m[k] = __builtin_amdgcn_raw_buffer_load_b32(mrs, mvo, ((k & 3) + 16 * (k >> 2)) * (int)p.mask_stride, 0);
                    else
                        m[k] = 0x01010101u;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long long src = (long long)ridx[k];
                if constexpr (CODES) {
                    m[k] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride + p.item0)[c];
                } else {
                    x[k] = reinterpret_cast<const float4*>(p.response + src * p.resp_stride + p.item0)[c];
                    if (p.mask_dtype == 0)
                        m[k] = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride + p.item0)[c];
                    else
                        m[k] = 0x01010101u;
                }
            }
        }
    };
    auto pack_ustep = [&](const long long bt, const int u, uint32_t (&cw)[8], int (&pk)[8]) {
        const int c = 32 * q + 16 * u + i16;
        const uint32_t tail_mask = c >= n4 ? 0u : ((I & 3) && c == (I >> 2)) ? ((1u << (8 * (I & 3))) - 1u) : 0xFFFFFFFFu;
        const long long left = (long long)p.B - bt * R;          // (wave-uniform) only the last batch has rows past the end
        if (left >= R) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (CODES) cw[k] = pack_cell_codes4(m[k], tail_mask, pk[k]);
                else cw[k] = pack_codes4(x[k], m[k] & tail_mask, pk[k]);
            }
        } else {
            const int nleft = (int)left;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t keep = (4 * g + (k & 3) + 16 * (k >> 2) < nleft) ? tail_mask : 0u;
                if constexpr (CODES) cw[k] = pack_cell_codes4(m[k], keep, pk[k]);
                else cw[k] = pack_codes4(x[k], m[k] & keep, pk[k]);
            }
        }
    };
    // packed counts of the lane's 8 persons (both u-steps) -> 16-lane sums -> wl.cnt
    auto put_counts = [&](const int (&pk)[8], const bool real) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int t = pk[k] + (pk[k + 4] << 8);       // fields of 8 bits: nobs_k | nobs_k+4 | n1_k | n1_k+4  (each <= 128)
            t += dpp_i<0xb1>(t);                    // quad_perm [1,0,3,2]
            t += dpp_i<0x4e>(t);                    // quad_perm [2,3,0,1]
            t += dpp_i<0x141>(t);                   // row_half_mirror
            t += dpp_i<0x140>(t);                   // row_mirror
            v[k] = t;
        }
        if constexpr (IRT != 3) {
            int obs = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) obs += pk[k] & 0xffff;
            if (real) unobs += 64 - obs;        // (a batch past the end is packed but never evaluated)
        }
        const int sel = i16 & 3;
        const int vv = sel == 0 ? v[0] : sel == 1 ? v[1] : sel == 2 ? v[2] : v[3];
        const int f = vv >> ((i16 & 4) ? 8 : 0);
        if (i16 < 8) wl.cnt[4 * g + (i16 & 3) + 16 * (i16 >> 2)] = (f & 0xff) | (((f >> 16) & 0xff) << 16);
    };

    // ---- (person, dim) lanes.  Slot s = 64 of the batch's 256 (person, dim) pairs; with 8 waves the slots of even batches
    //      belong to waves 0-3 and those of odd batches to waves 4-7, else to wave s mod nw.
    // product of experts + reparameterised sample of one slot (models.py:596-629)
    auto forward_slot = [&](const long long bt, const int par, const int s, const float eps_c) {
        const long long row0 = bt * R;
        const int e = 64 * s + lane, pp = e >> 3;
        const bool live = ed < A && (row0 + pp) < p.B;
        int cnt = 0;
        if (p.row_cnt) {
            cnt = live ? p.row_cnt[row0 + pp] : 0;
        } else {
#pragma unroll 1
            for (int w = 0; w < nw; ++w) cnt += wls[w].cnt[pp];
        }
        const float n1 = (float)(cnt >> 16);
        float nobs = (float)(cnt & 0xffff);
        const float n0 = nobs - n1;
        const float tau0 = cl.ctab[(0 * 2 + 0) * 8 + ed], tau1 = cl.ctab[(0 * 2 + 1) * 8 + ed];
        const float mt0 = cl.ctab[(1 * 2 + 0) * 8 + ed], mt1 = cl.ctab[(1 * 2 + 1) * 8 + ed];
        float lam = n0 * tau0 + n1 * tau1, smu = n0 * mt0 + n1 * mt1;
        if (p.pre_stats) {
            lam = 0.f; smu = 0.f; nobs = 0.f;
            if (live) {
                for (int pn = 0; pn < p.pre_panels; ++pn) {
                    const float* st = p.pre_stats + ((size_t)pn * p.B + (row0 + pp)) * (2 * A + 1);
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
        const float th0 = live ? amu + sig * eps_c : 0.f;
        const float thv = th0;
        if (live && p.primary) {
            const long long o = (row0 + pp) * A + ed;
            const float alv = -kLn2 * fast_log2(lam);
            p.ability_mu[o] = amu;
            p.ability_logvar[o] = alv;
            p.ability[o] = th0;
            s_kl += -0.5f * (1.0f + alv - amu * amu - inv_lam);
            s_logq0 += -0.5f * kLog2Pi - 0.5f * alv - 0.5f * eps_c * eps_c;
            s_logp += -0.5f * kLog2Pi - 0.5f * thv * thv;
            if (ed == 0) s_nobs += nobs;
        }
        if constexpr (GRAD) {
            cl.st[par][0][e] = amu; cl.st[par][1][e] = sig; cl.st[par][2][e] = inv_lam; cl.st[par][3][e] = eps_c;
            cl.st[par][4][e] = __builtin_bit_cast(float, cnt);
        }
        _Float16 hi, lo;
        split16(thv, hi, lo);
        const int slot = 8 * ((pp & 15) >> 2) + 4 * (pp >> 4) + (pp & 3);
        cl.thA[0][pp][ed] = hi;
        cl.thA[1][pp][ed] = lo;
        cl.thT[ed][slot] = hi;
        cl.thT[8 + ed][slot] = lo;
    };
    // backward of one slot through the sample and the product of experts; the 8 table-gradient sums of the wave's lanes
    // go to the wave's LDS record (fixed order: bitwise reproducible)
    auto backward_slot = [&](const long long bt, const int par, const int s) {
        const long long row0 = bt * R;
        const int e = 64 * s + lane, pp = e >> 3;
        const bool live = ed < A && (row0 + pp) < p.B;
        float g0 = 0.f;
#pragma unroll 1
        for (int w = 0; w < nw; ++w) g0 += wls[w].gth[par][pp][ed];
        const float gz0 = live ? g0 * kLn2 : 0.f;
        const float amu = cl.st[par][0][e], sig = cl.st[par][1][e], inv_lam = cl.st[par][2][e], eps_c = cl.st[par][3][e];
        const int cnt = __builtin_bit_cast(int, cl.st[par][4][e]);
        const float n1 = (float)(cnt >> 16), n0 = (float)(cnt & 0xffff) - n1;
        const float thv = live ? amu + sig * eps_c : 0.f;
        const bool reg_on = live && p.primary;
        const float gz1 = (reg_on && p.reg_mode != 0) ? thv : 0.f;
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
        if (p.post_coef) {
            if (live) {
                float* pc = p.post_coef + (size_t)(row0 + pp) * 4 * A;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    pc[(st * 2 + 0) * A + ed] = gmu[st] * inv_lam;
                    pc[(st * 2 + 1) * A + ed] = -(gmu[st] * amu + glv[st]) * inv_lam;
                }
            }
        }
        const float nn[2] = {n0, n1};
        float dt[8];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float tau = cl.ctab[(0 * 2 + c) * 8 + ed];
            const float te = cl.ctab[(2 * 2 + c) * 8 + ed], mm = cl.ctab[(3 * 2 + c) * 8 + ed];
            const float nl = live ? nn[c] * inv_lam : 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                dt[st * 4 + c * 2 + 0] = gmu[st] * nl * tau;
                const float g_tau = nl * (gmu[st] * (mm - amu) - glv[st]);
                dt[st * 4 + c * 2 + 1] = -g_tau * te;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = dt[k];
            v += dpp_f<0x128>(v);                     // row_ror 8: lanes d and d + 8 of a row
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            dt[k] = v;
        }
        if (lane < 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) cl.tred[q][k][lane] += dt[k];
        }
    };
    // slots of this wave for a batch: [s0, s1) in steps of `step` (one slot at most with 4 or more waves)
    auto my_slots = [&](const int par, int& s0, int& s1, int& step) {
        if (nw == 8) {
            s0 = q - 4 * par; s1 = s0 + 1; step = 1;
            if (s0 < 0 || s0 > 3) s1 = s0 = 0;
        } else {
            s0 = q; s1 = 4; step = nw;
        }
    };
    float epn = 0.f;                                  // eps of this wave's slot of the next batch (4 or more waves), loaded a batch ahead
    auto fetch_eps = [&](const long long bt, const int par) {
        if (nw < 4 || bt >= n_batches) return;
        int s0, s1, step;
        my_slots(par, s0, s1, step);
        if (s0 < s1) {
            const long long row = bt * R + ((64 * s0 + lane) >> 3);
            epn = (ed < A && row < p.B) ? p.eps[row * A + ed] : 0.f;
        }
    };
    auto person_forward = [&](const long long bt, const int par) {
        int s0, s1, step;
        my_slots(par, s0, s1, step);
#pragma unroll 1
        for (int s = s0; s < s1; s += step) {
            float eps_c = epn;
            if (nw < 4) {
                const long long row = bt * R + ((64 * s + lane) >> 3);
                eps_c = (ed < A && row < p.B) ? p.eps[row * A + ed] : 0.f;
            }
            forward_slot(bt, par, s, eps_c);
        }
    };
    auto person_backward = [&](const long long bt, const int par) {
        int s0, s1, step;
        my_slots(par, s0, s1, step);
#pragma unroll 1
        for (int s = s0; s < s1; s += step) backward_slot(bt, par, s);
    };

    // ---- one u-step of math: 4 item tiles x 2 M-tiles ----
    half8 A1[2], B2;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto math_ustep = [&](auto uc, const uint32_t (&cw)[8]) {
        constexpr int u = decltype(uc)::value;
        auto tile = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const half8 b1 = *reinterpret_cast<const half8*>(&wl.img[u][0] + b1ofs + t * kMsItemRow);
            const f32x4 d0 = mfma16(A1[0], b1, zero4), d1 = mfma16(A1[1], b1, zero4);
            const float lg[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
            float gl[8];
            float pr0 = 1.0f, pr1 = 1.0f, lmax = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = code_to_f32<t>(cw[k]);
                float& pr = (k & 1) ? pr1 : pr0;
                gl[k] = 0.f;
                if constexpr (IRT != 3) {
                    const float lc = med3(lg[k], -kLoS, kLoS);
                    const float eu = fast_exp2(-w * lc);                  // exactly 1 for a missing cell (w = 0)
                    const float tt = 1.0f + eu;
                    pr *= tt;                                             // <= (1 + 2^23)^4: one log2 per 4 terms
                    if constexpr (GRAD) gl[k] = fmaf(-w, fast_rcp(tt), w);   // w e / (1 + e) = d ll / d logit
                    if constexpr (GRAD) lmax = fmaxf(lmax, fabsf(lg[k]));
                } else {
                    // 3PL: p = guess + (1 - guess) sigmoid(l)  (models.py:758-765), probability clamp on p itself
                    const float l = lg[k];
                    const float ee = fast_exp2(-fabsf(l));
                    const float rr_ = fast_rcp(1.0f + ee);
                    const float er_ = ee * rr_;
                    const float sp = (l >= 0.f) ? rr_ : er_;
                    const float sn = (l >= 0.f) ? er_ : rr_;
                    const float prb = fmaf(om[u][t], sp, gs[u][t]);
                    const float qr = om[u][t] * sn;
                    const float pc = med3(prb, kEps32, 1.0f - kEps32);
                    const float arg = (w > 0.f) ? pc : med3(qr, kEps32, 1.0f - kEps32);
                    pr *= (w != 0.f) ? arg : 1.0f;
                    if constexpr (GRAD) {
                        const float wlv = (prb == pc) ? w : 0.f;
                        const float common = wlv * fast_rcp(arg) * om[u][t] * sn;
                        gl[k] = common * sp;
                        acc_g[u][t] = fmaf(common, gs[u][t], acc_g[u][t]);
                    }
                }
            }
            s_log += fast_log2(pr0) + fast_log2(pr1);
            asm volatile("" : "+v"(s_log));          // (keeps hipcc from sinking the whole batch's products to the loop end)
            if constexpr (GRAD) {
                if constexpr (IRT != 3) {
                    if (__any(lmax > kLoS)) {
                        // rare: the reference's gradient is exactly zero outside [-kLogitLo, kLogitHi]
#pragma unroll
                        for (int k = 0; k < 8; ++k) gl[k] = (lg[k] < -kLoS || lg[k] > kHiS) ? 0.f : gl[k];
                    }
                }
                acc_b[u][t] += ((gl[0] + gl[1]) + (gl[2] + gl[3])) + ((gl[4] + gl[5]) + (gl[6] + gl[7]));
                asm volatile("" : "+v"(acc_b[u][t]));
                if constexpr (IRT == 3) asm volatile("" : "+v"(acc_g[u][t]));
                half2v hh[4], ll[4];
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    hh[k2] = pkrtz(gl[2 * k2], gl[2 * k2 + 1]);
                    ll[k2] = pkrtz(gl[2 * k2] - (float)hh[k2][0], gl[2 * k2 + 1] - (float)hh[k2][1]);
                }
                const half8 a2h = cat8(hh[0], hh[1], hh[2], hh[3]), a2l = cat8(ll[0], ll[1], ll[2], ll[3]);
                if constexpr (IRT != 1) {
                    acc_ga[u][t] = mfma16(a2h, B2, acc_ga[u][t]);
                    acc_ga[u][t] = mfma16(a2l, B2, acc_ga[u][t]);
                }
                // LDS image for the transposed read: persons of M-tile 0 = registers 0-1, M-tile 1 = registers 2-3
                _Float16* wp = &wl.tr[0][0][0] + wofs + 256 * t;
                *reinterpret_cast<uint2*>(wp) = uint2{__builtin_bit_cast(uint32_t, hh[0]), __builtin_bit_cast(uint32_t, hh[1])};
                *reinterpret_cast<uint2*>(wp + 1024) = uint2{__builtin_bit_cast(uint32_t, hh[2]), __builtin_bit_cast(uint32_t, hh[3])};
                *reinterpret_cast<uint2*>(wp + 2048) = uint2{__builtin_bit_cast(uint32_t, ll[0]), __builtin_bit_cast(uint32_t, ll[1])};
                *reinterpret_cast<uint2*>(wp + 3072) = uint2{__builtin_bit_cast(uint32_t, ll[2]), __builtin_bit_cast(uint32_t, ll[3])};
                if constexpr (t & 1) {
                    constexpr int kt = t >> 1;
                    const _Float16* rp = &wl.tr[0][0][0] + rofs + 512 * kt;
                    const _Float16* ip = &wl.img[u][0] + b3ofs + 2 * kt * kMsItemRow;
                    const half8 b3 = cat8(lds_tr16(ip), lds_tr16(ip + kMsItemRow));
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const _Float16* r0 = rp + 1024 * (2 * hl + mt);
                            const half8 a3 = cat8(lds_tr16(r0), lds_tr16(r0 + 256));
                            acc_gt[mt] = mfma16(a3, b3, acc_gt[mt]);
                        }
                }
            }
        };
        tile(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        tile(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        tile(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        tile(std::integral_constant<int, 3>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    auto put_gtheta = [&](const int par) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = acc_gt[k >> 2][k & 3] + dpp_f<0x128>(acc_gt[k >> 2][k & 3]);   // row_ror 8: a_hi col + a_lo col
        if (i16 < 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) wl.gth[par][16 * (k >> 2) + 4 * g + (k & 3)][i16] = v[k];
        }
        acc_gt[0] = acc_gt[1] = zero4;
    };
    auto read_theta_ops = [&]() {
        const half8 ones = half8{(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)0.f,
                                 (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const half8 v = *reinterpret_cast<const half8*>(&cl.thA[g >> 1][16 * mt + i16][0]);
            A1[mt] = (g == 3) ? ones : v;
        }
        B2 = *reinterpret_cast<const half8*>(&cl.thT[i16][8 * g]);
    };

    // ================= prologue: first batch =================
    long long bt = blockIdx.x;
    uint32_t cwA0[8], cwA1[8], cwB0[8], cwB1[8];
    int pk[8];
    const long long G = gridDim.x;
    if (bt < n_batches) {
        fetch_idx(bt);
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = 0;
        load_ustep(bt, 0);
        pack_ustep(bt, 0, cwA0, pk);
        load_ustep(bt, 1);
        pack_ustep(bt, 1, cwA1, pk);
        put_counts(pk, true);
        fetch_idx(bt + G);
        load_ustep(bt + G, 0);
        fetch_eps(bt, 0);
        __syncthreads();
        person_forward(bt, 0);
        __syncthreads();
        read_theta_ops();
        fetch_eps(bt + G, 1);
    }
    int par = 0;                                      // parity of the workgroup's batch counter: LDS double buffers, slot owners
    for (; bt < n_batches; bt += G, par ^= 1) {
        const long long nxt = bt + G;
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = 0;
        math_ustep(std::integral_constant<int, 0>{}, cwA0);
        pack_ustep(nxt, 0, cwB0, pk);                 // rows (nxt, u-step 0) have landed under the math
        load_ustep(nxt, 1);
        fetch_idx(nxt + G);                      // (the row indices of the batch after that, consumed one u-step later)
        math_ustep(std::integral_constant<int, 1>{}, cwA1);
        pack_ustep(nxt, 1, cwB1, pk);
        load_ustep(nxt + G, 0);
        put_counts(pk, nxt < n_batches);
        if constexpr (GRAD) put_gtheta(par);
        __syncthreads();
        if (nxt < n_batches) person_forward(nxt, par ^ 1);   // short: counts -> theta operands (eps came a batch ahead)
        __syncthreads();
        read_theta_ops();
        fetch_eps(nxt + G, par);
        if constexpr (GRAD) person_backward(bt, par);     // nobody waits for this: it overlaps the other waves' math
#pragma unroll
        for (int k = 0; k < 8; ++k) { cwA0[k] = cwB0[k]; cwA1[k] = cwB1[k]; }
    }

    // ================= workgroup reduction -> partial record =================
    float* out = p.partial + (size_t)blockIdx.x * p.lay.stride;
    {
        // 1PL/2PL: every cell without an observation contributed exactly log2(1 + 2^0) = 1 to s_log
        const float ll = (IRT == 3 ? kLn2 : -kLn2) * wave_total(s_log - (float)unobs);
        const float t_kl = wave_total(s_kl), t_q0 = wave_total(s_logq0), t_lp = wave_total(s_logp);
        const float t_no = wave_total(s_nobs);
        if (lane == 0) {
            wl.red[0] = ll; wl.red[1] = t_kl; wl.red[2] = t_q0; wl.red[3] = t_lp; wl.red[4] = 0.f; wl.red[5] = t_no;
            wl.red[6] = 0.f; wl.red[7] = 0.f;
        }
    }
    __syncthreads();
    if (tid < 8) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += wls[w].red[tid];
        out[tid] = (tid < 6) ? t : 0.f;
    }
    if constexpr (GRAD) {
        if (tid < 8 * A) {
            const int a = tid >> 3, k = tid & 7;
            float t = 0.f;
            for (int w = 0; w < nw; ++w) t += cl.tred[w][k][a];
            const int st = k >> 2, c = (k >> 1) & 1, ms = k & 1;
            out[p.lay.off_table + (st * 2 + c) * 2 * A + ms * A + a] = t;
        }
        // item gradients.  d LL/d a: lane (col i16, g) holds items 4 g + j of tile (u, t): cols a and 8 + a add up
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (IRT != 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = acc_ga[u][t][j] + dpp_f<0x128>(acc_ga[u][t][j]);
                        const int il = kMsSpan * q + 64 * u + 4 * (4 * g + j) + t;
                        if (i16 < A && il < I) out[p.lay.off_item + (size_t)i16 * p.lay.i_pad + il] = -v;
                    }
                }
                // d LL/d b (and d/d guess-logit): the lane's 8 persons per batch -> sum over the 4 lane groups
                float b = acc_b[u][t];
                b += __shfl_xor(b, 16);
                b += __shfl_xor(b, 32);
                const int il = kMsSpan * q + 64 * u + 4 * i16 + t;
                const int brow = IRT == 1 ? 0 : A;
                if (g == 0 && il < I) out[p.lay.off_item + (size_t)brow * p.lay.i_pad + il] = b;
                if constexpr (IRT == 3) {
                    float gg = acc_g[u][t];
                    gg += __shfl_xor(gg, 16);
                    gg += __shfl_xor(gg, 32);
                    if (g == 0 && il < I) out[p.lay.off_item + (size_t)(A + 1) * p.lay.i_pad + il] = gg;
                }
            }
    }
}

template <int IRT, bool GRAD, int RM>
static hipError_t launch_msplit_one(const ElboParams& p, int nw, int grid, hipStream_t s) {
    // more than 64 KB of dynamic LDS has to be opted into (once per kernel; gfx950 has 160 KB per CU)
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&msplit_kernel<IRT, GRAD, RM>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)msplit_lds_bytes(8));
        if (e != hipSuccess) return e;
        lds_opt_in = true;
    }
    hipLaunchKernelGGL((msplit_kernel<IRT, GRAD, RM>), dim3(grid), dim3(64 * nw), msplit_lds_bytes(nw), s, p);
    return hipGetLastError();
}
template <int RM>
static hipError_t launch_msplit_rm(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    if (irt == 1) return grad ? launch_msplit_one<1, true, RM>(p, nw, grid, s) : launch_msplit_one<1, false, RM>(p, nw, grid, s);
    if (irt == 2) return grad ? launch_msplit_one<2, true, RM>(p, nw, grid, s) : launch_msplit_one<2, false, RM>(p, nw, grid, s);
    return grad ? launch_msplit_one<3, true, RM>(p, nw, grid, s) : launch_msplit_one<3, false, RM>(p, nw, grid, s);
}

}  // namespace vibo
