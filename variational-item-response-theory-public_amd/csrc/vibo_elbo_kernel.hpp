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
//  phase B  "person per lane": lane p owns person p of the tile (theta and
//           dLL/dtheta live in its registers); the NW waves split the items in
//           16-item blocks; item parameters are wave-uniform and arrive through
//           scalar loads.  Per term: logit -> masked Bernoulli log-lik (softplus
//           form; the reference's probability clamp is a wave-uniform rare path)
//           -> g = dLL/dlogit -> dLL/dtheta += g * dlogit/dtheta (in lane).
//           dLL/ditem = G^T . [theta | 1] is a contraction over the 64 PERSONS of
//           the tile, i.e. over lanes: the 16 g-registers of a block are transposed
//           inside each row of 16 lanes (DPP butterfly) into the A-operand layout of
//           v_mfma_f32_16x16x4_f32 and 16 MFMAs (exact fp32) accumulate the
//           [16 items x (A+2)] gradient tile; the matrix pipe runs beside the VALU.
//  epilogue per-wave partial dLL/dtheta -> LDS -> each wave owns ability dims
//           a = wave (mod NW): backward through the reparameterised sample and the
//           product of experts into per-lane table-gradient accumulators; KL /
//           log q - log p side; [B,A] posterior outputs.
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

typedef float float4v __attribute__((ext_vector_type(4)));

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

// Software-pipelined tile loader.  A wave owns RPW rows of every 64-person tile; they are fetched
// in NSTEP steps of 4 (row, 64-lane chunk) units = 20 VGPRs, one step per 16-item compute block, so
// each wave always has ~5 KB of HBM loads in flight while it computes (16 waves/CU -> 80 KB/CU).
template <int NW, int CMAX>
struct TileLoader {
    static constexpr int RPW = kTilePersons / NW;   // rows per wave
    static constexpr int UPS = 4;                    // (row, chunk) units per step
    static constexpr int RPS = UPS / CMAX;           // whole rows per step (CMAX in {1,2,4})
    static constexpr int NSTEP = RPW / RPS;
    float4 x[UPS];
    uint32_t m[UPS];

    template <int MK>
    __device__ __forceinline__ void issue(const ElboParams& p, const int tile, const int wave, const int lane,
                                          const int step) {
        const int n4 = p.I >> 2;
#pragma unroll
        for (int jj = 0; jj < RPS; ++jj) {
            const int r = wave * RPW + step * RPS + jj;
            const long long grow = (long long)tile * kTilePersons + r;
            if (grow < p.B) {
                const long long src = p.row_index ? p.row_index[grow] : grow;
                const float4* rp = reinterpret_cast<const float4*>(p.response + src * p.resp_stride);
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    const int q = lane + 64 * c;
                    if (q < n4) {
                        x[jj * CMAX + c] = rp[q];
                        m[jj * CMAX + c] = load_mask4<MK>(p, src, q);
                    }
                }
            }
        }
    }

    __device__ __forceinline__ void commit(const ElboParams& p, const int tile, const int wave, const int lane,
                                           const int step, unsigned char* codes, uint32_t* counts) {
        const int n4 = p.I >> 2;
        const int tail_words = ((16 - (p.I & 15)) & 15) / 4;
#pragma unroll
        for (int jj = 0; jj < RPS; ++jj) {
            const int r = wave * RPW + step * RPS + jj;
            const long long grow = (long long)tile * kTilePersons + r;
            const bool live = grow < p.B;
            uint32_t* dst = reinterpret_cast<uint32_t*>(codes + r * p.lds_stride);
            int packed = 0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                const int q = lane + 64 * c;
                if (q < n4) dst[q] = live ? pack_codes4(x[jj * CMAX + c], m[jj * CMAX + c], packed) : 0u;
            }
            if (lane < tail_words) dst[n4 + lane] = 0u;     // pad the row to a multiple of 16 items
            packed = wave_sum63(packed);
            if (lane == 63) counts[r] = (uint32_t)packed;
        }
    }
};

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
        if (lane < ((16 - (p.I & 15)) & 15)) dst[p.I + lane] = 0;        // pad the row to a multiple of 16 items
        packed = wave_sum63(packed);
        if (lane == 63) counts[r] = (uint32_t)packed;
    }
}

// ---------------------------------------------------------------------------
// per-person posterior for one ability dim (product of experts on counts)
// ---------------------------------------------------------------------------
template <int A>
__device__ __forceinline__ PersonDim person_dim(const ElboParams& p, const float* ctab, const uint32_t cnt,
                                                const long long grow, const bool valid, const int a) {
    PersonDim d;
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
// 16x16 transpose of a[16] inside every row of 16 lanes:  a'[t] @ pos m  =  a[m] @ pos t
// (4 butterfly stages; stage S swaps bit S of the lane position with bit S of the register index)
// ---------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ float row_xor(float v) {
    if constexpr (S == 1) return dpp_f<0xb1>(v);            // quad_perm [1,0,3,2]
    else if constexpr (S == 2) return dpp_f<0x4e>(v);       // quad_perm [2,3,0,1]
    else if constexpr (S == 8) return dpp_f<0x128>(v);      // row_ror 8
    else return dpp_f<0x1b>(dpp_f<0x141>(v));               // row_half_mirror (^7) then quad_perm [3,2,1,0] (^3) = ^4
}

template <int S>
__device__ __forceinline__ void transpose_stage(float (&a)[16], const bool bit) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if ((r & S) == 0) {
            const float lo = a[r], hi = a[r | S];
            const float recv = row_xor<S>(bit ? lo : hi);
            a[r] = bit ? recv : lo;
            a[r | S] = bit ? hi : recv;
        }
    }
}

__device__ __forceinline__ void transpose16(float (&a)[16], const int lane) {
    transpose_stage<8>(a, (lane & 8) != 0);
    transpose_stage<4>(a, (lane & 4) != 0);
    transpose_stage<2>(a, (lane & 2) != 0);
    transpose_stage<1>(a, (lane & 1) != 0);
}

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
// geometry (NW waves, SB 16-item blocks per wave, CMAX 64-lane float4 chunks per row):
//   (16,4,4) I <= 1024   (8,4,2) I <= 512   (4,8,2) I <= 304   (2,8,1) I <= 144
template <int A, int IRT, int NW, int SB, int CMAX, bool GRAD>
__global__ __launch_bounds__(NW * 64, 4) void elbo_kernel(const ElboParams p) {
    constexpr int AP = (A + 1) / 2;                       // float2 pairs
    constexpr int DPW = (A + NW - 1) / NW;                // ability dims a wave owns in the epilogue
    constexpr float kLoS = kLogitLo * kLog2e, kHiS = kLogitHi * kLog2e;   // clamp bounds in log2 units

    // LDS: two code tiles [64][lds_stride] (double buffer) | counts [2][64] | ctab [4][2][A]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* counts_base = reinterpret_cast<uint32_t*>(smem + 2 * p.lds_main);
    float* ctab = reinterpret_cast<float*>(counts_base + 2 * kTilePersons);

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
    float4v acc_item[SB];      // MFMA accumulators: lane (g,n), reg r  <->  item 16*blk + 4g + r, column n
    float acc_t[DPW][8];       // table grads of the dims this wave owns: [set*4 + c*2 + {m,s}]
#pragma unroll
    for (int s = 0; s < SB; ++s) acc_item[s] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < DPW; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc_t[k][j] = 0.f;
    float s_log = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_nobs = 0.f;

    // ---- prologue: this block's first tile -> buffer 0 (no overlap) -----------
    TileLoader<NW, CMAX> ld;
    auto load_issue = [&](int tl, int step) {
        if (p.mask_dtype == 0) ld.template issue<0>(p, tl, wave, lane, step);
        else if (p.mask_dtype == 1) ld.template issue<1>(p, tl, wave, lane, step);
        else ld.template issue<2>(p, tl, wave, lane, step);
    };
    auto load_commit = [&](int tl, int step, unsigned char* cd, uint32_t* ct) {
        ld.commit(p, tl, wave, lane, step, cd, ct);
    };
    auto load_scalar = [&](int tl, unsigned char* cd, uint32_t* ct) {
        if (p.mask_dtype == 0) phase_a_scalar<NW, 0>(p, tl, wave, lane, cd, ct);
        else if (p.mask_dtype == 1) phase_a_scalar<NW, 1>(p, tl, wave, lane, cd, ct);
        else phase_a_scalar<NW, 2>(p, tl, wave, lane, cd, ct);
    };
    constexpr int NSTEP = TileLoader<NW, CMAX>::NSTEP;
    if (p.vec_ok) {
#pragma unroll 1
        for (int st = 0; st < NSTEP; ++st) {
            load_issue(blockIdx.x, st);
            load_commit(blockIdx.x, st, smem, counts_base);
        }
    } else {
        load_scalar(blockIdx.x, smem, counts_base);
    }

    int buf = 0;
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x, buf ^= 1) {
        unsigned char* codes = smem + buf * p.lds_main;                 // current tile
        unsigned char* codes_nxt = smem + (buf ^ 1) * p.lds_main;       // being filled for the next tile
        uint32_t* counts = counts_base + buf * kTilePersons;
        uint32_t* counts_nxt = counts_base + (buf ^ 1) * kTilePersons;
        float* red = reinterpret_cast<float*>(codes);                   // aliases the current tile after S2
        const int next_tile = tile + gridDim.x;
        const bool has_next = next_tile < p.n_tiles;
        __syncthreads();   // S1: current tile (codes + counts) complete; previous epilogue done with its buffer

        // ================= phase B ==========================================
        const long long grow = (long long)tile * kTilePersons + lane;
        const bool valid = grow < p.B;
        const uint32_t cnt = counts[lane];

        float th[2 * AP];
#pragma unroll
        for (int a = 0; a < 2 * AP; ++a) th[a] = 0.f;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            if (a < Ar) {
                const PersonDim d = person_dim<A>(p, ctab, cnt, grow, valid, a);
                th[a] = valid ? (d.amu + d.sig * d.eps) : 0.f;
            }
        }
        float th_sum = 0.f;   // 1PL: logit = sum_a theta_a + b (kept in log2 units like the prepped b)
#pragma unroll
        for (int a = 0; a < A; ++a) th_sum += th[a];
        th_sum *= kLog2e;
        float2v th2[AP];
#pragma unroll
        for (int j = 0; j < AP; ++j) th2[j] = float2v{th[2 * j], th[2 * j + 1]};

        // B operand of the item-gradient MFMAs: column n of [theta | 1], transposed so that in MFMA t
        // lane (g,n) supplies column n of person 16g+t
        float bt[16];
        if constexpr (GRAD) {
#pragma unroll
            for (int n = 0; n < 16; ++n) bt[n] = 0.f;
            if constexpr (IRT == 1) {
                bt[0] = valid ? 1.f : 0.f;
            } else {
#pragma unroll
                for (int a = 0; a < A; ++a) bt[a] = th[a];
                bt[A] = valid ? 1.f : 0.f;
            }
            transpose16(bt, lane);
        }

        float2v gth2[AP];     // d LL / d theta of this wave's items, in units of log2e (2PL/3PL)
#pragma unroll
        for (int j = 0; j < AP; ++j) gth2[j] = float2v{0.f, 0.f};
        float gth_sum = 0.f;  // 1PL

        const unsigned char* my_codes = codes + lane * stride;
#pragma unroll 1
        for (int s = 0; s < SB; ++s) {
            const int b16 = wave + NW * s;          // 16-item block index (round-robin over waves)
            const int i0 = b16 * 16;
            const bool prefetch = has_next && p.vec_ok && s < NSTEP;
            if (prefetch) {
                load_issue(next_tile, s);            // HBM loads of the next tile fly under this block's math
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i0 < I) {
            const uint4 cw = *reinterpret_cast<const uint4*>(my_codes + i0);
            const uint32_t cwa[4] = {cw.x, cw.y, cw.z, cw.w};
            float g[16], gg[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                g[j] = 0.f;
                if constexpr (IRT == 3) gg[j] = 0.f;
            }
#pragma unroll
            for (int wq = 0; wq < 4; ++wq) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = wq * 4 + jj;
                    const int i = i0 + j;          // items >= I are zero-padded: code 0, item row 0
                    {
                        const uint32_t word = cwa[wq];
                        float w;
                        if (jj == 0) w = code_to_f32<0>(word);
                        else if (jj == 1) w = code_to_f32<1>(word);
                        else if (jj == 2) w = code_to_f32<2>(word);
                        else w = code_to_f32<3>(word);
                        const const_f32_ptr ip = as_constant(p.item_prep + (size_t)i * p.DP);   // uniform -> s_load
                        // ---- logit, in log2 units (item rows are pre-scaled by log2 e) ----
                        float l;
                        if constexpr (IRT == 1) {
                            l = ip[0] + th_sum;
                        } else if constexpr (A == 1) {
                            l = fmaf(ip[0], th[0], ip[1]);
                        } else {
                            float2v acc = float2v{ip[A], 0.f};
#pragma unroll
                            for (int q = 0; q < AP; ++q) acc = float2v{ip[2 * q], ip[2 * q + 1]} * th2[q] + acc;
                            l = acc.x + acc.y;
                        }
                        float gl = 0.f;   // d ll / d logit (natural units)
                        if constexpr (IRT != 3) {
                            // ll = log sigmoid(w*l) = -softplus(u), u = -w*l.  The reference clamps the
                            // probability to [eps32, 1-eps32] (utils.py:46-49 -> torch Bernoulli): value
                            // clamped at +-kLogitLo, gradient exactly zero outside [-kLogitLo, kLogitHi]
                            // (branch-free: a rare-path branch here makes hipcc spill the item rows).
                            const float l2 = med3(l, -kLoS, kHiS);
                            const float lc = fminf(l2, kLoS);
                            const float wg = (l == l2) ? w : 0.f;
                            const float eu = fast_exp2(-w * lc);
                            const float t = 1.0f + eu;
                            s_log = fmaf(fabsf(w), fast_log2(t), s_log);      // softplus(u)/ln2, masked
                            if constexpr (GRAD) gl = wg * (eu * fast_rcp(t));  // w * sigmoid(u)
                        } else {
                            const float guess = ip[A + 1], omg = ip[A + 2];
                            const float e = fast_exp2(-fabsf(l));
                            const float r = fast_rcp(1.0f + e);
                            const float er = e * r;
                            const float sp = (l >= 0.f) ? r : er;          // sigmoid(l)
                            const float sn = (l >= 0.f) ? er : r;          // sigmoid(-l)
                            const float pr = fmaf(omg, sp, guess);         // P(correct)  (models.py:765)
                            const float qr = omg * sn;                     // P(wrong)
                            const float pc = med3(pr, kEps32, 1.0f - kEps32);
                            const float arg = (w > 0.f) ? pc : med3(qr, kEps32, 1.0f - kEps32);
                            s_log = fmaf(fabsf(w), fast_log2(arg), s_log);
                            if constexpr (GRAD) {
                                const float wl = (pr == pc) ? w : 0.f;     // clamp kills the gradient
                                const float common = wl * fast_rcp(arg) * omg * sn;   // (x/p-(1-x)/(1-p)) (1-g) sig(-l)
                                gl = common * sp;                          // * d p / d logit
                                gg[j] = common * guess;                    // * d p / d guess-logit
                            }
                        }
                        if constexpr (GRAD) {
                            g[j] = gl;
                            if constexpr (IRT == 1) {
                                gth_sum += gl;
                            } else if constexpr (A == 1) {
                                gth2[0].x = fmaf(gl, ip[0], gth2[0].x);
                            } else {
                                const float2v g2 = float2v{gl, gl};
#pragma unroll
                                for (int q = 0; q < AP; ++q)
                                    gth2[q] = g2 * float2v{ip[2 * q], ip[2 * q + 1]} + gth2[q];
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the scalar item-row loads of later groups below
            }
            if constexpr (GRAD) {
                // dLL/ditem[16 x 16] += G^T[16 items x 64 persons] . [theta|1][64 persons x 16]
                transpose16(g, lane);
                float4v cur = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 16; ++t) cur = __builtin_amdgcn_mfma_f32_16x16x4f32(g[t], bt[t], cur, 0, 0, 0);
                if constexpr (IRT == 3) {
                    transpose16(gg, lane);
                    const float bg = ((lane & 15) == A + 1) ? 1.f : 0.f;   // guess-logit column
#pragma unroll
                    for (int t = 0; t < 16; ++t) cur = __builtin_amdgcn_mfma_f32_16x16x4f32(gg[t], bg, cur, 0, 0, 0);
                }
#pragma unroll
                for (int ss = 0; ss < SB; ++ss)
                    if (ss == s) acc_item[ss] += cur;
            }
            }
            if (prefetch) load_commit(next_tile, s, codes_nxt, counts_nxt);
        }
        if (has_next) {
            if (p.vec_ok) {
                if constexpr (NSTEP > SB) {
#pragma unroll 1
                    for (int st = SB; st < NSTEP; ++st) {
                        load_issue(next_tile, st);
                        load_commit(next_tile, st, codes_nxt, counts_nxt);
                    }
                }
            } else {
                load_scalar(next_tile, codes_nxt, counts_nxt);
            }
        }
        __syncthreads();   // S2: every wave is done reading the current code tile

        // ================= epilogue =========================================
        if constexpr (GRAD) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                float gv;
                if constexpr (IRT == 1) gv = gth_sum;
                else gv = ((a & 1) ? gth2[a >> 1].y : gth2[a >> 1].x) * kLn2;   // item rows carried log2 e
                red[(wave * A + a) * 64 + lane] = gv;
            }
            __syncthreads();   // S3: partial dLL/dtheta of all waves visible
        }
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int a = wave + NW * k;
            if (a < Ar && valid) {
                const PersonDim d = person_dim<A>(p, ctab, cnt, grow, valid, a);
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
                    float g0 = 0.f;                                  // d LL / d theta_0[a], all items
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) g0 += red[(w2 * A + a) * 64 + lane];
                    const float h = 0.5f * d.sig * d.eps;            // d theta / d logvar
                    float gmu[2], glv[2];
                    gmu[0] = g0;
                    glv[0] = g0 * h;
                    if (p.reg_mode == 0) {        // analytic KL  (utils.py:85-88)
                        gmu[1] = d.amu;
                        glv[1] = -0.5f * (1.0f - evar);
                    } else {                      // log q0(theta0) - log p(theta0)  (models.py:433-436)
                        gmu[1] = theta0;
                        glv[1] = theta0 * h - 0.5f;
                    }
                    const float nn[2] = {d.n0, d.n1};
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float tau = ctab[(CT_TAU * 2 + c) * A + a];
                        const float te = ctab[(CT_TE * 2 + c) * A + a];
                        const float mm = ctab[(CT_M * 2 + c) * A + a];
                        const float nl = nn[c] * d.inv_lam;
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            acc_t[k][st * 4 + c * 2 + 0] = fmaf(gmu[st] * nl, tau, acc_t[k][st * 4 + c * 2 + 0]);
                            const float g_tau = nl * (gmu[st] * (mm - d.amu) - glv[st]);
                            acc_t[k][st * 4 + c * 2 + 1] = fmaf(-g_tau, te, acc_t[k][st * 4 + c * 2 + 1]);
                        }
                    }
                }
            }
        }
        if (wave == 0 && valid) s_nobs += (float)(cnt & 0xffffu);
    }
    __syncthreads();

    // ================= block-level reduction -> partial record ===============
    float* out = p.partial + (size_t)blockIdx.x * p.lay.stride;
    float* scr = reinterpret_cast<float*>(smem);   // [NW][8]
    {
        const float ll = (IRT != 3) ? -(kLn2 * s_log) : (kLn2 * s_log);
        const float vals[6] = {ll, s_kl, s_logq0, s_logp, 0.f, s_nobs};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float t = wave_total(vals[k]);
            if (lane == 0) scr[wave * 8 + k] = t;
        }
    }
    __syncthreads();
    if (tid < 8) {
        float t = 0.f;
        if (tid < 6)
            for (int w = 0; w < NW; ++w) t += scr[w * 8 + tid];
        out[tid] = t;
    }
    if constexpr (GRAD) {
        // table grads: each ability dim is owned by exactly one wave
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int a = wave + NW * k;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = wave_total(acc_t[k][j]);
                if (lane == 0 && a < Ar) {
                    const int st = j >> 2, c = (j >> 1) & 1, ms = j & 1;
                    out[p.lay.off_table + (st * 2 + c) * 2 * Ar + ms * Ar + a] = t;
                }
            }
        }
        // item grads: MFMA tile of slot s: lane (g,n), reg r <-> item 16*(wave+NW*s) + 4g + r, column n
        const int n = lane & 15, gq = lane >> 4;
        int col = -1;                       // output column of [I][D] this lane's MFMA column maps to
        bool neg = false;
        if constexpr (IRT == 1) {
            if (n == 0) col = 0;
        } else {
            if (n < Ar) { col = n; neg = true; }     // d/d a_ia = -sum_p g * theta_a
            else if (n == A) col = Ar;               // d/d b_i  =  sum_p g
            else if (IRT == 3 && n == A + 1) col = Ar + 1;
        }
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            const int i0 = (wave + NW * s) * 16;
            if (i0 < I && col >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc_item[s][r];
                    out[p.lay.off_item + col * p.lay.i_pad + i0 + 4 * gq + r] = neg ? -v : v;
                }
            }
        }
    }
}

}  // namespace vibo
