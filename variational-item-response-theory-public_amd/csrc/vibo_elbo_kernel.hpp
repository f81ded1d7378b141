// vibo_elbo_kernel.hpp -- fused VIBO ELBO forward+backward kernel for gfx950 (MI355X).
//
// One persistent workgroup (NW waves) per CU-slot walks 64-person tiles of the response matrix.
//
//  loader   every wave owns 64/NW rows of each tile.  While it computes block s of the CURRENT tile it
//           has one step (4 x 1 KB of 16-byte loads = 20 VGPRs) of the NEXT tile in flight, then packs
//           it: each cell becomes ONE fp8 byte  w = +1 (correct) / -1 (wrong) / 0 (missing)  in the
//           other half of a double-buffered LDS tile, and the per-row counts (n_correct, n_observed)
//           the unconditional product of experts needs (models.py:596-629 collapses to those counts)
//           are wave-reduced.  Each response row is read from HBM exactly once.
//  phase B  the 1PL/2PL/3PL decode and its backward are three small GEMMs per (64 persons x 16 items)
//           block, all on the matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32), so the VALU only does
//           the elementwise logistic math:
//             L  [16p x 16i] = [theta|1] . [-a|b]^T            (K = A+1)      logits, log2 units
//             elementwise on the MFMA D layout (lane = (4 persons, 1 item)):
//                 masked Bernoulli log-lik (softplus form, reference clamp semantics), g = dLL/dL
//             dI [16i x 16c] += G^T . [theta|1]                (K = 64 persons) d LL / d item
//                 -- the D-layout registers of G ARE the A operand of this product: no shuffle
//             dT [16p x 16a] += G . [-a]                        (K = 16 items)  d LL / d theta
//                 -- needs G transposed: 16x16 tile through a per-wave LDS staging slab
//  epilogue per-wave partial dLL/dtheta -> LDS -> the wave that owns ability dim a pushes it through
//           the reparameterised sample and the product of experts into table-gradient accumulators,
//           adds the KL / (log q - log p) side, writes the [B,A] posterior outputs, and prepares
//           [theta|1] of the NEXT tile in LDS for all waves.
//
// All reductions have a fixed order for a fixed grid, so results are bitwise reproducible.
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
                                                const float eps, const int a) {
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
    d.eps = eps;
    return d;
}

// ---------------------------------------------------------------------------
// the kernel
// geometry (NW waves, SB 16-item blocks per wave, CMAX 64-lane float4 chunks per row):
//   (16,4,4) I <= 1024   (8,4,2) I <= 512   (4,8,2) I <= 304   (2,8,1) I <= 144
// LDS: code tile [2][64][lds_stride] | share [2][A+1][64] ([theta|valid] of a tile) |
//      stage [NW][16][20] (G-tile transpose) | counts [2][64] | ctab [4][2][A]
// ---------------------------------------------------------------------------
template <int A, int IRT, int NW, int SB, int CMAX, bool GRAD>
__global__ __launch_bounds__(NW * 64, 4) void elbo_kernel(const ElboParams p) {
    constexpr int KK = (A + 1 + 3) / 4;                   // K chunks of the logit GEMM ([theta|1] has A+1 columns)
    constexpr int DPW = (A + NW - 1) / NW;                // ability dims a wave owns in the epilogue
    constexpr int SROW = 20;                              // staging row stride (floats): conflict-free b32 writes
    constexpr int PS = 65;                                // person-row stride (floats) of share / red: spreads MFMA-layout reads over banks
    constexpr float kLoS = kLogitLo * kLog2e, kHiS = kLogitHi * kLog2e;   // clamp bounds in log2 units
    constexpr int STRIDE = code_tile_stride(NW);          // bytes per person row of the code tile (odd multiple of 16)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* share_base = reinterpret_cast<float*>(smem + 2 * p.lds_main);
    float* stage_base = share_base + 2 * (A + 1) * PS;
    uint32_t* counts_base = reinterpret_cast<uint32_t*>(stage_base + NW * 16 * SROW);
    float* ctab = reinterpret_cast<float*>(counts_base + 2 * kTilePersons);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gq = lane >> 4, nq = lane & 15;             // MFMA lane coordinates
    const int I = p.I;
    const int Ar = p.A;
    constexpr int stride = STRIDE;
    float* stage = stage_base + wave * 16 * SROW;

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
    float4v acc_item[SB];      // d LL/d item tiles: lane (g,c), reg r  <->  item 16*blk + 4g + r, column c
    float acc_t[DPW][8];       // table grads of the dims this wave owns: [set*4 + c*2 + {m,s}]
#pragma unroll
    for (int s = 0; s < SB; ++s) acc_item[s] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < DPW; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc_t[k][j] = 0.f;
    float s_log = 0.f, s_kl = 0.f, s_logq0 = 0.f, s_logp = 0.f, s_nobs = 0.f;

    // ---- loaders ------------------------------------------------------------
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
    // [theta | valid] of one tile -> share (each ability dim is produced by its owner wave; lane = person)
    auto make_share = [&](int tl, const uint32_t* cnts, float* share, const float (&epsv)[DPW]) {
        const long long grow = (long long)tl * kTilePersons + lane;
        const bool valid = grow < p.B;
        const uint32_t cnt = cnts[lane];
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int a = wave + NW * k;
            if (a < A) {
                float th = 0.f;
                if (a < Ar && valid) {
                    const PersonDim d = person_dim<A>(p, ctab, cnt, epsv[k], a);
                    th = d.amu + d.sig * d.eps;
                }
                share[a * PS + lane] = th;
            }
        }
        if (wave == NW - 1) share[A * PS + lane] = valid ? 1.f : 0.f;
    };
    auto load_eps = [&](int tl, float (&epsv)[DPW]) {
        const long long grow = (long long)tl * kTilePersons + lane;
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int a = wave + NW * k;
            epsv[k] = (a < Ar && grow < p.B) ? p.eps[grow * Ar + a] : 0.f;
        }
    };

    // ---- prologue: this block's first tile -> buffer 0 (no overlap) -----------
    constexpr int NSTEP = TileLoader<NW, CMAX>::NSTEP;
    float eps_cur[DPW], eps_nxt[DPW];
    load_eps(blockIdx.x, eps_cur);
    if (p.vec_ok) {
#pragma unroll 1
        for (int st = 0; st < NSTEP; ++st) {
            load_issue(blockIdx.x, st);
            load_commit(blockIdx.x, st, smem, counts_base);
        }
    } else {
        load_scalar(blockIdx.x, smem, counts_base);
    }
    __syncthreads();
    make_share(blockIdx.x, counts_base, share_base, eps_cur);

    int buf = 0;
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x, buf ^= 1) {
        unsigned char* codes = smem + buf * p.lds_main;                 // current tile
        unsigned char* codes_nxt = smem + (buf ^ 1) * p.lds_main;       // being filled for the next tile
        uint32_t* counts = counts_base + buf * kTilePersons;
        uint32_t* counts_nxt = counts_base + (buf ^ 1) * kTilePersons;
        const float* share = share_base + buf * (A + 1) * PS;
        float* share_nxt = share_base + (buf ^ 1) * (A + 1) * PS;
        float* red = reinterpret_cast<float*>(codes);                   // aliases the current tile after S2
        const int next_tile = tile + gridDim.x;
        const bool has_next = next_tile < p.n_tiles;
        __syncthreads();   // S1: codes / counts / share of the current tile complete

        if (has_next) load_eps(next_tile, eps_nxt);   // latency hidden under the item loop

        float4v acc_g[4];      // d LL/d theta tiles: lane (g,a), reg r <-> person 16pt+4g+r, dim a  (x log2e)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc_g[pt] = float4v{0.f, 0.f, 0.f, 0.f};

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
                // ---- per-block operands from the prepped item rows (L2/L1 resident, zero padded) ----
                const float* irow = p.item_prep + (size_t)(i0 + nq) * p.DP;          // item i0+n
                float bop[KK];                        // B of the logit GEMM: lane (g,n) = row[item n][4kk+g]
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) bop[kk] = (4 * kk + gq < p.DP) ? irow[4 * kk + gq] : 0.f;
                float gs = 0.f, om = 0.f;
                if constexpr (IRT == 3) {
                    gs = irow[A + 1];
                    om = irow[A + 2];
                }
                float ai[4];                          // B of the d-theta GEMM: lane (g,a) = -a'[item 4g+r][a]
                if constexpr (GRAD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ai[r] = (nq < A) ? p.item_prep[(size_t)(i0 + 4 * gq + r) * p.DP + nq] : 0.f;
                }
                float4v cur = float4v{0.f, 0.f, 0.f, 0.f};
                const unsigned char* cbase = codes + (4 * gq) * stride + i0 + nq;
#pragma unroll 1
                for (int pt = 0; pt < 4; ++pt) {
                    // logits of persons 16pt+4g+r (r = 0..3) x item i0+n:
                    //   A operand @ lane (g,i) = [theta|1][person 16pt+i][column 4kk+g]  (from the tile's share)
                    float4v L = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        const int c = 4 * kk + gq;
                        const float aop = (c <= A) ? share[c * PS + 16 * pt + nq] : 0.f;
                        L = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, bop[kk], L, 0, 0, 0);
                    }
                    float G[4], GG[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t cb = cbase[(16 * pt + r) * stride];
                        const float w = code_to_f32<0>(cb);
                        const float l = L[r];
                        float gl = 0.f;
                        if constexpr (IRT != 3) {
                            // ll = log sigmoid(w*l) = -softplus(u), u = -w*l.  Reference clamp (utils.py:46-49
                            // -> torch Bernoulli probs clamp): value clamped at +-kLogitLo, gradient exactly
                            // zero outside [-kLogitLo, kLogitHi].
                            const float l2 = med3(l, -kLoS, kHiS);
                            const float lc = fminf(l2, kLoS);
                            const float wg = (l == l2) ? w : 0.f;
                            const float eu = fast_exp2(-w * lc);
                            const float t = 1.0f + eu;
                            s_log = fmaf(fabsf(w), fast_log2(t), s_log);      // softplus(u)/ln2, masked
                            if constexpr (GRAD) gl = wg * (eu * fast_rcp(t));  // w * sigmoid(u)
                        } else {
                            const float e = fast_exp2(-fabsf(l));
                            const float rr = fast_rcp(1.0f + e);
                            const float er = e * rr;
                            const float sp = (l >= 0.f) ? rr : er;         // sigmoid(l)
                            const float sn = (l >= 0.f) ? er : rr;         // sigmoid(-l)
                            const float pr = fmaf(om, sp, gs);             // P(correct)  (models.py:765)
                            const float qr = om * sn;                      // P(wrong)
                            const float pc = med3(pr, kEps32, 1.0f - kEps32);
                            const float arg = (w > 0.f) ? pc : med3(qr, kEps32, 1.0f - kEps32);
                            s_log = fmaf(fabsf(w), fast_log2(arg), s_log);
                            if constexpr (GRAD) {
                                const float wl = (pr == pc) ? w : 0.f;     // clamp kills the gradient
                                const float common = wl * fast_rcp(arg) * om * sn;   // (x/p-(1-x)/(1-p)) (1-g) sig(-l)
                                gl = common * sp;                          // * d p / d logit
                                GG[r] = common * gs;                       // * d p / d guess-logit
                            }
                        }
                        G[r] = gl;
                    }
                    if constexpr (GRAD) {
                        // d LL/d item: A = G (its D layout is already A[i=item n][k=g]), B = [theta|1] of person 16pt+4g+r
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float bt = (nq <= A) ? share[nq * PS + 16 * pt + 4 * gq + r] : 0.f;
                            cur = __builtin_amdgcn_mfma_f32_16x16x4f32(G[r], bt, cur, 0, 0, 0);
                        }
                        if constexpr (IRT == 3) {
                            const float bgc = (nq == A + 1) ? 1.f : 0.f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) cur = __builtin_amdgcn_mfma_f32_16x16x4f32(GG[r], bgc, cur, 0, 0, 0);
                        }
                        // d LL/d theta: transpose the 16x16 G tile through the staging slab
#pragma unroll
                        for (int r = 0; r < 4; ++r) stage[(4 * gq + r) * SROW + nq] = G[r];
                        const float4v GT = *reinterpret_cast<const float4v*>(stage + nq * SROW + 4 * gq);
#define VIBO_GTH_ACC(Q)                                                                                       \
    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                             \
        acc_g[Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(GT[r], ai[r], acc_g[Q], 0, 0, 0);
                        if (pt == 0) { VIBO_GTH_ACC(0) }
                        else if (pt == 1) { VIBO_GTH_ACC(1) }
                        else if (pt == 2) { VIBO_GTH_ACC(2) }
                        else { VIBO_GTH_ACC(3) }
#undef VIBO_GTH_ACC
                    }
                }
                if constexpr (GRAD) {
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
        __syncthreads();   // S2: current code tile is dead; counts of the next tile are complete

        // ================= epilogue =========================================
        if constexpr (GRAD) {
            // partial d LL/d theta of this wave's items: lane (g,a), reg r <-> person 16pt+4g+r, dim a
            if (nq < A) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[(wave * A + nq) * PS + 16 * pt + 4 * gq + r] = acc_g[pt][r] * kLn2;   // rows carried log2 e
            }
        }
        if (has_next) make_share(next_tile, counts_nxt, share_nxt, eps_nxt);
        __syncthreads();       // S3: partial dLL/dtheta of all waves visible

        const long long grow = (long long)tile * kTilePersons + lane;
        const bool valid = grow < p.B;
        const uint32_t cnt = counts[lane];
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int a = wave + NW * k;
            if (a < Ar && valid) {
                const PersonDim d = person_dim<A>(p, ctab, cnt, eps_cur[k], a);
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
                    for (int w2 = 0; w2 < NW; ++w2) g0 += red[(w2 * A + a) * PS + lane];
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
#pragma unroll
        for (int k = 0; k < DPW; ++k) eps_cur[k] = eps_nxt[k];
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
        // item grads: MFMA tile of slot s: lane (g,c), reg r <-> item 16*(wave+NW*s) + 4g + r, column c
        int col = -1;                       // output column of [I][D] this lane's MFMA column maps to
        bool neg = false;
        if constexpr (IRT == 1) {
            if (nq == A) col = 0;                       // d/d b_i = sum_p g
        } else {
            if (nq < Ar) { col = nq; neg = true; }      // d/d a_ia = -sum_p g * theta_a
            else if (nq == A) col = Ar;                 // d/d b_i  =  sum_p g
            else if (IRT == 3 && nq == A + 1) col = Ar + 1;
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
