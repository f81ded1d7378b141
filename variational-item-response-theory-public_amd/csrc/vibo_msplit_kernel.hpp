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

#ifdef VIBO_MS_TIMING
// development build (make TIMING=1): shader-clock time per phase of the batch loop, summed per wave
constexpr int kMsTSlots = 32;
__device__ long long g_ms_timing[1024 * 8 * kMsTSlots];
#define MS_T(i) { __builtin_amdgcn_sched_barrier(0); long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
                  __builtin_amdgcn_sched_barrier(0); tacc[i] += now_ - tlast; tlast = now_; }
#define MS_WAITV() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// one-shot marks (prologue 16.., end code 24..): shader-clock cycles since the wave's entry / since the loop's end
#define MS_P(i, base) { __builtin_amdgcn_sched_barrier(0); long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
                  __builtin_amdgcn_sched_barrier(0); tacc[i] = now_ - (base); }
#else
#define MS_T(i)
#define MS_WAITV()
#define MS_P(i, base)
#endif

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
    _Float16 tr[2][2][32 * 16];   // [hi | lo][M-tile][row = 16 (t & 1) + i16][16 persons]: g pieces of one 32-item K-tile
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
    float tacc[2][12][256];       // [batch parity][k][(person, dim) pair]: running sums of the pair's table-gradient terms
                                  // (k < 8) and of its kl | logq0 | logp | nobs terms: one LDS add per value and batch
};
constexpr int kMsMF = VIBO_MAX_FLOWS;
struct alignas(16) MsFlowLds {      // planar flows only (appended to the dynamic LDS of the FLOWS instantiations)
    float fpar[kMsMF][2][8];        // uhat | w per ability dim
    float fsc[kMsMF][2];            // b, w.uhat
    float tps[2][kMsMF][2][kMsRows];   // [batch parity][flow][tanh | psi][person]: forward state kept for the backward
    float lacc[2][kMsRows];         // running sums of the persons' log|det| terms
    float facc[2][4][2][kMsMF][3][8];   // [batch parity][slot][set: LL | REG][flow][uhat | w | b][dim]: running flow-parameter gradients
};
// 3PL only (behind the flow block): the guess probabilities of a wave's 128 items, [u-step][tile t][chunk i16] -- they used to
// sit in 16 registers per lane (guess and 1 - guess of its 8 items) next to the 8 guess-gradient accumulators, which is what
// pushed every 3PL instantiation with gradients over the 256-register budget (8..136 B of scratch per lane)
constexpr int kMsGuessFloats = 2 * 4 * 16;
// XM == 3 only (behind the guess block): per wave the experts of its 128 items -- [u-step][tile t][chunk i16] x (tau | mu tau of a
// wrong answer, tau | mu tau of a right one) -- and its share of the batch's per-person sums [lam | s][row]
constexpr int kMsFuseFloats = 2 * 4 * 16 * 4 + 2 * kMsRows;
inline size_t msplit_lds_bytes(int nw, bool flows = false, bool guess = false, bool fuse = false) {
    return sizeof(MsCommonLds) + (size_t)nw * sizeof(MsWaveLds) + (flows ? sizeof(MsFlowLds) : 0) +
           (guess ? (size_t)nw * kMsGuessFloats * sizeof(float) : 0) + (fuse ? (size_t)nw * kMsFuseFloats * sizeof(float) : 0);
}
// sum over the 8 consecutive lanes that hold one person's ability dims (every lane of the group gets it)
__device__ __forceinline__ float ms_group_sum(float v) {
    v += dpp_f<0xb1>(v);                           // quad_perm [1,0,3,2]
    v += dpp_f<0x4e>(v);                           // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);                          // row_half_mirror
    return v;
}

__device__ __forceinline__ half2v pkrtz(float a, float b) {
    return __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// x = hi + lo, both f16 (round toward zero): |x - hi - lo| <= 2^-22 |x|
__device__ __forceinline__ void split16(float x, _Float16& hi, _Float16& lo) {
    const half2v h = pkrtz(x, 0.f);
    hi = h[0];
    lo = pkrtz(x - (float)h[0], 0.f)[0];
}
// running sum in LDS, one owner lane per address: plain read-modify-write (measured: 12 ds_add_f32 per batch and owner
// lane instead cost the whole kernel 0.29 ms of 1.46 -- the LDS serialises its float atomics)
__device__ __forceinline__ void lds_add(float* p, float v) { *p += v; }
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
// FLOWS: planar flows on the ability sample (flows.py:21-66, models.py:342-348) in the (person, dim) lanes.
// NW8: the workgroup has exactly 8 waves (897..1024 items, the benchmark's width): slot ownership and the sums over the
// waves' LDS records are compile-time.  XM: which hooks the launch uses (0: none -- those branches do not exist in the code;
// 1: panel / conditional: p.row_cnt, p.pre_stats, p.post_coef, p.primary == 0, p.panel_count; 2: p.given_post / p.given_grad;
// 3: conditional posterior with the experts' sums formed HERE (one panel, ability_dim 1, fp32 rows): p.cond_table, p.codes_out, p.post_coef).
// blockDim.x = 64 nw, dynamic LDS = msplit_lds_bytes(nw, FLOWS).
// The kernel runs at 2 waves per SIMD, where a wave issues at most one instruction per ~5 cycles whatever its type: every
// scalar instruction, branch and spill reload in the batch loop costs as much as a vector instruction (round 3: the loop
// lost ~500 of its ~2 300 executed instructions per wave and batch this way).
template <int IRT, bool GRAD, int RM, bool FLOWS, bool NW8, int XM>
__global__ __launch_bounds__(512, 2) void msplit_kernel(const ElboParams p) {
    // XM (hook mode): 0 = none; 1 = the panel / conditional hooks (p.row_cnt, p.pre_stats, p.post_coef, p.primary, p.panel_count);
    // 2 = the caller-supplied posterior read / written by the slot lanes themselves (p.given_post, p.given_grad; one panel).
    // Two modes instead of one EXTRA flag (round 4): each carries the other's pointers, branches and -- the given mode's expf /
    // logvar loads -- spilled registers no longer.
    // 3 (round 6) = the conditional posterior q(theta | responses, items) (models.py:664-710) of ONE panel at ability_dim 1 with its
    // first pass folded in: the lanes that pack a row's cells also gather the experts their codes select (tau, mu tau of the item's
    // wrong / right row of p.cond_table, kept in LDS) and the 16 lanes of a row group add them up; the slot lanes take lam and s from
    // those sums instead of from cond_pre_kernel's pre_stats, and the rows' 1-byte cell codes leave through p.codes_out for the
    // table-gradient pass.  One 5 B/cell stream instead of cond_pre's 5 + 1 and this kernel's 1.
    constexpr bool EXTRA = XM != 0, XCOND = XM == 1, XGIVEN = XM == 2, XFUSE = XM == 3;
    constexpr bool XCOEF = XCOND || XFUSE;        // the backward hands per-person coefficients to the table-gradient pass
    static_assert(!XFUSE || (RM != 2 && !FLOWS), "the fused conditional mode reads fp32 rows and has no flow block");
    // The flow instantiations that spilled (3PL, the hook modes, gathered rows) form their LDS addresses in the slot code
    // instead of keeping them loop-invariant in registers: see backward_slot.  (The others -- 2PL / 1PL flows on rows in
    // order, no spills -- measured 2 % slower with the same pins, so they keep the hoisted addresses.)
    constexpr bool kPinFlowAddr = FLOWS && (IRT == 3 || XM != 0 || RM == 1);
    constexpr bool CODES = RM == 2;
    constexpr int R = kMsRows;
    constexpr float kLoS = kLogitLo * kLog2e, kHiS = kLogitHi * kLog2e;
    // fp8 (e4m3) look-up tables of the pack.  fp32 rows: selector {missing, wrong, -, right}; cell codes: {wrong, right, missing}.
    // Every model carries -w (the exponent is -w x logit: one VOP2 multiply).
    constexpr uint32_t kLutFp32 = 0xB8003800u;
    constexpr uint32_t kLutCode = 0x0000B838u;
    extern __shared__ __attribute__((aligned(16))) unsigned char ms_smem[];
    MsCommonLds& cl = *reinterpret_cast<MsCommonLds*>(ms_smem);
    MsWaveLds* wls = reinterpret_cast<MsWaveLds*>(ms_smem + sizeof(MsCommonLds));

#ifdef VIBO_MS_TIMING
    long long t_entry, t_real_entry;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry), "=s"(t_real_entry) :: "memory");
    long long tacc[kMsTSlots];
    for (int k = 0; k < kMsTSlots; ++k) tacc[k] = 0;
    long long tlast;
#endif
    const int tid = threadIdx.x;
    insitu_enter(p.insitu);
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = NW8 ? 8 : (int)(blockDim.x >> 6);
    MsWaveLds& wl = wls[q];
    MsFlowLds& fl = *reinterpret_cast<MsFlowLds*>(ms_smem + sizeof(MsCommonLds) + (size_t)nw * sizeof(MsWaveLds));   // (FLOWS only)
    // Panel mode in ONE launch (rows of more than 1024 items, EXTRA only): p.panel_count 1024-item panels x G workgroups each,
    // workgroup blockIdx.x = G-slot * panel_count + panel -- the panels of a row batch start side by side (one DRAM page run),
    // every workgroup builds its operand image once and streams ceil(batches / G) batches of ITS panel, and the records land
    // where the per-panel launches of rounds 1-4 put them (panel-major).  Those launches paid the kernel's one-shot prologue /
    // end code (~18 us) and a 12.2 -> 13-round quantisation once per panel of a 100 000-row call.
    int wg = (int)blockIdx.x, G = (int)gridDim.x, item0 = p.item0, I = p.I, rec = (int)blockIdx.x;
    bool primary = XCOND ? p.primary != 0 : true;
    float* post_coef = XCOEF ? p.post_coef : nullptr;
    if constexpr (XCOND) {
        if (p.panel_count > 1) {
            const int panel = wg % p.panel_count;
            wg /= p.panel_count; G /= p.panel_count;
            item0 = panel * 1024;
            I = min(p.I_total - item0, 1024);
            primary = panel == 0;
            if (post_coef) post_coef += (size_t)panel * p.B * 4 * p.A;
            rec = panel * G + wg;
        }
    }
    const int A = p.A;
    const int n4 = (I + 3) >> 2;
    const int i16 = lane & 15, g = lane >> 4;

    const int ed = lane & 7;
    // (the planner keeps num_person <= 2^31 - 2^16: row numbers and batch counters are 32-bit)
    const int n_batches = (int)(((long long)p.B + R - 1) / R);
    float4 x[CODES ? 1 : 8];                        // [2 j + u]: person j of the half, chunk u
    uint32_t m[8];
    float4 x2[CODES ? 1 : 8];                       // the first batch's M-tile 1 (requested with M-tile 0 at the top of the kernel: dead
    uint32_t m2[8];                                 //  once the prologue has packed it)
    // Gathered rows: the row numbers of a batch are needed before its first row load can go out.  They are fetched TWO batches
    // ahead (ridx_n, a whole iteration before their use) -- fetched right in front of the row loads they feed, every batch
    // waited a memory round trip for them at the top of the loop (gathered rows: +22 % on fp32 rows, +60 % on cell codes).
    int ridx[8], ridx_n[8];
    auto fetch_idx = [&](const int bt, int (&dst)[8]) __attribute__((always_inline)) {
        if constexpr (RM != 0) {
            if (!p.row_index) return;
            const int row0 = bt * R;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = row0 + 4 * g + (k & 3) + 16 * (k >> 2);
                const int rc = min(row, p.B - 1);
                dst[k] = bt < n_batches ? (int)p.row_index[rc] : 0;
            }
        }
    };
    // Row loads.  The unit is a "half": the lane's 4 persons of one M-tile x BOTH chunks of the wave's item span, so
    // that a wave asks for 512 contiguous bytes of a response row (128 of its mask row) back to back and neighbouring
    // waves touch a row at the same time: segment-edge cache lines are fetched once (units of 32 rows x 64 items left them
    // to be re-fetched half a batch later: 1.34x the algorithmic HBM traffic).
    // In-order rows go through buffer loads: one resource per batch (scalar registers) whose record limit ends at the last
    // row of the matrix (rows past the end read as zeros), per-lane offsets shared by all rows of the lane and the row step
    // as a scalar offset -- no per-row address registers, and no branch in the load sequence: without a mask the mask
    // resource has no records (its loads return zeros) and the cells are switched on when they are packed (`fillw`).
    // Gathered rows compute their addresses at the load.  Chunks past the row's end read its last chunk; both cases are
    // masked when the cells are packed.
    // (Round 6 measured the lane's two chunks ADJACENT -- 2 i16 and 2 i16 + 1, the two mask words one 8-byte load: 24 load
    //  instructions per batch instead of 32, 8 instead of 16 on cell codes.  Cell codes 670 -> 657 us per 1M x 1k, but fp32 rows
    //  942-957 -> 972 us: every 16-byte response load then covers 512 bytes of a row at half density.  Not adopted.)
    const int cc0 = min(32 * q + i16, n4 - 1), cc1 = min(32 * q + 16 + i16, n4 - 1);
    const unsigned rstride4 = (unsigned)p.resp_stride * 4u, mstride = (unsigned)p.mask_stride;      // bytes per row
    const unsigned mvo0 = 4u * g * mstride + 4u * cc0, mvo1 = 4u * g * mstride + 4u * cc1;
    const unsigned rvo0 = 4u * g * rstride4 + 16u * cc0, rvo1 = 4u * g * rstride4 + 16u * cc1;
    const bool have_mask = CODES || p.mask_dtype == 0;        // (wave-uniform)
    const uint32_t fillw = have_mask ? 0u : 0x01010101u;
    struct RowSrc { __amdgpu_buffer_rsrc_t r, m, c; };
    // XM == 3: where the batch's cell codes go (p.codes_out, minibatch order; no records without it or past the last batch: the
    // stores fall away), and the lane's offsets there -- a chunk past the row's end gets an offset past every record limit
    const unsigned cstride = XFUSE ? (unsigned)p.codes_stride : 0u;
    const unsigned cvo0 = (32 * q + i16) < n4 ? 4u * g * cstride + 4u * (32 * q + i16) : 0x40000000u;
    const unsigned cvo1 = (32 * q + 16 + i16) < n4 ? 4u * g * cstride + 4u * (32 * q + 16 + i16) : 0x40000000u;
    auto row_src = [&](const int bt) {
        const int rows = min(p.B - bt * R, R);                // rows of this batch (<= 0 past the last batch: no records)
        const unsigned br = rows > 0 ? (unsigned)(rows - 1) * rstride4 + 16u * n4 : 0u;
        const unsigned bm = (rows > 0 && have_mask) ? (unsigned)(rows - 1) * mstride + 4u * n4 : 0u;
        RowSrc rs;
        if constexpr (XFUSE && GRAD) {
            const unsigned bc = (rows > 0 && p.codes_out) ? (unsigned)(rows - 1) * cstride + 4u * n4 : 0u;
            rs.c = __builtin_amdgcn_make_buffer_rsrc(p.codes_out + (size_t)bt * R * p.codes_stride, (short)0, (int)bc, 0x00020000);
        }
        rs.m = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(static_cast<const uint8_t*>(p.mask) + (size_t)bt * R * p.mask_stride + item0),
                                                 (short)0, (int)bm, 0x00020000);
        if constexpr (!CODES) rs.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.response + (size_t)bt * R * p.resp_stride + item0),
                                                                       (short)0, (int)br, 0x00020000);
        else rs.r = rs.m;
        if constexpr (!(XFUSE && GRAD)) rs.c = rs.m;
        return rs;
    };
    // quarter (h, j) = person 4 h + j of the lane (row 4 g + j + 16 h of the batch), both chunks: 2 x 16 B + 2 x 4 B
    auto load_quarter = [&](const int bt, const RowSrc& rs, const int h, const int j, float4 (&x)[CODES ? 1 : 8], uint32_t (&m)[8]) __attribute__((always_inline)) {
        bool linear = RM == 0;
        if constexpr (RM == 2) linear = p.row_index == nullptr;
        if (linear) {
            const int mso = (j + 16 * h) * (int)mstride;
            m[2 * j] = __builtin_amdgcn_raw_buffer_load_b32(rs.m, mvo0, mso, 0);
            m[2 * j + 1] = __builtin_amdgcn_raw_buffer_load_b32(rs.m, mvo1, mso, 0);
            if constexpr (!CODES) {
                const int so = (j + 16 * h) * (int)rstride4;
                x[2 * j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs.r, rvo0, so, 0));
                x[2 * j + 1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs.r, rvo1, so, 0));
            }
        } else {
            if (bt >= n_batches) return;
            const long long src = (long long)ridx[4 * h + j];
            if constexpr (CODES) {
                const uint32_t* mp = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride + item0);
                m[2 * j] = mp[cc0];
                m[2 * j + 1] = mp[cc1];
            } else {
                const float4* rp = reinterpret_cast<const float4*>(p.response + src * p.resp_stride + item0);
                x[2 * j] = rp[cc0];
                x[2 * j + 1] = rp[cc1];
                // (no branch around the mask loads: without a mask they read the response row again -- a valid address -- and a
                //  select drops the value; with the two-way form hipcc merged the stores of the two paths into one with a run-time
                //  offset, which put the whole register array into scratch memory)
                const bool hm = p.mask_dtype == 0;
                const uint32_t* mp = hm ? reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p.mask) + src * p.mask_stride + item0)
                                        : reinterpret_cast<const uint32_t*>(rp);
                const uint32_t w0 = mp[cc0], w1 = mp[cc1];
                m[2 * j] = hm ? w0 : 0u;
                m[2 * j + 1] = hm ? w1 : 0u;
            }
        }
    };
    // the single-panel statistics of the conditional / given posterior travel a batch ahead with eps (prs0..2) in every row mode
    // (round 3: cell codes only; on fp32 rows the slot lanes read them in the sync phase -- a memory round trip between the
    // barriers of every batch: VIBO_POSTERIOR_GIVEN 1.25 -> 1.21 ms at 8 dims, 1.06 -> 0.99 at 1; +5 spilled registers there)
    constexpr bool kPrs = true;
    float prs0 = 0.f, prs1 = 0.f, prs2 = 0.f;
    int prc = 0;                                      // whole-row answer counts of the panel mode (p.row_cnt), the same way
    // slots of this wave for a batch: [s0, s1) in steps of `step` (one slot at most with 4 or more waves)
    auto my_slots = [&](const int par, int& s0, int& s1, int& step) {
        if (NW8 || nw == 8) {
            s0 = q - 4 * par; s1 = s0 + 1; step = 1;
            if (s0 < 0 || s0 > 3) s1 = s0 = 0;
        } else {
            s0 = q; s1 = 4; step = nw;
        }
    };
    float epn = 0.f;                                  // eps of this wave's slot of the next batch (4 or more waves), loaded a batch ahead
    auto fetch_eps = [&](const int bt, const int par) __attribute__((always_inline)) {
        if ((!NW8 && nw < 4) || bt >= n_batches) return;
        int s0, s1, step;
        my_slots(par, s0, s1, step);
        if (s0 < s1) {
            const int row = bt * R + ((64 * s0 + lane) >> 3);
            // (the lane's dim goes through an empty asm: hipcc otherwise keeps 64-bit per-lane bases `pointer + 4 ed` of every
            //  array read here live across the whole batch -- in the instantiations at the register limit they were the spilled
            //  values, reloaded from scratch right here in every batch)
            int ed = lane & 7;
            asm volatile("" : "+v"(ed));
            epn = (ed < A && row < p.B) ? p.eps[(long long)row * A + ed] : 0.f;
            if constexpr (XGIVEN && kPrs) {            // caller-supplied posterior (given_pre_kernel's statements)
                const bool lv = ed < A && row < p.B;
                const float* po = p.given_post + (size_t)(lv ? row : 0) * 2 * A;
                const float lam_g = expf(-po[lv ? A + ed : 0]);
                prs0 = lv ? lam_g : 0.f; prs1 = lv ? po[ed] * lam_g : 0.f; prs2 = lv ? (float)p.I_total : 0.f;
            }
            if constexpr (XCOND && kPrs) {
                if (p.pre_stats && p.pre_panels == 1) {
                    const bool lv = ed < A && row < p.B;
                    const float* st = p.pre_stats + (size_t)(lv ? row : 0) * (2 * A + 1);
                    prs0 = lv ? st[ed] : 0.f; prs1 = lv ? st[A + ed] : 0.f; prs2 = lv ? st[2 * A] : 0.f;
                }
                if (p.row_cnt) {
                    const bool lv = ed < A && row < p.B;
                    prc = p.row_cnt[lv ? row : 0];
                    if (!lv) prc = 0;
                }
            }
        }
    };
    // ---- One-shot requests.
    // What the start of a launch costs (tools/ms_timing.py, prologue marks; 100 000 x 1 000): every workgroup needs its whole first
    // batch -- 160 KB, 41 MB over the chip -- before its first tile, the chip's workgroups ask for them within 0.3 us of each other,
    // and the memory system delivers that burst at ~4.9 TB/s: 8.4 us during which nothing computes, whatever the order of the
    // requests (rounds 5 and 6 tried: M-tile 1 with M-tile 0, the item sample first).  But the workgroups are not equally
    // loaded: n_batches % G of them stream one batch more than the others and finish last.  So the others -- `late`, a batch of
    // slack each -- hold their first row requests back until the long workgroups' burst is through (their item sample and
    // operand image are built in the meantime): the long ones start computing ~5 us earlier, the launch ends ~5 us earlier.
    const int rem_wg = n_batches % G;
    const bool late = rem_wg != 0 && wg >= rem_wg;                         // (wave-uniform)
    const unsigned long long t_wave_start = late ? realtime_ticks() : 0ull;
    int bt = wg;
    const RowSrc src_first = row_src(bt);
    auto first_rows = [&]() __attribute__((always_inline)) {      // the first batch's rows (both M-tiles) and noise; (two call sites)
        if (bt < n_batches) {
            fetch_idx(bt, ridx);
#pragma unroll
            for (int j = 0; j < 4; ++j) load_quarter(bt, src_first, 0, j, x, m);
#pragma unroll
            for (int j = 0; j < 4; ++j) load_quarter(bt, src_first, 1, j, x2, m2);
            fetch_eps(bt, 0);
            fetch_idx(bt + G, ridx_n);
        }
    };
    // (rows before the item sample in the long workgroups.  The CU's memory pipeline is one queue for its eight waves and takes
    //  ~6 k cycles to accept a batch's 256 load instructions: with the item sample requested first, the later waves' 4.6 KB sat
    //  behind the earlier waves' rows anyway -- it came in at 14-17 k cycles instead of 11-14 k, the first barrier at 17 k
    //  instead of 14 k: measured both ways, tools/ms_timing.py)
    if (!late) first_rows();
    __builtin_amdgcn_sched_barrier(0);
    // The wave's 128 items x D entries are one contiguous span of the caller's [I][D] item sample: read as that -- lane l takes
    // floats l, l + 64, ... (whole cache lines per instruction) -- and handed to the item's lane through the wave's own LDS below.
    // (Rounds 2-5: lane = item, one load per entry, lanes D floats apart: 18 x 18 line requests per wave for the same 4.6 KB, and
    //  all 2 048 waves of the launch asking the same 36 KB's L2 channels at once -- the item sample took 9-12 k cycles to arrive
    //  however early it was requested, tools/ms_timing.py.)
    constexpr int kItemReq = (kMsSpan * (VIBO_MAX_ABILITY_DIM + 2) + 63) / 64;      // loads per lane at D = 10
    float it_req[kItemReq];
    const int it_floats = kMsSpan * p.D;                                    // (wave-uniform)
    {
        const long long it_base = (long long)(item0 + kMsSpan * q) * p.D;
        const long long it_last = (long long)(item0 + I) * p.D - 1;         // last entry of this launch's (panel's) items
#pragma unroll
        for (int k = 0; k < kItemReq; ++k) {
            it_req[k] = 0.f;
            if (64 * k < it_floats) {                                       // (scalar branch; clamped index: always a valid address)
                const long long f = it_base + min(64 * k + lane, it_floats - 1);
                it_req[k] = p.item_raw[f < it_last ? f : it_last];
            }
        }
    }
    // XM == 3: the (mu, logvar) of this wave's items' two experts, [pass h: item 64 h + lane][code]
    float ct_mu[2][2], ct_lv[2][2];
    if constexpr (XFUSE) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int il = kMsSpan * q + 64 * h + lane;
            const int gi = item0 + (il < I ? il : 0);                       // (clamped: always a valid address)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float* te = p.cond_table + ((size_t)c * p.I_total + gi) * 2;
                ct_mu[h][c] = te[0];
                ct_lv[h][c] = te[1];
            }
        }
    }
    float tab_m = 0.f, tab_s = 0.f;                 // (threads 0..15: entry (c, a) of the 2-row expert table, mean | log variance)
    if (tid < 16) {
        const int c = tid >> 3, a = min(tid & 7, p.A - 1);
        tab_m = p.table[c * 2 * p.A + a];
        tab_s = p.table[c * 2 * p.A + p.A + a];
    }
    __builtin_amdgcn_sched_barrier(0);
    MS_P(16, t_entry)                                 // item sample (+ the long workgroups' first rows) requested

    // expert-table constants of the (person, dim) lanes (the two table entries were requested with the item sample)
    if (tid < 16) {
        const int c = tid >> 3, a = tid & 7;
        const float m = a < A ? tab_m : 0.f, s = a < A ? tab_s : 0.f;
        const float es = __expf(s);
        const float tau = 1.0f / (es + kPoeEps);
        cl.ctab[(0 * 2 + c) * 8 + a] = tau;
        cl.ctab[(1 * 2 + c) * 8 + a] = m * tau;
        cl.ctab[(2 * 2 + c) * 8 + a] = tau * tau * es;
        cl.ctab[(3 * 2 + c) * 8 + a] = m;
    }
    if (p.step_tick && blockIdx.x == 0 && tid == 0) *p.step_tick += 1;
    if constexpr (FLOWS) {
        if (tid < kMsMF * 8) {
            const int f = tid >> 3, a = tid & 7;
            const bool ok = f < p.n_flows && a < A;
            fl.fpar[f][0][a] = ok ? p.flow[(size_t)f * (2 * A + 1) + a] : 0.f;
            fl.fpar[f][1][a] = ok ? p.flow[(size_t)f * (2 * A + 1) + A + a] : 0.f;
        }
        if (tid < kMsMF) {
            float cwu = 0.f, b = 0.f;
            if (tid < p.n_flows) {
                const float* fp = p.flow + (size_t)tid * (2 * A + 1);
                for (int a = 0; a < A; ++a) cwu = fmaf(fp[A + a], fp[a], cwu);
                b = fp[2 * A];
            }
            fl.fsc[tid][0] = b;
            fl.fsc[tid][1] = cwu;
        }
        for (int k = tid; k < 2 * kMsRows; k += (int)blockDim.x) (&fl.lacc[0][0])[k] = 0.f;
        for (int k = tid; k < 2 * 4 * 2 * kMsMF * 3 * 8; k += (int)blockDim.x) (&fl.facc[0][0][0][0][0][0])[k] = 0.f;
    }

    // ---- item operands of this wave's 128 items: read from the caller's [I][D] item sample and brought to the kernel's
    //      form here (log2 units: na_a = -a_ia log2 e, or +log2 e for 1PL; nb = b_i log2 e; guess = sigmoid -- what
    //      item_prep_kernel does for the other kernels, without its launch), then to a wave-private LDS image as f16
    //      hi/lo pieces; the MFMA operands are read from there when needed:
    //   logit MFMA, tile (u, t): lane (item 4 (32 q + 16 u + i16) + t, g) reads 16 B: g 0/2 = na hi, 1 = na lo, 3 = bias pieces
    //   d LL/d theta MFMA, (u, kt): lane (col i16, g) gets k = 8 g + kk <-> item (chunk 4 g + (kk & 3), t = 2 kt + (kk >> 2)) by two
    //   transposed reads of the [na_hi | na_lo] rows
    //
    // Range of the f16 pieces.  A two-piece f16 value carries an ABSOLUTE error of ~2^-24 once its lo piece is subnormal
    // (|x| < 2^-3), and saturates above 65 504.  Two per-launch powers of two keep the operands where the split is good
    // for any item parameters -- at no cost in the hot path (the cells' codes stay +-1, the logits come out unscaled):
    //   * theta <-> a balance 2^jsh: the image holds a 2^jsh, the (person, dim) lanes hand over theta 2^-jsh; jsh =
    //     -floor(exponent(max |a|) / 2), i.e. both sides near sqrt(|a theta|) (discriminations of 1e-4 or 1e6 alike).
    //     d LL/d theta = sum g a comes out times 2^jsh, d LL/d a = sum g theta times 2^-jsh: undone for free in the
    //     backward's existing scale factor and once in the epilogue;
    //   * difficulties: the three bias pieces hold b 2^-bsh and the "1" entries of the logit MFMA's A operand 2^bsh
    //     (bsh = 0 below 2^15).
    // Beyond that (|b| > 2^30, or a sample with |theta 2^-jsh| > 65 504) the results turn into NaN rather than into
    // silently wrong numbers; VIBO_FLAG_KERNEL_VALU runs such inputs on the fp32 VALU kernel.
    // 3PL: guess probabilities of the wave's items, read back per tile (see kMsGuessFloats)
    float* const gsl = reinterpret_cast<float*>(ms_smem + sizeof(MsCommonLds) + (size_t)nw * sizeof(MsWaveLds) + (FLOWS ? sizeof(MsFlowLds) : 0)) + q * kMsGuessFloats;
    // XM == 3: this wave's block behind the guess block: experts [u][t][i16] x float4, then its sums [lam | s][row]
    float* const xls0 = reinterpret_cast<float*>(ms_smem + sizeof(MsCommonLds) + (size_t)nw * sizeof(MsWaveLds) + (FLOWS ? sizeof(MsFlowLds) : 0)) +
                        (IRT == 3 ? nw * kMsGuessFloats : 0);
    float* const xls = xls0 + q * kMsFuseFloats;
    if constexpr (XFUSE) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = kMsSpan * q + 64 * h + lane < I;
            float4 e = float4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float t0 = 1.0f / (expf(ct_lv[h][0]) + kPoeEps), t1 = 1.0f / (expf(ct_lv[h][1]) + kPoeEps);      // utils.py:105-113
                e = float4{t0, ct_mu[h][0] * t0, t1, ct_mu[h][1] * t1};
            }
            reinterpret_cast<float4*>(xls)[(h * 4 + (lane & 3)) * 16 + (lane >> 2)] = e;      // item 64 h + 4 i16 + t -> [h][t][i16]
        }
    }
    MS_P(17, t_entry)                                 // encoder table in LDS (its loads waited for)
    // item sample: through the wave's own LDS (the g-piece / operand-image area, not in use yet) to the item's lane
    float* const it_stage = reinterpret_cast<float*>(&wl.tr[0][0][0]);
    static_assert(sizeof(wl.tr) + sizeof(wl.img) >= (size_t)kMsSpan * (VIBO_MAX_ABILITY_DIM + 2) * sizeof(float), "item staging area");
#pragma unroll
    for (int k = 0; k < kItemReq; ++k)
        if (64 * k < it_floats) it_stage[64 * k + lane] = it_req[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave's own LDS writes, in order: visible to its reads)
    float a_req[2][8], b_req[2];
    float g_raw[2][4];                                      // 3PL: guess logits of the lane's items of tile (u, t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float* src = it_stage + (64 * h + lane) * p.D;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) a_req[h][kk] = IRT != 1 ? src[min(kk, A - 1)] : 0.f;
        b_req[h] = src[IRT == 1 ? 0 : A];
    }
    if constexpr (IRT == 3) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) g_raw[u][t] = it_stage[(64 * u + 4 * i16 + t) * p.D + A + 1];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (all read before the operand image overwrites the area)
    float na_raw[2][8], nb_raw[2];
    float amax = 0.f, bmax = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int xl = 64 * h + lane;                                   // item of the wave, one per lane and pass
        const int il = kMsSpan * q + xl;
        const bool ok = il < I;
        const float (&a_raw)[8] = a_req[h];
        const float b_raw = b_req[h];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float na = 0.f;
            if constexpr (IRT == 1) na = (ok && kk < A) ? kLog2e : 0.f;                     // models.py:731
            else na = (ok && kk < A) ? -a_raw[kk] * kLog2e : 0.f;                          // models.py:744,759
            na_raw[h][kk] = na;
            amax = !(fabsf(na) <= 3.0e38f) ? 3.0e38f : fmaxf(amax, fabsf(na));       // (NaN / Inf: "too large", sticky)
        }
        nb_raw[h] = ok ? b_raw * kLog2e : 0.f;
        bmax = !(fabsf(nb_raw[h]) <= 3.0e38f) ? 3.0e38f : fmaxf(bmax, fabsf(nb_raw[h]));
    }
    {
        // workgroup maxima
        float ma = amax, mb = bmax;
        ma = fmaxf(ma, dpp_f<0xb1>(ma)); ma = fmaxf(ma, dpp_f<0x4e>(ma));
        ma = fmaxf(ma, dpp_f<0x124>(ma)); ma = fmaxf(ma, dpp_f<0x128>(ma));
        ma = fmaxf(ma, __shfl_xor(ma, 16)); ma = fmaxf(ma, __shfl_xor(ma, 32));
        mb = fmaxf(mb, dpp_f<0xb1>(mb)); mb = fmaxf(mb, dpp_f<0x4e>(mb));
        mb = fmaxf(mb, dpp_f<0x124>(mb)); mb = fmaxf(mb, dpp_f<0x128>(mb));
        mb = fmaxf(mb, __shfl_xor(mb, 16)); mb = fmaxf(mb, __shfl_xor(mb, 32));
        if (lane == 0) { wl.red[1] = ma; wl.red[2] = mb; }
    }
    MS_P(18, t_entry)                                 // item sample in, wave maxima
    __syncthreads();
    MS_P(19, t_entry)
    float wg_amax = 0.f, wg_bmax = 0.f;
    for (int w = 0; w < nw; ++w) { wg_amax = fmaxf(wg_amax, wls[w].red[1]); wg_bmax = fmaxf(wg_bmax, wls[w].red[2]); }
    // frexp exponents e: max < 2^e
    const int e_a = __builtin_amdgcn_readfirstlane((int)((__builtin_bit_cast(uint32_t, wg_amax) >> 23) & 0xff) - 126);
    const int e_b = __builtin_amdgcn_readfirstlane((int)((__builtin_bit_cast(uint32_t, wg_bmax) >> 23) & 0xff) - 126);
    int jsh = -(e_a >> 1);
    jsh = jsh < -14 ? -14 : jsh > 12 ? 12 : jsh;
    int bsh = e_b > 15 ? e_b - 15 : 0;
    const bool range_fault = bsh > 15 || e_a + jsh > 15;      // (wave-uniform) not representable: poison the outputs
    if (bsh > 15) bsh = 15;
    const float sc_a = __builtin_bit_cast(float, (uint32_t)(127 + jsh) << 23);         // 2^jsh: image = a 2^jsh
    const float sc_t = __builtin_bit_cast(float, (uint32_t)(127 - jsh) << 23);         // 2^-jsh: operands = theta 2^-jsh
    const float sc_b = __builtin_bit_cast(float, (uint32_t)(127 - bsh) << 23);         // 2^-bsh: bias pieces
    const _Float16 one_b = (_Float16)__builtin_bit_cast(float, (uint32_t)(127 + bsh) << 23);   // 2^bsh: their A-operand entries
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int xl = 64 * h + lane;
        _Float16* dst = &wl.img[h][0] + ((xl & 63) >> 2) * kMsItemLane + (xl & 3) * kMsItemRow;
        half8 hi8, lo8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            _Float16 hi, lo;
            split16(na_raw[h][kk] * sc_a, hi, lo);
            hi8[kk] = hi; lo8[kk] = lo;
        }
        const float nb = range_fault ? __builtin_nanf("") : nb_raw[h] * sc_b;
        _Float16 b0, b1, b2, b3;
        split16(nb, b0, b1);
        split16(nb - (float)b0 - (float)b1, b2, b3);
        *reinterpret_cast<half8*>(dst) = hi8;
        *reinterpret_cast<half8*>(dst + 8) = lo8;
        *reinterpret_cast<half8*>(dst + 16) = half8{b0, b1, b2, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    }
    if constexpr (IRT == 3) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int il = kMsSpan * q + 64 * u + 4 * i16 + t;
                const float gv = il < I ? 1.0f / (1.0f + expf(-g_raw[u][t])) : 0.f;           // models.py:758
                if (g == 0) gsl[(u * 4 + t) * 16 + i16] = gv;
            }
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
    for (int k = tid; k < 2 * 12 * 256; k += (int)blockDim.x) (&cl.tacc[0][0][0])[k] = 0.f;
    float s_log = 0.f;
    int unobs = 0;
    bool sat3 = false;                  // 3PL: this wave met a probability past the clamp in this batch (wave-uniform: see the tile)
    MS_P(20, t_entry)                                 // operand image written
    __syncthreads();
    MS_P(21, t_entry)

    // LDS image offsets (halfs): producer row 16 t + i16, piece g; consumer rows 32 kt + 4 g + (i16 >> 2) (+ 16), piece i16 & 3
    const int wofs = 16 * i16 + 4 * (g ^ (i16 >> 2));
    const int rofs = 64 * g + 16 * (i16 >> 2) + 4 * ((i16 & 3) ^ g);

    // pack a quarter: the code words of person 4 h + j for both u-steps and its counts
    const uint32_t tm_tail = (I & 3) ? ((1u << (8 * (I & 3))) - 1u) : 0xFFFFFFFFu;
    const uint32_t tm0 = (32 * q + i16) >= n4 ? 0u : ((I & 3) && (32 * q + i16) == (I >> 2)) ? tm_tail : 0xFFFFFFFFu;
    const uint32_t tm1 = (32 * q + 16 + i16) >= n4 ? 0u : ((I & 3) && (32 * q + 16 + i16) == (I >> 2)) ? tm_tail : 0xFFFFFFFFu;
    // Rows past the matrix' end (last batch only): in-order rows with a mask read zeros there (the resource's record
    // limit) and need nothing; gathered rows, rows without a mask and cell codes (0 = an answer) are masked per lane -- in
    // a variant of the pack of its own, so that the common case carries no selects.
    bool need_in = RM != 0 || !have_mask;
    if constexpr (RM == 2) need_in = true;
    // pk[j]: 8-bit fields nobs | nobs of M-tile 1 | n1 | n1 of M-tile 1 of persons j and 4 + j (each <= 8 per lane)
    float2v xacc[XFUSE ? 8 : 1];                     // XM == 3: (lam, s) shares of the lane's 8 persons over its 8 items
#pragma unroll
    for (int k = 0; k < (XFUSE ? 8 : 1); ++k) xacc[k] = float2v{0.f, 0.f};
    // (XM == 3: the experts of the lane's 8 items, [4 u + t]; read per quarter in front of the tiles -- nothing can stay live across a
    //  tile there -- and once for the four quarters packed back to back behind the last tile)
    struct Experts { float4 e[XFUSE ? 8 : 1]; };
    auto read_experts = [&](Experts& ex) __attribute__((always_inline)) {
        if constexpr (XFUSE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) ex.e[k] = reinterpret_cast<const float4*>(xls)[k * 16 + i16];
        }
    };
    auto pack_quarter = [&](auto inc, const int left, const int h, const int j, uint32_t (&cw0)[8], uint32_t (&cw1)[8], int (&pk)[4],
                            const float4 (&x)[CODES ? 1 : 8], const uint32_t (&m)[8], const RowSrc& rs, const Experts& ex) {
        constexpr bool IN = decltype(inc)::value;
        uint32_t k0 = tm0, k1 = tm1;
        if constexpr (IN) {
            const bool in = 4 * g + j + 16 * h < left;
            k0 = in ? tm0 : 0u; k1 = in ? tm1 : 0u;
        }
        // 1PL/2PL carry -w in the codes (the exponent is -w x logit: one VOP2 multiply)
        int nobs = 0, n1 = 0;
        if constexpr (CODES) {
            cw0[4 * h + j] = pack_cell_codes4_lut(m[2 * j], k0, kLutCode, nobs, n1);
            cw1[4 * h + j] = pack_cell_codes4_lut(m[2 * j + 1], k1, kLutCode, nobs, n1);
        } else if constexpr (XFUSE) {
            // the same words, plus what cond_pre_kernel does with the cells: [right] / [wrong] indicators (exactly 0 / 1) times the
            // item's two experts, summed per person; and the cells' Format P codes for the table-gradient pass
            float2v acc = xacc[4 * h + j];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t me = ((u ? m[2 * j + 1] : m[2 * j]) | fillw) & (u ? k1 : k0);      // observed cells (bytes 0 / 1)
                uint32_t xm;                                                                      // ... answered right
                const uint32_t cw = pack_codes4_lut_x(x[2 * j + u], me, kLutFp32, nobs, n1, xm);
                if (u) cw1[4 * h + j] = cw; else cw0[4 * h + j] = cw;
                const uint32_t p0 = me ^ xm;
                float wp[4], wn[4];
                asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(wp[0]) : "v"(xm));
                asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(wp[1]) : "v"(xm));
                asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(wp[2]) : "v"(xm));
                asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(wp[3]) : "v"(xm));
                asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(wn[0]) : "v"(p0));
                asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(wn[1]) : "v"(p0));
                asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(wn[2]) : "v"(p0));
                asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(wn[3]) : "v"(p0));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 e = ex.e[u * 4 + t];
                    acc = float2v{wn[t], wn[t]} * float2v{e.x, e.y} + acc;
                    acc = float2v{wp[t], wp[t]} * float2v{e.z, e.w} + acc;
                }
                if constexpr (GRAD) {
                    const uint32_t code = xm | ((me ^ 0x01010101u) << 1);                         // 0 wrong | 1 right | 2 missing
                    // (the stores themselves: 1.76 -> 1.72 ms per 1M x 1k call without them -- 1 GB leaves at a twentieth of the cost of
                    //  the pass they replace)
                    __builtin_amdgcn_raw_buffer_store_b32(code, rs.c, u ? cvo1 : cvo0, (j + 16 * h) * (int)cstride, 0);
                }
            }
            xacc[4 * h + j] = acc;
        } else {
            cw0[4 * h + j] = pack_codes4_lut(x[2 * j], (m[2 * j] | fillw) & k0, kLutFp32, nobs, n1);
            cw1[4 * h + j] = pack_codes4_lut(x[2 * j + 1], (m[2 * j + 1] | fillw) & k1, kLutFp32, nobs, n1);
        }
        pk[j] += (nobs | (n1 << 16)) << (8 * h);
        // (pinned here: hipcc otherwise sinks the whole pack to the end of the batch, and the next loads take new registers)
        asm volatile("" : "+v"(cw0[4 * h + j]), "+v"(cw1[4 * h + j]), "+v"(pk[j]));
        if constexpr (XFUSE) asm volatile("" : "+v"(xacc[4 * h + j]));
    };
    auto pack_one = [&](const int bt, const int h, const int j, uint32_t (&cw0)[8], uint32_t (&cw1)[8], int (&pk)[4],
                        const float4 (&x)[CODES ? 1 : 8], const uint32_t (&m)[8], const RowSrc& rs) __attribute__((always_inline)) {
        const int left = p.B - bt * R;               // (wave-uniform) only the last batch has rows past the end
        Experts ex;
        read_experts(ex);
        if (need_in && left < R) pack_quarter(std::true_type{}, left, h, j, cw0, cw1, pk, x, m, rs, ex);
        else pack_quarter(std::false_type{}, left, h, j, cw0, cw1, pk, x, m, rs, ex);
    };
    auto pack_half = [&](const int bt, const int h, uint32_t (&cw0)[8], uint32_t (&cw1)[8], int (&pk)[4],
                         const float4 (&x)[CODES ? 1 : 8], const uint32_t (&m)[8], const RowSrc& rs) {
        const int left = p.B - bt * R;               // (wave-uniform) only the last batch has rows past the end
        if constexpr (XFUSE && RM != 0) {
            // (gathered rows carry 16 row numbers: the experts held across four quarters spilled 20-41 registers there)
#pragma unroll
            for (int j = 0; j < 4; ++j) pack_one(bt, h, j, cw0, cw1, pk, x, m, rs);
            return;
        }
        Experts ex;
        read_experts(ex);
        if (need_in && left < R) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pack_quarter(std::true_type{}, left, h, j, cw0, cw1, pk, x, m, rs, ex);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) pack_quarter(std::false_type{}, left, h, j, cw0, cw1, pk, x, m, rs, ex);
        }
    };
    // XM == 3: the lane's 16 sums (8 persons x lam | s) -> the sums over the 16 lanes of its row group, by halving: in each of the four
    // steps a lane keeps the half of its values its own bit selects and adds the partner's copies of those (fixed order: bitwise
    // reproducible) -- lane i16 ends with value i16 = 8 kind + 4 h + j and stores it
    auto put_stats = [&]() {
        if constexpr (XFUSE) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = xacc[k][0]; v[8 + k] = xacc[k][1]; }
            const bool b3 = (i16 & 8) != 0, b2 = (i16 & 4) != 0, b1 = (i16 & 2) != 0, b0 = (i16 & 1) != 0;
            float w8[8], w4[4], w2[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float give = b3 ? v[k] : v[k + 8]; w8[k] = (b3 ? v[k + 8] : v[k]) + dpp_f<0x128>(give); }   // row_ror 8
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float give = b2 ? w8[k] : w8[k + 4]; w4[k] = (b2 ? w8[k + 4] : w8[k]) + dpp_f<0x141>(give); }  // row_half_mirror
#pragma unroll
            for (int k = 0; k < 2; ++k) { const float give = b1 ? w4[k] : w4[k + 2]; w2[k] = (b1 ? w4[k + 2] : w4[k]) + dpp_f<0x4e>(give); }   // quad_perm [2,3,0,1]
            const float give = b0 ? w2[0] : w2[1];
            const float w1 = (b0 ? w2[1] : w2[0]) + dpp_f<0xb1>(give);                                                                     // quad_perm [1,0,3,2]
            xls[2 * 4 * 16 * 4 + (i16 >> 3) * kMsRows + 4 * g + (i16 & 3) + 16 * ((i16 >> 2) & 1)] = w1;
#pragma unroll
            for (int k = 0; k < 8; ++k) xacc[k] = float2v{0.f, 0.f};
        }
    };
    // packed counts of the lane's 8 persons (both u-steps) -> 16-lane sums -> wl.cnt
    auto put_counts = [&](const int (&pk)[4], const bool real) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int t = pk[k];
            t += dpp_i<0xb1>(t);                    // quad_perm [1,0,3,2]
            t += dpp_i<0x4e>(t);                    // quad_perm [2,3,0,1]
            t += dpp_i<0x141>(t);                   // row_half_mirror
            t += dpp_i<0x140>(t);                   // row_mirror
            v[k] = t;
        }
        {
            int obs = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) obs += (pk[k] & 0xff) + ((pk[k] >> 8) & 0xff);
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
    // EXT: the variant that reads global memory in the sync phase (whole-row counts of the panel mode, the conditional /
    // given posterior's statistics).  The plain variant issues no load at all: a load destination shared with it would make
    // hipcc guard the register with an s_waitcnt vmcnt(0) -- behind the row loads that are in flight across the sync phase.
    // (on cell codes, RM == 2, the single-panel statistics of the conditional / given posterior come a batch ahead with
    //  eps -- prs -- and the plain variant serves them: the conditional posterior's matrix pass runs on emitted codes)
    // prior experts of the missing cells (models.py:613-620): weight 1 / (1 + eps) each, or dropped
    const float prior_w = p.missing_mode == 0 ? 1.0f / (1.0f + kPoeEps) : 0.f;
    auto forward_slot = [&](auto extc, const int bt, const int par, const int s, const float eps_c) {
        constexpr bool EXT = decltype(extc)::value;
        static_assert(EXTRA || !EXT, "the global-memory variant belongs to the EXTRA instantiations");
        const int row0 = bt * R;
        const int e = 64 * s + lane, pp = e >> 3;
        const bool live = ed < A && (row0 + pp) < p.B;
        int cnt = 0;
        bool have_cnt = false;
        if constexpr (EXT && XCOND) {
            if (p.row_cnt) {
                cnt = live ? p.row_cnt[row0 + pp] : 0;
                have_cnt = true;
            }
        }
        if constexpr (EXTRA && !EXT && XCOND) {
            // (panel mode: the whole-row counts came a batch ahead with eps.  Read here, between the barriers, the load's
            //  s_waitcnt vmcnt(0) also waited for the first half of the next batch's rows, requested just before barrier A:
            //  forward phase 1 690 -> 690 cycles per batch, wide plain rows 1.83 -> 1.73 ms)
            if (p.row_cnt) { cnt = prc; have_cnt = true; }
        }
        if (!have_cnt) {
            if constexpr (NW8) {
                int ppo = pp;
                if constexpr (kPinFlowAddr) asm volatile("" : "+v"(ppo));     // (see backward_slot)
#pragma unroll
                for (int w = 0; w < 8; ++w) cnt += wls[w].cnt[ppo];
            } else {
#pragma unroll 1
                for (int w = 0; w < nw; ++w) cnt += wls[w].cnt[pp];
            }
        }
        const float n1 = (float)(cnt >> 16);
        float nobs = (float)(cnt & 0xffff);
        const float n0 = nobs - n1;
        const float tau0 = cl.ctab[(0 * 2 + 0) * 8 + ed], tau1 = cl.ctab[(0 * 2 + 1) * 8 + ed];
        const float mt0 = cl.ctab[(1 * 2 + 0) * 8 + ed], mt1 = cl.ctab[(1 * 2 + 1) * 8 + ed];
        // (written as one fma each: the contraction hipcc picks for a sum of two products depends on the surrounding code,
        //  and the variants of this kernel have to agree bit for bit)
        float lam = fmaf(n0, tau0, n1 * tau1), smu = fmaf(n0, mt0, n1 * mt1);
        if constexpr (EXT && XCOND) {
            if (p.pre_stats) {
                lam = 0.f; smu = 0.f; nobs = 0.f;
                if (live) {
                    for (int pn = 0; pn < p.pre_panels; ++pn) {
                        const float* st = p.pre_stats + ((size_t)pn * p.B + (row0 + pp)) * (2 * A + 1);
                        lam += st[ed]; smu += st[A + ed]; nobs += st[2 * A];
                    }
                }
            }
        }
        if constexpr (EXT && XGIVEN) {
            lam = 0.f; smu = 0.f; nobs = (float)p.I_total;
            if (live) {
                const float* po = p.given_post + (size_t)(row0 + pp) * 2 * A;
                lam = expf(-po[A + ed]);
                smu = po[ed] * lam;
            }
        }
        if constexpr (EXTRA && !EXT && kPrs) {
            if (XGIVEN || p.pre_stats) { lam = prs0; smu = prs1; nobs = prs2; }
        }
        if constexpr (XFUSE) {               // the waves' shares of the experts' sums (put_stats), fixed order
            lam = 0.f; smu = 0.f;
            if constexpr (NW8) {
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const float* st = xls0 + w * kMsFuseFloats + 2 * 4 * 16 * 4;
                    lam += st[pp];
                    smu += st[kMsRows + pp];
                }
            } else {
#pragma unroll 1
                for (int w = 0; w < nw; ++w) {
                    const float* st = xls0 + w * kMsFuseFloats + 2 * 4 * 16 * 4;
                    lam += st[pp];
                    smu += st[kMsRows + pp];
                }
            }
        }
        const float nmiss = (float)p.I_total - nobs;
        lam = fmaf(nmiss, prior_w, lam);
        if (!live) lam = 1.0f;
        const float inv_lam = 1.0f / lam;
        const float amu = smu * inv_lam;
        const float sig = fast_rsq(lam);
        const float th0 = live ? amu + sig * eps_c : 0.f;
        float thv = th0;
        float ladj = 0.f;
        if constexpr (FLOWS) {              // z <- z + uhat tanh(w.z + b)
            int ppf = pp, edf = ed;             // (opaque copies: see backward_slot)
            if constexpr (kPinFlowAddr) asm volatile("" : "+v"(ppf), "+v"(edf));
#pragma unroll
            for (int f = 0; f < kMsMF; ++f) {
                if (f < p.n_flows) {
                    const float ud = fl.fpar[f][0][edf], wd = fl.fpar[f][1][edf];
                    const float aa = ms_group_sum(thv * wd) + fl.fsc[f][0];
                    // tanh(x) = 1 - 2 / (1 + e^2x), |x| clamped so that e^2x stays finite (tanh(+-15) = +-1 in fp32)
                    const float t = 1.0f - 2.0f * fast_rcp(1.0f + fast_exp2((2.0f * kLog2e) * med3(aa, -15.f, 15.f)));
                    const float psi = 1.0f + (1.0f - t * t) * fl.fsc[f][1];
                    if (GRAD && ed == 0) {
                        fl.tps[par][f][0][ppf] = t;
                        fl.tps[par][f][1][ppf] = psi;
                    }
                    ladj += kLn2 * fast_log2(fabsf(psi) + 1e-8f);
                    thv = fmaf(ud, t, thv);
                }
            }
        }
        const float ths = thv * sc_t;                   // operands carry theta 2^-jsh (see the operand image)
        if (live && primary) {
            const long long o = (long long)(row0 + pp) * A + ed;
            const float alv = -kLn2 * fast_log2(lam);
            // (non-temporal: 96 MB of per-person outputs per 1M x 8 launch that nothing on the device reads back soon -- kept out
            //  of the L2 the row loads share their segment-edge lines through: -0.9 % on the headline call)
            __builtin_nontemporal_store(amu, p.ability_mu + o);
            __builtin_nontemporal_store(alv, p.ability_logvar + o);
            __builtin_nontemporal_store(th0, p.ability + o);
            if constexpr (FLOWS) {
                p.ability_k[o] = thv;
                if (ed == 0) {
                    p.ability_ladj[row0 + pp] = ladj;
                    lds_add(&fl.lacc[par][pp], ladj);
                }
            }
            // a sample beyond the f16 range (or NaN): the regulariser sum -- and with it the loss -- becomes NaN.  (A NaN logit
            // alone would not do: the clamp's v_med3 returns a finite bound for it.)
            const float kl_term = -0.5f * (1.0f + alv - amu * amu - inv_lam);
            lds_add(&cl.tacc[par][8][e], fabsf(ths) <= 65504.f ? kl_term : __builtin_nanf(""));
            lds_add(&cl.tacc[par][9][e], -0.5f * kLog2Pi - 0.5f * alv - 0.5f * eps_c * eps_c);
            lds_add(&cl.tacc[par][10][e], -0.5f * kLog2Pi - 0.5f * thv * thv);
            if (ed == 0) lds_add(&cl.tacc[par][11][e], nobs);
        }
        if constexpr (GRAD) {
            cl.st[par][0][e] = amu; cl.st[par][1][e] = sig; cl.st[par][2][e] = inv_lam; cl.st[par][3][e] = eps_c;
            cl.st[par][4][e] = __builtin_bit_cast(float, cnt);
        }
        _Float16 hi, lo;
        split16(ths, hi, lo);
        const int slot = 8 * ((pp & 15) >> 2) + 4 * (pp >> 4) + (pp & 3);
        cl.thA[0][pp][ed] = hi;
        cl.thA[1][pp][ed] = lo;
        cl.thT[ed][slot] = hi;
        cl.thT[8 + ed][slot] = lo;
    };
    // backward of one slot through the sample and the product of experts; the 8 table-gradient sums of the wave's lanes
    // go to the wave's LDS record (fixed order: bitwise reproducible)
    auto backward_slot = [&](const int bt, const int par, const int s) {
        const int row0 = bt * R;
        const int e = 64 * s + lane, pp = e >> 3;
        const bool live = ed < A && (row0 + pp) < p.B;
        float g0 = 0.f;
        if constexpr (NW8) {
            // (flows: the element index behind an opaque copy -- the eight record addresses are then formed here from one base
            //  and immediate offsets; left to itself the compiler keeps eight loop-invariant addresses in registers for the
            //  whole batch loop, which the flow instantiations, already at the 256-register budget, pay for in spills)
            int go = pp * 8 + ed;
            if constexpr (kPinFlowAddr) asm volatile("" : "+v"(go));
            const float* const g8 = &wls[0].gth[par][0][0] + go;
#pragma unroll
            for (int w = 0; w < 8; ++w) g0 += g8[w * (int)(sizeof(MsWaveLds) / sizeof(float))];
        } else {
#pragma unroll 1
            for (int w = 0; w < nw; ++w) g0 += wls[w].gth[par][pp][ed];
        }
        float gz0 = live ? g0 * (kLn2 * sc_t) : 0.f;          // (log2 units -> nats; sum g a' = 2^jsh sum g a)
        const float amu = cl.st[par][0][e], sig = cl.st[par][1][e], inv_lam = cl.st[par][2][e], eps_c = cl.st[par][3][e];
        const int cnt = __builtin_bit_cast(int, cl.st[par][4][e]);
        const float n1 = (float)(cnt >> 16), n0 = (float)(cnt & 0xffff) - n1;
        float thv = live ? amu + sig * eps_c : 0.f;
        const bool reg_on = live && primary;
        float gz1 = 0.f;
        if constexpr (!FLOWS) {
            gz1 = (reg_on && p.reg_mode != 0) ? thv : 0.f;        // d REG / d theta_K (-log p)
        } else {
            // the sample after the flows again (the forward's arithmetic on the kept tanh values: same bits).  (ppf, edf: the row
            // and dim indices behind opaque copies, so that the record and parameter addresses are formed here -- one base,
            // immediate offsets -- instead of being hoisted out of the batch loop, one register per address: 16 and more
            // loop-invariant registers that the 3PL instantiations, at the 256-register budget, paid for in spills)
            int ppf = pp, edf = ed;
            if constexpr (kPinFlowAddr) asm volatile("" : "+v"(ppf), "+v"(edf));
#pragma unroll
            for (int f = 0; f < kMsMF; ++f)
                if (f < p.n_flows) thv = fmaf(fl.fpar[f][0][edf], fl.tps[par][f][0][ppf], thv);
            gz1 = (reg_on && p.reg_mode != 0) ? thv : 0.f;
            float znext = thv;                  // output of the flow being backpropagated
            const float lv = live ? 1.0f : 0.f, lv1 = reg_on ? 1.0f : 0.f;
            const int sl = (e >> 6) & 3;
            auto flow_back = [&](const int f) {
                const float ud = fl.fpar[f][0][edf], wd = fl.fpar[f][1][edf];
                const float cwu = fl.fsc[f][1];
                const float t = fl.tps[par][f][0][ppf], psi = fl.tps[par][f][1][ppf];
                const float zin = fmaf(-ud, t, znext);       // its input (to an ulp of the forward's value)
                znext = zin;
                const float omt = 1.0f - t * t;
                // set 0 (LL): no log-det term
                const float g_a0 = ms_group_sum(gz0 * ud) * omt;
                const float f00 = gz0 * t, f01 = lv * g_a0 * zin, f02 = lv * g_a0;
                gz0 = fmaf(g_a0, wd, gz0) * lv;
                // set 1 (REG = log q0 - sum log|det| - log p): d REG / d ladj = -1.  Only the primary panel owns the regulariser
                // (wave-uniform): the other nine panels of a 10 000-item row skip its arithmetic and its three folds per flow.
                float f10 = 0.f, f11 = 0.f, f12 = 0.f;
                if (primary) {
                    const float unit = ((psi >= 0.f) ? 1.0f : -1.0f) / (fabsf(psi) + 1e-8f);
                    const float dl_dpsi = -unit;
                    const float g_t = dl_dpsi * (-2.0f * t * cwu) + ms_group_sum(gz1 * ud);
                    const float g_c = dl_dpsi * omt;
                    const float g_a1 = g_t * omt;
                    f10 = lv1 * (gz1 * t + g_c * wd); f11 = lv1 * (g_a1 * zin + g_c * ud); f12 = lv1 * g_a1;
                    gz1 = fmaf(g_a1, wd, gz1) * lv1;
                }
                // the slot's 8 persons (lanes 8 apart) -> lanes 0..7, then one running sum per (slot, set, kind, dim)
                auto fold = [&](float v, float* dst) {
                    v += dpp_f<0x128>(v);               // row_ror 8
                    v = xor16_add(v);
                    v = xor32_add(v);
                    if (lane < 8) lds_add(dst, v);
                };
                fold(f00, &fl.facc[par][sl][0][f][0][ed]); fold(f01, &fl.facc[par][sl][0][f][1][ed]); fold(f02, &fl.facc[par][sl][0][f][2][ed]);
                if (primary) {
                    fold(f10, &fl.facc[par][sl][1][f][0][ed]); fold(f11, &fl.facc[par][sl][1][f][1][ed]); fold(f12, &fl.facc[par][sl][1][f][2][ed]);
                }
            };
#pragma unroll 1
            for (int f = p.n_flows - 1; f >= 0; --f) flow_back(f);
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
        if (XGIVEN && p.given_grad) {
            if (live) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    float* gg = p.given_grad + ((size_t)st * p.B + (row0 + pp)) * 2 * A;
                    gg[ed] = gmu[st];
                    gg[A + ed] = glv[st];
                }
            }
        }
        if (XCOEF && post_coef) {
            if (live) {
                float* pc = post_coef + (size_t)(row0 + pp) * 4 * A;
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
        for (int k = 0; k < 8; ++k) lds_add(&cl.tacc[par][k][e], dt[k]);      // (one lane per address: order is program order)
    };
    auto person_forward = [&](const int bt, const int par) {
        int s0, s1, step;
        my_slots(par, s0, s1, step);
        if constexpr (EXTRA || !NW8) {
            bool ext = nw < 4;                                   // (wave-uniform)
            if constexpr (XCOND) ext = ext || (p.pre_stats && !(kPrs && p.pre_panels == 1));      // (given_post, row_cnt: prs / prc)
            if (ext) {
                // (without the EXTRA hooks the plain variant serves the narrow workgroups too: eps is loaded here)
#pragma unroll 1
                for (int s = s0; s < s1; s += step) {
                    float eps_c = epn;
                    if (nw < 4) {
                        const int row = bt * R + ((64 * s + lane) >> 3);
                        eps_c = (ed < A && row < p.B) ? p.eps[(long long)row * A + ed] : 0.f;
                    }
                    forward_slot(std::integral_constant<bool, EXTRA>{}, bt, par, s, eps_c);
                }
                return;
            }
        }
        if (s0 < s1) forward_slot(std::false_type{}, bt, par, s0, epn);
    };
    auto person_backward = [&](const int bt, const int par) {
        int s0, s1, step;
        my_slots(par, s0, s1, step);
#pragma unroll 1
        for (int s = s0; s < s1; s += step) backward_slot(bt, par, s);
    };

    // ---- the batch's math: 8 item tiles (2 u-steps x 4) x 2 M-tiles, software-pipelined by hand:
    //      tile n issues the logit MFMAs of tile n + 1 first (their latency and the LDS read of the item operand hide
    //      under tile n's element-wise stream), and the d LL/d theta MFMAs of a 32-item K-tile run one tile after its
    //      pieces were written to the LDS image (the transposed reads are issued at the start of that tile)
    half8 A1[2], B2;
    float wev[8], wod[8];                // code values of the even tile / of the odd tile that follows it
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    // item operand of the logit MFMAs of tile (u, t): read two tiles before its use (the LDS latency used to sit between
    // the read and the MFMA at the top of every tile)
    auto item_op = [&](auto uc, auto tc) {
        constexpr int u = decltype(uc)::value, t = decltype(tc)::value;
        return *reinterpret_cast<const half8*>(&wl.img[u][0] + b1ofs + t * kMsItemRow);
    };
    auto logits = [&](const half8 b1, f32x4& d0, f32x4& d1) {
        d0 = mfma16(A1[0], b1, zero4);
        d1 = mfma16(A1[1], b1, zero4);
    };
    struct KTileOps { half8 a3[2][2], b3; };
    auto ktile_read = [&](auto uc, auto ktc, KTileOps& ko) {
        constexpr int u = decltype(uc)::value, kt = decltype(ktc)::value;
        const _Float16* rp = &wl.tr[0][0][0] + rofs;
        const _Float16* ip = &wl.img[u][0] + b3ofs + 2 * kt * kMsItemRow;
        ko.b3 = cat8(lds_tr16(ip), lds_tr16(ip + kMsItemRow));
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const _Float16* r0 = rp + 512 * (2 * hl + mt);
                ko.a3[hl][mt] = cat8(lds_tr16(r0), lds_tr16(r0 + 256));
            }
    };
    auto ktile_mfma = [&](const KTileOps& ko) {
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc_gt[mt] = mfma16(ko.a3[hl][mt], ko.b3, acc_gt[mt]);
    };
    // tile (u, t) on the logits d0 | d1; n0 | n1 receive the next tile's logits
    // bop: item operand of the NEXT tile's logits (read earlier); bnx receives the operand of the tile after that
    auto tile = [&](auto uc, auto tc, const f32x4 d0, const f32x4 d1, f32x4& n0, f32x4& n1, const uint32_t (&cw)[8],
                    const half8 bop, half8& bnx) {
        constexpr int u = decltype(uc)::value, t = decltype(tc)::value;
        constexpr bool last = u == 1 && t == 3;
        // The two waves of a SIMD (q and q + 4) run the same stream; the arbiter favours the older one, which then reaches the
        // batch's barrier ~4 000 cycles early while the other finishes alone at a single wave's issue rate (phase timing:
        // 12.4 k vs 16.2 k cycles per batch).  Swapping the leader every tile keeps the pair within a tile of each other.
        // (A/B on one box, ability_dim 8: 1.053 ms against 1.076 without; swapping every half tile 1.071, every two tiles 1.065.
        //  Round 6: the alternation has no restoring force -- a wave one tile ahead of its partner sits in a tile of the SAME
        //  priority, the tie goes to the older wave -- and waves 0-3 do reach the barrier ~3 k cycles before waves 4-7; a feedback
        //  form -- every wave publishes its tile counter in LDS, reads its partner's, the one behind takes the priority -- was
        //  built and measured: 952 -> 1 112 us per 1M x 1k call, the eight extra LDS round trips and scalar compares per batch
        //  cost far more than the balance wins.)
        // (s_setprio takes an immediate: set one value, skip the other for half of the waves -- one short forward branch)
        if constexpr (CODES) {
            // Cell-code rows (round 6): priority FALLS with the wave's own progress through the batch (4 levels) -- a wave that has
            // run ahead of its SIMD partner sits at a lower level than the partner: a restoring force without communication (the
            // alternation below has none: a wave one tile ahead sits in a tile of the same priority and wins the tie by age).
            // At equal progress the two staircases are offset by one tile, so the lead still alternates.  Same-box A/B, 1M x 1k:
            // cell codes 671 -> 645 us (-3.9 %); fp32 rows 817 -> 832 (ability_dim 1), ~930 -> 965 (8): they keep the alternation
            // (their tiles carry the next batch's row loads).  Three more staircases were measured on fp32 rows -- restarting per
            // u-step 840 / 984 us, the offset given to the older waves 823 / 957, two levels 822 / 954, against 818 / 960 for the
            // alternation (ability_dim 1 / 8): none wins there.
            constexpr int n = u * 4 + t;
            constexpr int pa = 3 - ((n + 1) >> 1) < 0 ? 0 : 3 - ((n + 1) >> 1);      // waves 0-3 (the older ones: they win ties)
            constexpr int pb = 3 - (n >> 1);                                            // waves 4-7
            asm volatile("s_bitcmp1_b32 %0, 2\n\ts_setprio %1\n\ts_cbranch_scc0 .Lmsprio%=\n\ts_setprio %2\n.Lmsprio%=:" :: "s"(q), "n"(pa), "n"(pb) : "scc");
        } else if constexpr (((u * 4 + t) & 1) != 0)
            asm volatile("s_bitcmp1_b32 %0, 2\n\ts_setprio 0\n\ts_cbranch_scc0 .Lmsprio%=\n\ts_setprio 1\n.Lmsprio%=:" :: "s"(q) : "scc");
        else
            asm volatile("s_bitcmp1_b32 %0, 2\n\ts_setprio 1\n\ts_cbranch_scc0 .Lmsprio%=\n\ts_setprio 0\n.Lmsprio%=:" :: "s"(q) : "scc");
        if constexpr (!last) logits(bop, n0, n1);
        // (tile (u, t) reads the operand of tile + 2; the batch's last two tiles read those of the next batch's first two)
        {
            constexpr int lin = (u * 4 + t + 2) & 7;
            bnx = item_op(std::integral_constant<int, (lin >> 2)>{}, std::integral_constant<int, (lin & 3)>{});
        }
        // K-tile whose pieces were completed by the previous tile: (u, 0) at t = 2, (0, 1) at (1, 0)
        constexpr bool pend = GRAD && (t == 2 || (u == 1 && t == 0));
        const float lg[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
        float gl[8];
        float pr0 = 1.0f, pr1 = 1.0f;
        // the codes of tiles t and t + 1 are converted together at the even tile (one instruction per person and pair)
        if constexpr ((t & 1) == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float2v w2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)cw[k], t == 2);
                wev[k] = w2[0];
                wod[k] = w2[1];
            }
        }
        const float (&wc)[8] = (t & 1) ? wod : wev;
        if constexpr (IRT != 3) {
            // wc = -w.  exponent e = -w l; ll = -log2(1 + 2^e); d ll/d l = w 2^e / (1 + 2^e) = wc / (1 + 2^e) - wc.
            // The reference's probability clamp (utils.py:46-49 -> torch) only matters for |logit| > 15.94: value capped at
            // +-kLogitLo, gradient exactly zero outside [-kLogitLo, kLogitHi] -- a wave-uniform slow path; everywhere else the
            // plain formula is the reference's.
            // The tile's largest |logit| by four 3-input maxima (hipcc's own lowering of the fmaxf chain took six instructions),
            // the plain formula for all cells first, and ONE branch to the slow path that overrides it.
            // (the maxima read the exponents u = -w l, not the logits: the MFMA -> VALU wait states are the compiler's business
            //  -- it knows nothing about registers read inside an asm statement -- and a missing cell, u = 0, never triggers)
            float tt[8], uu[8], ee[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uu[k] = wc[k] * lg[k];
                ee[k] = fast_exp2(uu[k]);
                tt[k] = 1.0f + ee[k];
            }
            {
                // products of four (one log2 per 4 terms, <= (1 + 2^23)^4) -- and ONE reciprocal per two cells: with
                // p = t_a t_b,  1 / t_a = t_b / p  and  1 / t_b = t_a / p  (two packed multiplies instead of a second
                // quarter-rate v_rcp; ~2 ulp instead of 1)
                const float2v t01 = {tt[0], tt[1]}, t23 = {tt[2], tt[3]}, t45 = {tt[4], tt[5]}, t67 = {tt[6], tt[7]};
                const float2v pa = t01 * t23, pb = t45 * t67;          // (t0 t2, t1 t3), (t4 t6, t5 t7)
                const float2v pp = pa * pb;
                pr0 = pp[0]; pr1 = pp[1];
                if constexpr (GRAD) {
                    const float2v ra = {fast_rcp(pa[0]), fast_rcp(pa[1])}, rb = {fast_rcp(pb[0]), fast_rcp(pb[1])};
                    const float2v r01 = ra * t23, r23 = ra * t01, r45 = rb * t67, r67 = rb * t45;
                    const float rr[8] = {r01[0], r01[1], r23[0], r23[1], r45[0], r45[1], r67[0], r67[1]};
#pragma unroll
                    for (int k = 0; k < 8; ++k) gl[k] = fmaf(wc[k], rr[k], -wc[k]);
                }
            }
            float lmax;
            asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(lmax) : "v"(uu[0]), "v"(uu[1]), "v"(uu[2]));
            asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(lmax) : "v"(lmax), "v"(uu[3]), "v"(uu[4]));
            asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(lmax) : "v"(lmax), "v"(uu[5]), "v"(uu[6]));
            asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(lmax) : "v"(lmax), "v"(uu[7]));
            if (__any(!(lmax <= kLoS))) {                              // (wave-uniform, rare)
                pr0 = pr1 = 1.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float& pr = (k & 1) ? pr1 : pr0;
                    // exp2 is monotone and exact at +-23: clamping e = 2^u to [2^-23, 2^23] IS the clamp of the logit -- no second
                    // exponential; the gradient's zero band (l < -15.94 or l > 16.64) as "clamping l to it changes l".
                    // (A saturated item sends its wave through here in every batch and the other waves wait for it at the
                    //  barrier: a freshly initialised model has a few such items, ~10 % of the kernel before this was trimmed.)
                    const float tk = 1.0f + med3(ee[k], 0x1p-23f, 0x1p23f);
                    pr *= tk;
                    if constexpr (GRAD) gl[k] = (med3(lg[k], -kLoS, kHiS) != lg[k]) ? 0.f : fmaf(wc[k], fast_rcp(tk), -wc[k]);
                }
            }
        } else {
#pragma clang fp contract(off)
            // (no implicit contraction in this block: whether hipcc fuses `n iq - x` into one fma depends on the code around it,
            //  and the row modes -- which schedule the tile differently, see below -- have to agree bit for bit; the fused
            //  forms below are written out)
            // 3PL (models.py:758-765):  p = guess + (1 - guess) sigmoid(l).  With E = 2^u, u = -w l (the 2PL exponent):
            //   answered right:  p     = (1 + guess E) / (1 + E)        answered wrong:  1 - p = (1 - guess) / (1 + E)
            // i.e. both are v = n / t with t = 1 + E and n = 1 + guess sel, sel = E (right) | -1 (wrong) | 0 (missing: v = 1/2, a
            // log2 of exactly -1 that the count of unobserved cells takes back, as in 1PL/2PL).  ONE exponential and ONE reciprocal
            // (of n t: 1/t = n / (n t), 1/n = t / (n t)) per cell instead of an exponential, two reciprocals and the sign selects of
            // the sigmoid-first form (rounds 2-4: ~20 plain + 3.25 quarter-rate instructions per cell; now ~12 + 2.25):
            //   d ll/d l      = wc (1/t - [right] 1/n - [wrong or missing] 1)       (log2 units, wc = -w)
            //   d ll/d guess  = sel / n;  its chain factor guess (1 - guess) is applied once per item in the epilogue.
            // The reference's probability clamp (p to [eps32, 1 - eps32], gradient zero where it bites) is "v left [eps32, 1 - eps32]"
            // for right and wrong cells alike: a wave-uniform second path over the tile, as in 2PL.
            // (the address goes through an empty asm so that the loop-invariant read is not hoisted back into registers)
            int go = (u * 4 + t) * 16 + i16;
            asm volatile("" : "+v"(go));
            const float gs_ut = gsl[go];
            // (Gathered rows: the four cells of an M-tile at a time, everything consumed on the spot -- an empty asm after each
            //  half.  With all eight in flight the reciprocals, selections and ratios of the tile -- 40 registers -- sit on top of
            //  the logits and codes the second path needs; until the translation units were built with
            //  -sink-insts-to-avoid-spills (Makefile) that spilled 40-430 registers in the flow / hook / gathered instantiations and
            //  every row mode kept the halves apart.  With the flag the 3PL kernels use 211-241 registers and the other row modes
            //  let the compiler interleave the eight cells: 3PL 1M x 1k 1.27 -> 1.24 ms; gathered rows 1.32 -> 1.37, so they
            //  keep the pins.)
            float dgs = 0.f;                              // the tile's d ll / d guess (added to the accumulator once, below)
            bool redo = sat3;
            if (!sat3) {
                float vhi = 0.f, vlo = 1.0f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float vq[4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const int k = 4 * h + k4;
                        const float ek = fast_exp2(wc[k] * lg[k]);
                        const float tk = 1.0f + ek;
                        const bool right = wc[k] < 0.f;
                        const float sel = right ? ek : -wc[k];
                        const float nk = fmaf(gs_ut, sel, 1.0f);
                        const float iq = fast_rcp(nk * tk);
                        const float rtk = nk * iq, rnk = tk * iq;
                        vq[k4] = nk * rtk;
                        if constexpr (GRAD) {
                            gl[k] = wc[k] * fmaf(nk, iq, -(right ? rnk : 1.0f));
                            dgs = fmaf(sel, rnk, dgs);
                        }
                    }
                    pr0 *= vq[0] * vq[2];
                    pr1 *= vq[1] * vq[3];
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(vhi) : "v"(vhi), "v"(vq[0]), "v"(vq[1]));
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(vhi) : "v"(vhi), "v"(vq[2]), "v"(vq[3]));
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(vlo) : "v"(vlo), "v"(vq[0]), "v"(vq[1]));
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(vlo) : "v"(vlo), "v"(vq[2]), "v"(vq[3]));
                    if constexpr (RM == 1) {              // (gathered rows only: see above)
                        if constexpr (GRAD) {
                            asm volatile("" : "+v"(gl[4 * h]), "+v"(gl[4 * h + 1]), "+v"(gl[4 * h + 2]), "+v"(gl[4 * h + 3]), "+v"(dgs), "+v"(pr0), "+v"(pr1));
                        } else {
                            asm volatile("" : "+v"(pr0), "+v"(pr1));
                        }
                    }
                }
                // Out of [eps32, 1 - eps32] anywhere in the wave's tile?  (The extrema, not the products: a product of four below
                // eps32 says nothing -- four answers of probability 2 % in one lane's quad get there, which simulated or badly
                // fitted responses do all the time: config 5's benchmark matrix sent every tile down the second path that way.)
                // min3 / max3 drop a NaN operand (an exponential that overflowed: inf / inf), the products keep it.
                if (__any(!(vlo >= kEps32 && vhi <= 1.0f - kEps32 && pr0 * pr1 > 0.f))) redo = sat3 = true;      // (wave-uniform)
            }
            if (redo) {
                // The same quantities with the clamp applied cell by cell: v clamped IS the reference's clamp of p (right: v = p;
                // wrong: v = 1 - p, and 1 - clamp(p) = clamp(1 - p)), gradient zero where it bit; the exponent is held inside
                // +-60 first (n t <= 2^121; beyond it v is 0, 1, guess or 1 - guess to the last bit anyway).  Unclamped cells
                // come out bit for bit as above.  Not rare everywhere: ONE cell in the wave's 512 sends a tile here, which with
                // flows that push theta out, or items past |logit| 16, is most tiles -- so a wave that got here stays here for
                // the rest of the batch (sat3) and pays both forms once per batch at most.  (Sticky for the whole kernel was
                // measured too: one early hit in any of its 8 waves then slows a workgroup for good -- they meet at two barriers
                // per batch -- and the plain 3PL call lost most of what the first form wins: 1.25 -> 1.33 ms per 1M x 1k.)
                pr0 = pr1 = 1.0f;
                dgs = 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float vq[4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const int k = 4 * h + k4;
                        const float ek = fast_exp2(med3(wc[k] * lg[k], -60.f, 60.f));
                        const float tk = 1.0f + ek;
                        const bool right = wc[k] < 0.f;
                        const float sel = right ? ek : -wc[k];
                        const float nk = fmaf(gs_ut, sel, 1.0f);
                        const float iq = fast_rcp(nk * tk);
                        const float rtk = nk * iq, rnk = tk * iq;
                        const float v = nk * rtk;
                        vq[k4] = med3(v, kEps32, 1.0f - kEps32);
                        if constexpr (GRAD) {
                            const bool keep = vq[k4] == v;
                            gl[k] = keep ? wc[k] * fmaf(nk, iq, -(right ? rnk : 1.0f)) : 0.f;
                            dgs = fmaf(keep ? sel : 0.f, rnk, dgs);
                        }
                    }
                    pr0 *= vq[0] * vq[2];               // (the first form's order of products and sums: same bits)
                    pr1 *= vq[1] * vq[3];
                    if constexpr (GRAD) {
                        asm volatile("" : "+v"(gl[4 * h]), "+v"(gl[4 * h + 1]), "+v"(gl[4 * h + 2]), "+v"(gl[4 * h + 3]), "+v"(dgs), "+v"(pr0), "+v"(pr1));
                    } else {
                        asm volatile("" : "+v"(pr0), "+v"(pr1));
                    }
                }
            }
            if constexpr (GRAD) acc_g[u][t] += dgs;
        }
        // (1PL/2PL: pr = products of t = 1 + E >= 1; 3PL: of v = n / t <= 1 -- s_log is the negative log-likelihood in log2 units
        //  plus one per unobserved cell either way)
        if constexpr (IRT != 3) s_log += fast_log2(pr0) + fast_log2(pr1);
        else s_log -= fast_log2(pr0) + fast_log2(pr1);
        asm volatile("" : "+v"(s_log));          // (keeps hipcc from sinking the whole batch's products to the loop end)
        if constexpr (GRAD) {
            acc_b[u][t] += ((gl[0] + gl[1]) + (gl[2] + gl[3])) + ((gl[4] + gl[5]) + (gl[6] + gl[7]));
            asm volatile("" : "+v"(acc_b[u][t]));
            if constexpr (IRT == 3) asm volatile("" : "+v"(acc_g[u][t]));
            KTileOps ko;              // (the transposed reads fly under the split below)
            if constexpr (pend) ktile_read(std::integral_constant<int, (t == 2 ? u : 0)>{}, std::integral_constant<int, (t == 2 ? 0 : 1)>{}, ko);
            // f16 hi/lo pieces of g: hi = rtz(g), lo = f16(g - hi) (lo_pieces4: one hazard-safe asm block, see vibo_device.hpp)
            half2v hh[4], ll[4];
            {
                uint32_t hw[4], lw[4];
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    hh[k2] = pkrtz(gl[2 * k2], gl[2 * k2 + 1]);
                    hw[k2] = __builtin_bit_cast(uint32_t, hh[k2]);
                }
                lo_pieces4(hw, gl, lw);
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) ll[k2] = __builtin_bit_cast(half2v, lw[k2]);
            }
            const half8 a2h = cat8(hh[0], hh[1], hh[2], hh[3]), a2l = cat8(ll[0], ll[1], ll[2], ll[3]);
            if constexpr (pend) ktile_mfma(ko);
            if constexpr (IRT != 1) {
                acc_ga[u][t] = mfma16(a2h, B2, acc_ga[u][t]);
                acc_ga[u][t] = mfma16(a2l, B2, acc_ga[u][t]);
            }
            // LDS image for the transposed read: persons of M-tile 0 = registers 0-1, M-tile 1 = registers 2-3
            _Float16* wp = &wl.tr[0][0][0] + wofs + 256 * (t & 1);
            *reinterpret_cast<uint2*>(wp) = uint2{__builtin_bit_cast(uint32_t, hh[0]), __builtin_bit_cast(uint32_t, hh[1])};
            *reinterpret_cast<uint2*>(wp + 512) = uint2{__builtin_bit_cast(uint32_t, hh[2]), __builtin_bit_cast(uint32_t, hh[3])};
            *reinterpret_cast<uint2*>(wp + 1024) = uint2{__builtin_bit_cast(uint32_t, ll[0]), __builtin_bit_cast(uint32_t, ll[1])};
            *reinterpret_cast<uint2*>(wp + 1536) = uint2{__builtin_bit_cast(uint32_t, ll[2]), __builtin_bit_cast(uint32_t, ll[3])};
            if constexpr (last) {                 // the batch's last K-tile: nothing left to hide it under
                KTileOps kl;
                ktile_read(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, kl);
                ktile_mfma(kl);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using IC0 = std::integral_constant<int, 0>;
    using IC1 = std::integral_constant<int, 1>;
    using IC2 = std::integral_constant<int, 2>;
    using IC3 = std::integral_constant<int, 3>;
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
        const half8 ones = half8{one_b, one_b, one_b, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const half8 v = *reinterpret_cast<const half8*>(&cl.thA[g >> 1][16 * mt + i16][0]);
            A1[mt] = (g == 3) ? ones : v;
        }
        B2 = *reinterpret_cast<const half8*>(&cl.thT[i16][8 * g]);
    };

    // ================= prologue: first batch =================
    uint32_t cwA0[8], cwA1[8], cwB0[8], cwB1[8];
    int pk[4];
    if (bt < n_batches) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = 0;
        if (late) {
            // (a batch of slack: the rows are asked for once the long workgroups' first batches -- rem_wg x the bytes of a batch at
            //  ~5 TB/s, + 1 us of latency -- are through, 6 us at most; ticks of 10 ns)
            const long long batch_bytes = (long long)R * n4 * (CODES ? 4 : 20);
            long long wait_ticks = (long long)rem_wg * batch_bytes / 50000 + 100;
            wait_ticks = wait_ticks > 600 ? 600 : wait_ticks;
            while ((long long)(realtime_ticks() - t_wave_start) < wait_ticks) __builtin_amdgcn_s_sleep(8);
            first_rows();
        }
        pack_half(bt, 0, cwA0, cwA1, pk, x, m, src_first);
        MS_P(22, t_entry)
        pack_half(bt, 1, cwA0, cwA1, pk, x2, m2, src_first);
        MS_P(23, t_entry)
        put_counts(pk, true);
        put_stats();
        asm volatile("" : "+v"(epn));                 // (in before the loop: no wait on it behind the loop's own loads)
        if constexpr (EXTRA && kPrs) asm volatile("" : "+v"(prs0), "+v"(prs1), "+v"(prs2));
        if constexpr (XCOND && kPrs) asm volatile("" : "+v"(prc));
    }
    int par = 0;                                      // parity of the workgroup's batch counter: LDS double buffers, slot owners
    // item operands of the first two tiles' logits (every batch reads the same image: carried across the back edge)
    half8 bopA = item_op(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    half8 bopB = item_op(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
#ifdef VIBO_MS_TIMING
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast) :: "memory");
    tacc[12] = tlast - t_entry;                      // kernel prologue (operand images, first batch)
    tacc[14] = t_real_entry;                         // wall clock (100 MHz) at entry
#endif
    // One iteration = [request the first half of the next batch's rows] [sync phase of this batch: counts -> theta, with the
    // backward of the previous batch in the waves that have no forward slot] [math of this batch, with the next batch's rows
    // packed / requested under it].
    // The rows travel in two halves (M-tile 0, M-tile 1), each requested at least four tiles before it is packed; the first
    // goes out right before the sync phase (the texture pipeline digests the workgroup's 128 load instructions while the
    // waves sit in the barriers; issued after them, with all 8 waves in step, every wave stalled ~5000 cycles on it).
    // (Splitting the bursts by wave group -- the four waves without a slot between the barriers, the slot owners after the
    // second one, the second half one tile apart -- was measured too: 1.098 vs 1.074 ms.)
    // Nothing loaded is in flight across the loop's back edge: hipcc otherwise parks such registers in a second set and
    // copies them at the back edge behind an s_waitcnt vmcnt(0), which serialises the prefetch.  The sched_barriers keep
    // the loads behind the pack that frees their registers.
    for (; bt < n_batches; bt += G, par ^= 1) {
        const int nxt = bt + G;
        sat3 = false;                                 // (3PL: see the tile)
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = 0;
        if constexpr (RM != 0) {
            if (p.row_index) {
#pragma unroll
                for (int k = 0; k < 8; ++k) ridx[k] = ridx_n[k];      // (requested an iteration ago)
                fetch_idx(nxt + G, ridx_n);
            }
        }
        const RowSrc sn = row_src(nxt);
        auto burst_a = [&]() {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) load_quarter(nxt, sn, 0, j, x, m);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto pack_a_burst_b = [&]() {
            pack_half(nxt, 0, cwB0, cwB1, pk, x, m, sn);
            fetch_eps(nxt, par ^ 1);                  // (complete by the second pack: free to carry across the back edge)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) load_quarter(nxt, sn, 1, j, x, m);
            __builtin_amdgcn_sched_barrier(0);
        };
        // (Measured and left alone, round 6: these four requests woven one at a time between
        // put_counts / put_gtheta / the cell-word moves of the previous iteration's tail, one batch
        // further ahead -- 843.9 vs 838.7 us at 1M x 1000 A=8, 797 vs 796 at A=1: nothing; the same
        // with the falling tile priority of the cell-code rows on top -- 858 / 806 us: slower.)
        burst_a();
        MS_T(4)
        __syncthreads();                              // counts of bt and d LL/d theta shares of bt - G are out
        MS_T(5)
        person_forward(bt, par);                      // short: counts -> theta operands (eps came a batch ahead)
        MS_T(6)
        if constexpr (GRAD && NW8) {
            // 8 waves: the four waves without a forward slot in this batch own the backward slots of the previous one -- they
            // run it here, while the others compute theta, instead of after the second barrier (where it delayed their math
            // by ~1 000 cycles and the whole workgroup at the next barrier)
            if (bt >= G + wg) person_backward(bt - G, par ^ 1);
        }
        MS_T(9)
        __syncthreads();
        MS_T(7)
        read_theta_ops();
        MS_T(8)
        if constexpr (GRAD && !NW8) {
            if (bt >= G + wg) person_backward(bt - G, par ^ 1);   // nobody waits for this
        }
        f32x4 da0, da1, db0, db1;
        logits(bopA, da0, da1);
        if constexpr (RM == 0 && IRT != 3) {
            // fp32 rows in order, 1PL / 2PL (round 6): the next batch's M-tile 0 quarter j is packed, and its M-tile 1 quarter j requested into the
            // registers that frees, IN FRONT OF tile (0, j) -- four load instructions per tile instead of a burst of sixteen behind
            // tile (0, 3).  The CU's address path takes a load instruction per ~16 cycles: when the eight waves' bursts meet there,
            // each wave sits in its burst for 1.3-2.6 k cycles (tools/ms_timing.py, `pack0`; under a priority scheme that keeps
            // the wave pairs level: 3.3 k) and cannot issue anything else meanwhile.  Same-box A/B, 1M x 1k: ability_dim 8
            // 876 -> 861 / 950 -> 916 us (two boxes), ability_dim 1 822 -> 800; 125 000 x 1 000 124.5 -> 120.4.  Cell codes
            // (4-byte loads) and gathered fp32 rows (per-row addresses formed at the load) measured the other way round -- 645 -> 691 us,
            // 1.23 -> 1.34 ms -- and keep the burst, as does the 3PL tile (1.248 -> 1.259 ms).  Caller-supplied posterior 1.135 ->
            // 1.105, forward only 0.970 -> 0.927, wide rows 1.654 -> 1.635.  Also measured: the M-tile 0 requests
            // spread over tiles (1, j) as well -- loads in flight across the loop's back edge -- +-0 against this (885 vs 884,
            // 809 vs 803); the slower wave group (4-7) issuing its M-tile 0 burst behind the first barrier: 863 vs 858, 810 vs 805.
            auto side = [&](const int j) __attribute__((always_inline)) {
                pack_one(nxt, 0, j, cwB0, cwB1, pk, x, m, sn);
                __builtin_amdgcn_sched_barrier(0);
                load_quarter(nxt, sn, 1, j, x, m);
                __builtin_amdgcn_sched_barrier(0);
            };
            side(0);
            tile(IC0{}, IC0{}, da0, da1, db0, db1, cwA0, bopB, bopA);     // (reads the operand of (0, 2) into bopA, ...)
            side(1);
            tile(IC0{}, IC1{}, db0, db1, da0, da1, cwA0, bopA, bopB);
            side(2);
            tile(IC0{}, IC2{}, da0, da1, db0, db1, cwA0, bopB, bopA);
            side(3);
            tile(IC0{}, IC3{}, db0, db1, da0, da1, cwA0, bopA, bopB);
            MS_T(0)
            fetch_eps(nxt, par ^ 1);                  // (complete by the second pack: free to carry across the back edge)
            MS_T(1)
        } else {
            tile(IC0{}, IC0{}, da0, da1, db0, db1, cwA0, bopB, bopA);
            tile(IC0{}, IC1{}, db0, db1, da0, da1, cwA0, bopA, bopB);
            tile(IC0{}, IC2{}, da0, da1, db0, db1, cwA0, bopB, bopA);
            tile(IC0{}, IC3{}, db0, db1, da0, da1, cwA0, bopA, bopB);
            MS_T(0)
            pack_a_burst_b();
            MS_T(1)
        }
        tile(IC1{}, IC0{}, da0, da1, db0, db1, cwA1, bopB, bopA);
        tile(IC1{}, IC1{}, db0, db1, da0, da1, cwA1, bopA, bopB);
        tile(IC1{}, IC2{}, da0, da1, db0, db1, cwA1, bopB, bopA);     // (... of the next batch's (0, 0) into bopA)
        tile(IC1{}, IC3{}, db0, db1, da0, da1, cwA1, bopA, bopB);     // (not used: last; reads (0, 1) into bopB)
        MS_T(2)
        // (XM == 3: the rest of the iteration above the partner wave's tiles.  With the experts' sums in the packs, the wave that leaves
        //  its tiles first -- waves 0-3, see the tile -- has ~3 k cycles of pack and halving sums in front of the batch barrier, at the
        //  tile's last priority (0) against its partner's tiles: tools/ms_timing.py showed 7.2 k cycles there against the partner's
        //  4.3 k, and the partner waiting.  1M x 1k: 1.625 -> 1.600 ms.  The falling-priority staircase of the cell-code rows on top: +-0.)
        if constexpr (XFUSE) __builtin_amdgcn_s_setprio(2);
        pack_half(nxt, 1, cwB0, cwB1, pk, x, m, sn);
        asm volatile("" : "+v"(epn));                 // (eps is in: nothing is pending at the back edge)
        if constexpr (EXTRA && kPrs) asm volatile("" : "+v"(prs0), "+v"(prs1), "+v"(prs2));
        if constexpr (XCOND && kPrs) asm volatile("" : "+v"(prc));
        if constexpr (RM != 0) {                      // (nor the row numbers: they were requested before this batch's rows)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(ridx_n[k]));
        }
        MS_T(3)
        put_counts(pk, nxt < n_batches);
        MS_T(10)
        put_stats();
        MS_T(11)
        if constexpr (GRAD) put_gtheta(par);
#pragma unroll
        for (int k = 0; k < 8; ++k) { cwA0[k] = cwB0[k]; cwA1[k] = cwB1[k]; }
    }
#ifdef VIBO_MS_TIMING
    long long t_loop_end;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_loop_end) :: "memory");
#endif
    if constexpr (GRAD) {
        // backward of the workgroup's last batch
        __syncthreads();
        if (bt >= G + wg) person_backward(bt - G, par ^ 1);
    }

    // ================= workgroup reduction -> partial record =================
    // The lane-derived indices are formed afresh here (from an opaque copy of the thread id): shared with the prologue's, the ones
    // the batch loop has no use for stayed live across it -- in the instantiations at the 256-register limit they were spilled
    // to scratch (the kernel trace's `scratch` column), however rarely reloaded.
    MS_P(24, t_loop_end)
    int tid_e = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, i16_e = lane_e & 15, g_e = lane_e >> 4;
    float* out = p.partial + (size_t)rec * p.lay.stride;
    {
        // every cell without an observation contributed exactly log2(1 + 2^0) = 1 to s_log
        const float ll = -kLn2 * wave_total(s_log - (float)unobs);
        if (lane_e == 0) wl.red[0] = range_fault ? __builtin_nanf("") : ll;      // (operands beyond the rescaling range: loud)
    }
    __syncthreads();
    MS_P(25, t_loop_end)
    // scalars: 0 ll | 1 kl | 2 logq0 | 3 logp | 4 ladj | 5 nobs
    if (tid_e == 0) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += wls[w].red[0];
        out[0] = t;
        float la = 0.f;
        if constexpr (FLOWS) {
            for (int par2 = 0; par2 < 2; ++par2)
                for (int e = 0; e < kMsRows; ++e) la += fl.lacc[par2][e];
        }
        out[4] = la; out[6] = 0.f; out[7] = 0.f;
    }
    // sums of the (person, dim) pairs' running terms, fixed order: wave w takes term 8 + w (, 8 + w + nw, ...) -- every lane its
    // eight values, then the wave total (one thread per term walked 512 LDS values in a row: ~10 k cycles at the end of
    // every workgroup)
    for (int k = 8 + q; k < 12; k += nw) {
        float v = 0.f;
#pragma unroll
        for (int par2 = 0; par2 < 2; ++par2)
#pragma unroll
            for (int j = 0; j < 4; ++j) v += cl.tacc[par2][k][lane_e + 64 * j];
        const float t = wave_total(v);
        if (lane_e == 0) out[k == 8 ? 1 : k == 9 ? 2 : k == 10 ? 3 : 5] = t;
    }
    MS_P(26, t_loop_end)
    if constexpr (GRAD) {
        // table gradient: term k (< 8) of ability dim a = the sum of the 2 x 32 (person, dim a) pairs' running sums.  Wave q takes
        // terms q, q + nw, ...: every lane adds its eight values (pairs e = lane + 64 j keep the lane's dim, e & 7), then the eight
        // lanes of a dim fold (fixed order: bitwise reproducible).  (Rounds 2-5: 8 A threads of wave 0 walked 64 LDS values each,
        // ~6 k cycles on the critical path of every workgroup's exit -- tools/ms_timing.py, end-code marks.)
        for (int k = q; k < 8; k += nw) {
            float v = 0.f;
#pragma unroll
            for (int par2 = 0; par2 < 2; ++par2)
#pragma unroll
                for (int j = 0; j < 4; ++j) v += cl.tacc[par2][k][lane_e + 64 * j];
            v += dpp_f<0x128>(v);               // row_ror 8: lanes l and l + 8 of a row
            v = xor16_add(v);
            v = xor32_add(v);
            const int a = lane_e & 7;
            const int st = k >> 2, c = (k >> 1) & 1, ms = k & 1;
            if (lane_e < 8 && a < A) out[p.lay.off_table + (st * 2 + c) * 2 * A + ms * A + a] = v;
        }
        if constexpr (FLOWS) {
            const int per = 2 * A + 1;
            for (int idx = tid_e; idx < 2 * p.n_flows * per; idx += (int)blockDim.x) {
                const int st = idx / (p.n_flows * per), f = (idx / per) % p.n_flows, j = idx % per;
                const int kind = j < A ? 0 : j < 2 * A ? 1 : 2;
                const int a = kind == 0 ? j : kind == 1 ? j - A : 0;
                float t = 0.f;
                for (int k = 0; k < 8; ++k) t += fl.facc[k >> 2][k & 3][st][f][kind][a];
                out[p.lay.off_flow + idx] = t;
            }
        }
        // item gradients.  d LL/d a: lane (col i16, g) holds items 4 g + j of tile (u, t): cols a and 8 + a add up.
        // The wave's 128 items x (A + 2) rows go through its own LDS (the operand images are dead by now; nobody else reads
        // them: no barrier) and leave as whole 256-byte rows -- the registers' own layout is 4-byte stores scattered over 32
        // cache lines each, ~5 us at the end of every workgroup.
        MS_P(27, t_loop_end)
        constexpr int kStage = 130;                  // floats per staged row (128 + 2: the 32 writers of a row group hit 32 banks)
        float* stage = reinterpret_cast<float*>(&wl.tr[0][0][0]);
        static_assert(sizeof(wl.tr) + sizeof(wl.img) >= (size_t)(VIBO_MAX_ABILITY_DIM + 2) * kStage * sizeof(float), "staging area");
        const int brow = IRT == 1 ? 0 : A;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (IRT != 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = (acc_ga[u][t][j] + dpp_f<0x128>(acc_ga[u][t][j])) * sc_a;       // sum g theta' = 2^-jsh sum g theta
                        if (i16_e < A) stage[i16_e * kStage + 64 * u + 4 * (4 * g_e + j) + t] = -v;
                    }
                }
                // d LL/d b (and d/d guess-logit): the lane's 8 persons per batch -> sum over the 4 lane groups
                float b = acc_b[u][t];
                b = xor16_add(b);
                b = xor32_add(b);
                if (g_e == 0) stage[brow * kStage + 64 * u + 4 * i16_e + t] = b;
                if constexpr (IRT == 3) {
                    float gg = acc_g[u][t];
                    gg = xor16_add(gg);
                    gg = xor32_add(gg);
                    // (the tiles summed d ll / d guess; the parameter is the guess logit: x guess (1 - guess), models.py:758)
                    const float gs_e = gsl[(u * 4 + t) * 16 + i16_e];
                    if (g_e == 0) stage[(A + 1) * kStage + 64 * u + 4 * i16_e + t] = gg * (gs_e * (1.0f - gs_e));
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave's own LDS writes, in order: visible to its reads)
        MS_P(28, t_loop_end)
        const int n_rows = IRT == 1 ? 1 : IRT == 2 ? A + 1 : A + 2;
        for (int row = 0; row < n_rows; ++row) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int il = kMsSpan * q + 64 * h + lane_e;
                if (il < I) out[p.lay.off_item + (size_t)row * p.lay.i_pad + il] = stage[row * kStage + 64 * h + lane_e];
            }
        }
    }
    insitu_exit(p.insitu, gridDim.x);
#ifdef VIBO_MS_TIMING
    {
        long long t_exit, t_real_exit;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_exit), "=s"(t_real_exit) :: "memory");
        tacc[13] = t_exit - t_loop_end;              // kernel epilogue (records)
        tacc[15] = t_real_exit;
        if (lane == 0 && blockIdx.x < 1024) {
            for (int k = 0; k < kMsTSlots; ++k) g_ms_timing[((size_t)blockIdx.x * 8 + q) * kMsTSlots + k] = tacc[k];
        }
    }
#endif
}

template <int IRT, bool GRAD, int RM, bool FLOWS, bool NW8, int XM>
static hipError_t launch_msplit_inst(const ElboParams& p, int nw, int grid, hipStream_t s) {
    // more than 64 KB of dynamic LDS has to be opted into (once per kernel; gfx950 has 160 KB per CU)
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&msplit_kernel<IRT, GRAD, RM, FLOWS, NW8, XM>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)msplit_lds_bytes(8, FLOWS, IRT == 3));
        if (e != hipSuccess) return e;
        lds_opt_in = true;
    }
    hipLaunchKernelGGL((msplit_kernel<IRT, GRAD, RM, FLOWS, NW8, XM>), dim3(grid), dim3(64 * nw), msplit_lds_bytes(nw, FLOWS, IRT == 3), s, p);
    return hipGetLastError();
}
// XM == 3 (p.cond_table): its own translation units (vibo_msplit_xa / xg.hip)
template <int IRT, bool GRAD, int RM, bool NW8>
static hipError_t launch_msplit_fused_inst(const ElboParams& p, int nw, int grid, hipStream_t s) {
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&msplit_kernel<IRT, GRAD, RM, false, NW8, 3>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)msplit_lds_bytes(8, false, IRT == 3, true));
        if (e != hipSuccess) return e;
        lds_opt_in = true;
    }
    hipLaunchKernelGGL((msplit_kernel<IRT, GRAD, RM, false, NW8, 3>), dim3(grid), dim3(64 * nw), msplit_lds_bytes(nw, false, IRT == 3, true), s, p);
    return hipGetLastError();
}
template <int RM>
static hipError_t launch_msplit_fused(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    if (!p.cond_table || p.A != 1 || p.n_flows != 0 || p.row_cnt || p.pre_stats || p.given_post || p.given_grad || !p.primary || p.panel_count > 1 ||
        (grad && !p.post_coef)) return hipErrorInvalidValue;
#define VIBO_MSF(IRT_, GRAD_) (nw == 8 ? launch_msplit_fused_inst<IRT_, GRAD_, RM, true>(p, nw, grid, s) : launch_msplit_fused_inst<IRT_, GRAD_, RM, false>(p, nw, grid, s))
    if (irt == 1) return grad ? VIBO_MSF(1, true) : VIBO_MSF(1, false);
    if (irt == 2) return grad ? VIBO_MSF(2, true) : VIBO_MSF(2, false);
    return grad ? VIBO_MSF(3, true) : VIBO_MSF(3, false);
#undef VIBO_MSF
}
template <int IRT, bool GRAD, int RM, bool FLOWS>
static hipError_t launch_msplit_one(const ElboParams& p, int nw, int grid, hipStream_t s) {
    // XM: the hook mode (see the kernel): 2 = the slot lanes read / write a caller-supplied posterior, 1 = panel / conditional hooks;
    // NW8: exactly 8 waves per workgroup
    const int xm = (p.given_post || p.given_grad) ? 2
                   : (p.row_cnt || p.pre_stats || p.post_coef || !p.primary || p.panel_count > 1) ? 1 : 0;
    if (p.cond_table) return hipErrorInvalidValue;      // (XM == 3: launch_msplit_fused)
    if (xm == 2 && (!p.given_post || p.row_cnt || p.pre_stats || p.post_coef || !p.primary || p.panel_count > 1)) return hipErrorInvalidValue;
    if (nw == 8) return xm == 2 ? launch_msplit_inst<IRT, GRAD, RM, FLOWS, true, 2>(p, nw, grid, s)
                      : xm == 1 ? launch_msplit_inst<IRT, GRAD, RM, FLOWS, true, 1>(p, nw, grid, s)
                                : launch_msplit_inst<IRT, GRAD, RM, FLOWS, true, 0>(p, nw, grid, s);
    return xm == 2 ? launch_msplit_inst<IRT, GRAD, RM, FLOWS, false, 2>(p, nw, grid, s)
         : xm == 1 ? launch_msplit_inst<IRT, GRAD, RM, FLOWS, false, 1>(p, nw, grid, s)
                   : launch_msplit_inst<IRT, GRAD, RM, FLOWS, false, 0>(p, nw, grid, s);
}
template <int RM, bool FLOWS>
static hipError_t launch_msplit_rm(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    if (irt == 1) return grad ? launch_msplit_one<1, true, RM, FLOWS>(p, nw, grid, s) : launch_msplit_one<1, false, RM, FLOWS>(p, nw, grid, s);
    if (irt == 2) return grad ? launch_msplit_one<2, true, RM, FLOWS>(p, nw, grid, s) : launch_msplit_one<2, false, RM, FLOWS>(p, nw, grid, s);
    return grad ? launch_msplit_one<3, true, RM, FLOWS>(p, nw, grid, s) : launch_msplit_one<3, false, RM, FLOWS>(p, nw, grid, s);
}

}  // namespace vibo
