// vibo_device.hpp -- small device helpers shared by the VIBO kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vibo {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kLog2Pi = 1.8378770664093453f;
// torch.distributions.Bernoulli clamps probs to [eps32, 1-eps32] (utils.py:46-49):
constexpr float kEps32 = 1.1920928955078125e-07f;
constexpr float kLogitLo = 15.942384719848633f;   // sigmoid(l) < eps32 below -kLogitLo
constexpr float kLogitHi = 16.635532333438686f;   // 24 ln 2: 1/(1+exp(-l)) rounds to 1.0f above
constexpr float kPoeEps = 1e-8f;                  // product_of_experts eps (utils.py:105)

typedef float float2v __attribute__((ext_vector_type(2)));

// Read-only, wave-uniform data (item parameters) is fetched through the scalar
// cache: a pointer in the constant address space makes hipcc emit s_load_dword*
// instead of per-lane vector loads.
typedef const __attribute__((address_space(4))) float* const_f32_ptr;
__device__ __forceinline__ const_f32_ptr as_constant(const float* p) {
    return (const_f32_ptr)(uintptr_t)p;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// DPP move with the lanes a row_mask disables reading `old` (= 0 here).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

// Streaming accesses (every byte touched once by one lane: the conditional posterior's pre pass, the row-count pass): the
// non-temporal hint keeps them from displacing what the next kernel wants in L2 / MALL: cond_pre -2 %, the row-count pass -8 %.
// (NOT for the matrix kernel's row loads -- neighbouring waves share the cache lines at the edges of their 512-byte segments:
// +4..8 % with the hint -- nor for cm_forward_fp32, where four lanes share a row's 256 bytes: +30 %; the narrow-row kernel: +-1 %.)
typedef float vibo_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const vibo_f4v t = __builtin_nontemporal_load(reinterpret_cast<const vibo_f4v*>(p));
    return float4{t[0], t[1], t[2], t[3]};
}
__device__ __forceinline__ uint32_t nt_load1(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void nt_store1(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }

// v + (the value of the lane 16 / 32 away) through v_permlane16_swap / v_permlane32_swap (gfx950): VALU instructions, where
// `v += __shfl_xor(v, 16)` compiles to a ds_bpermute -- an LDS round trip and an s_waitcnt lgkmcnt in the middle of the stream.
// The instruction swaps the odd rows (upper half) of its first operand with the even rows (lower half) of its second: with two
// copies of v going in, the two registers coming out hold "own" and "partner" in one order or the other -- the same two addends
// as the shuffle form, the same bits (checked lane by lane on the hardware).  Inline asm: hipcc 7.2's
// __builtin_amdgcn_permlane16_swap hands back the FIRST result register twice.  (s_nop: the copies are fresh VALU results, and
// the compiler does not see which instruction reads them.)
// (gfx950 only: the swap instructions do not exist elsewhere, and the hazard argument above was checked on this target --
//  tests/test_gpu_parity.py::test_lane_swap_sums_equal_the_shuffle_form compares both forms lane by lane on the hardware)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "vibo_device.hpp: v_permlane16/32_swap (xor16_add / xor32_add) are gfx950 instructions -- build with --offload-arch=gfx950"
#endif
__device__ __forceinline__ float xor16_add(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float xor32_add(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// In-situ launch timer (vibo_set_insitu_timer, include/vibo_hip.h): the kernel's own duration measured by the kernel, inside whatever
// it runs in -- a replayed hipGraph, where HIP events cannot be recorded, no tracer attached.  Eight 64-bit words of device memory:
//   0 earliest workgroup entry of the launch in flight | 1 latest workgroup exit | 2 workgroups that have left
//   3 sum of the launches' durations | 4 launches | 5 shortest | 6 longest | 7 the last launch's      (ticks of the 100 MHz
//   s_memrealtime clock, the same on every XCD)
// Every access is a relaxed device-scope atomic (performed at the memory side, past the per-XCD L2s), ordered by their round
// trips -- a workgroup's exit stamp has RETURNED before its ticket goes out -- instead of release / acquire fences, which write
// back and invalidate whole L2s on eight XCDs.  The workgroup that draws the last ticket closes the launch: it adds exit - entry
// to the sums and re-arms words 0..2.  Cost with a timer: two atomics per workgroup; without one (null pointer): a scalar test.
constexpr unsigned long long kInsituIdle = ~0ull;
__device__ __forceinline__ unsigned long long realtime_ticks() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void insitu_enter(unsigned long long* t) {
    if (t != nullptr && threadIdx.x == 0) atomicMin(&t[0], realtime_ticks());
}
// call once per workgroup, from all its threads, after the workgroup's last global store was issued
__device__ __forceinline__ void insitu_exit(unsigned long long* t, const unsigned n_workgroups) {
    if (t == nullptr) return;                  // (wave-uniform: a kernel argument)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have landed
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned long long now = realtime_ticks();
    const unsigned long long seen = atomicMax(&t[1], now);
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(seen) : "memory");      // the stamp is in before the ticket goes out
    if (atomicAdd(&t[2], 1ull) + 1ull != (unsigned long long)n_workgroups) return;
    // every other workgroup's ticket -- hence its stamps -- has been performed
    const unsigned long long t0 = atomicMin(&t[0], kInsituIdle), t1 = atomicMax(&t[1], 0ull);
    const unsigned long long dur = t1 - t0;
    atomicAdd(&t[3], dur);
    atomicAdd(&t[4], 1ull);
    atomicMin(&t[5], dur);
    atomicMax(&t[6], dur);
    atomicExch(&t[7], dur);
    atomicExch(&t[0], kInsituIdle);
    atomicExch(&t[1], 0ull);
    atomicExch(&t[2], 0ull);
}

// Sum over the 64 lanes of a wave; the total is valid in lanes 48..63 (read lane 63).
__device__ __forceinline__ float wave_sum63(float v) {
    v += dpp_f<0xb1>(v);         // quad_perm [1,0,3,2]
    v += dpp_f<0x4e>(v);         // quad_perm [2,3,0,1]
    v += dpp_f<0x124>(v);        // row_ror 4
    v += dpp_f<0x128>(v);        // row_ror 8
    v += dpp_f<0x142, 0xa>(v);   // row_bcast 15 -> rows 1,3
    v += dpp_f<0x143, 0xc>(v);   // row_bcast 31 -> rows 2,3
    return v;
}
__device__ __forceinline__ int wave_sum63(int v) {
    v += dpp_i<0xb1>(v);
    v += dpp_i<0x4e>(v);
    v += dpp_i<0x124>(v);
    v += dpp_i<0x128>(v);
    v += dpp_i<0x142, 0xa>(v);
    v += dpp_i<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ float lane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int lane63(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ float wave_total(float v) { return lane63(wave_sum63(v)); }

// One fp8(e4m3) byte of a packed word -> f32.  Codes used: 0x38 = +1, 0xB8 = -1, 0x00 = 0.
template <int SEL>
__device__ __forceinline__ float code_to_f32(uint32_t word) {
    return __builtin_amdgcn_cvt_f32_fp8((int)word, SEL);
}

// 4 responses (fp32 0.0/1.0) + 4 mask bytes (0/1) -> 4 fp8 codes; counts packed n1<<16 | nobs.
// NEG: the sign of the codes is flipped (0x38 = +1 for a wrong answer, 0xB8 = -1 for a right one): the matrix row-split
// kernel multiplies the logit by -w
template <bool NEG = false>
__device__ __forceinline__ uint32_t pack_codes4(const float4 x, const uint32_t m, int& packed) {
    const uint32_t x0 = __builtin_bit_cast(uint32_t, x.x), x1 = __builtin_bit_cast(uint32_t, x.y);
    const uint32_t x2 = __builtin_bit_cast(uint32_t, x.z), x3 = __builtin_bit_cast(uint32_t, x.w);
    // byte 3 of 1.0f is 0x3F, of 0.0f is 0x00: bit 24 tells "correct"
    const uint32_t hi = __builtin_amdgcn_perm(x1, x0, 0x0c0c0703u) | __builtin_amdgcn_perm(x3, x2, 0x07030c0cu);
    const uint32_t xb = hi & 0x01010101u;
    const uint32_t code = ((NEG ? 0x38383838u : 0xB8B8B8B8u) ^ (xb << 7)) & ((m << 8) - m);   // m * 0xFF without the slow v_mul_lo_u32
    packed += __builtin_popcount(m) + (__builtin_popcount(xb & m) << 16);
    return code;
}

// Format P: 4 one-byte cell codes (0 = answered wrong, 1 = answered right, 2 = missing; VIBO_MASK_CODES) -> the same
// fp8 codes / packed counts.  `keep` = 0xFF for the bytes that belong to the row (padding past the row's end is dropped).
template <bool NEG = false>
__device__ __forceinline__ uint32_t pack_cell_codes4(const uint32_t w, const uint32_t keep, int& packed) {
    const uint32_t xb = w & 0x01010101u;
    const uint32_t m = (((w >> 1) & 0x01010101u) ^ 0x01010101u) & keep;
    const uint32_t code = ((NEG ? 0x38383838u : 0xB8B8B8B8u) ^ (xb << 7)) & ((m << 8) - m);
    packed += __builtin_popcount(m) + (__builtin_popcount(xb & m) << 16);
    return code;
}
// The same two packs for the matrix row-split kernel, as byte look-ups (v_perm_b32) with the counts in two plain
// accumulators: no `m * 0xFF` -- hipcc turns the shift-and-subtract form above back into a quarter-rate v_mul_lo_u32.
//   fp32 rows: selector byte = 2 [correct] + [observed] in {0..3} -> {0, code(wrong), 0, code(right)}
// `lut`: bytes {missing, wrong, -, right} as fp8 (e4m3) values, e.g. 0xB8003800 = {0, +1, 0, -1}
__device__ __forceinline__ uint32_t pack_codes4_lut(const float4 x, const uint32_t m, const uint32_t lut, int& nobs, int& n1) {
    const uint32_t x0 = __builtin_bit_cast(uint32_t, x.x), x1 = __builtin_bit_cast(uint32_t, x.y);
    const uint32_t x2 = __builtin_bit_cast(uint32_t, x.z), x3 = __builtin_bit_cast(uint32_t, x.w);
    // (the mask bytes are 0 / 1, so the "& m" also isolates bit 0 of the high bytes)
    const uint32_t xm = (__builtin_amdgcn_perm(x1, x0, 0x0c0c0703u) | __builtin_amdgcn_perm(x3, x2, 0x07030c0cu)) & m;
    const uint32_t sel = (xm << 1) + m;          // 0 missing | 1 wrong | 3 right
    nobs += __builtin_popcount(m);
    n1 += __builtin_popcount(xm);
    return __builtin_amdgcn_perm(0u, lut, sel);
}
// (the same, also handing out the [observed and right] bytes: the matrix kernel's fused conditional mode forms its indicators from them)
__device__ __forceinline__ uint32_t pack_codes4_lut_x(const float4 x, const uint32_t m, const uint32_t lut, int& nobs, int& n1, uint32_t& xm_out) {
    const uint32_t x0 = __builtin_bit_cast(uint32_t, x.x), x1 = __builtin_bit_cast(uint32_t, x.y);
    const uint32_t x2 = __builtin_bit_cast(uint32_t, x.z), x3 = __builtin_bit_cast(uint32_t, x.w);
    const uint32_t xm = (__builtin_amdgcn_perm(x1, x0, 0x0c0c0703u) | __builtin_amdgcn_perm(x3, x2, 0x07030c0cu)) & m;
    const uint32_t sel = (xm << 1) + m;          // 0 missing | 1 wrong | 3 right
    nobs += __builtin_popcount(m);
    n1 += __builtin_popcount(xm);
    xm_out = xm;
    return __builtin_amdgcn_perm(0u, lut, sel);
}
//   cell codes: selector byte = the code itself (0 wrong / 1 right / 2 missing), bytes outside `keep` forced to 2
__device__ __forceinline__ uint32_t pack_cell_codes4_lut(const uint32_t w, const uint32_t keep, const uint32_t lut, int& nobs, int& n1) {
    const uint32_t sel = (w & keep) | (0x02020202u & ~keep);
    const uint32_t ob = ~(sel >> 1) & 0x01010101u;
    nobs += __builtin_popcount(ob);
    n1 += __builtin_popcount(sel & ob);
    return __builtin_amdgcn_perm(0u, lut, sel);
}
constexpr uint32_t kAllMissing4 = 0x02020202u;

// 4 responses (fp32 0.0/1.0) + 4 mask bytes (0/1) -> the 4 Format P cell codes (0 wrong / 1 right / 2 missing): what a first pass
// over fp32 rows leaves behind for the passes that follow it (1 B instead of 5 B per cell)
__device__ __forceinline__ uint32_t cell_codes4(const float4 x, const uint32_t m) {
    const uint32_t x0 = __builtin_bit_cast(uint32_t, x.x), x1 = __builtin_bit_cast(uint32_t, x.y);
    const uint32_t x2 = __builtin_bit_cast(uint32_t, x.z), x3 = __builtin_bit_cast(uint32_t, x.w);
    const uint32_t hi = __builtin_amdgcn_perm(x1, x0, 0x0c0c0703u) | __builtin_amdgcn_perm(x3, x2, 0x07030c0cu);
    return (hi & m & 0x01010101u) | ((m ^ 0x01010101u) << 1);
}

// Four (a, b) float pairs -> their f16 hi pieces (round toward zero, one v_cvt_pkrtz each -- by the caller) and lo pieces
// lo = f16(x - hi) by two mixed-precision fmas per pair that write the two halves of one register.
// ONE asm block that ends in `s_nop 1`: the compiler's hazard recognizer does not see what inline asm writes, and a VALU result
// needs two wait states before a v_mfma reads it (plus the destination-select forwarding state of the half-register writes).
// Rounds 2-4 had the two instructions as separate asm statements: wherever the scheduler placed the second one right in front
// of the MFMA that reads the pair, the MFMA took the register's OLD upper half -- seen in round 5 as d LL/d a of the odd item
// tiles off by 10x, not reproducible run to run, in the one instantiation (3PL + flows + gathered fp32 rows) whose schedule
// happened to do that.
__device__ __forceinline__ void lo_pieces4(const uint32_t (&hi)[4], const float (&x)[8], uint32_t (&lo)[4]) {
    asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]),
          "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
}

}  // namespace vibo
