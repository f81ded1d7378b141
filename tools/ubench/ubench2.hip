// Second micro-benchmark round on MI355X: issue cost (ns per wave-instruction per SIMD, 4 waves/SIMD, 8
// independent chains) of the cross-lane / scalar-operand instructions the row-split kernel uses.
//   hipcc --offload-arch=gfx950 -O3 ubench2.hip -o ubench2 && ./ubench2
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITERS 2048
#define UN 8

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed, int sl) {
    float a[UN], b[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = a[i] * 0.5f; }
    float s0 = seed * 1.0001f, s1 = seed * 0.9999f;
    s0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s0)));
    s1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s1)));
    unsigned long long msk = sl ? 0x5555555555555555ull : 0x3333333333333333ull;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(b[i]));
            else if (OP == 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            else if (OP == 3) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            else if (OP == 4) { int s; asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(a[i]), "s"(sl + i)); asm volatile("" :: "s"(s)); }
            else if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(msk));
            else if (OP == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(s0));
            else if (OP == 7) asm volatile("v_cvt_f32_fp8_sdwa %0, %1 src0_sel:BYTE_2" : "=v"(a[i]) : "v"(b[i]));
            else if (OP == 8) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            else if (OP == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 11) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            else if (OP == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b[i]) : "vcc");
            else if (OP == 13) asm volatile("v_max3_f32 %0, |%0|, |%1|, |%1|" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 14) asm volatile("s_nop 0");
            else if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            else if (OP == 16) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < UN; ++i) s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (threadIdx.x == 0 && blockIdx.x == 0 ? (float)(t1 - t0) * 1e-30f : 0.f);
}

// packed fma with an SGPR-pair operand (what hipcc emits for theta in SGPRs)
template <int OP>
__global__ __launch_bounds__(256) void kp(float* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[UN], b[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) { a[i] = f2{seed + i + threadIdx.x * 1e-3f, seed}; b[i] = a[i] * 0.5f; }
    f2 sv = f2{seed * 1.0001f, seed * 1.0001f};
    unsigned long long sbits = __builtin_bit_cast(unsigned long long, sv);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sbits), hi = __builtin_amdgcn_readfirstlane((unsigned)(sbits >> 32));
    unsigned long long sp = ((unsigned long long)hi << 32) | lo;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "s"(sp));
            else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b[i]), "s"(sp));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UN; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* out; hipMalloc(&out, (1 << 22) + 64);
    const int blocks = 256 * 4;
    const char* names[] = {"v_fma_f32 vvv", "v_fma_f32 v,s,v", "v_permlane32_swap", "v_permlane16_swap", "v_readlane (sgpr lane)", "v_cndmask e64 (sgpr mask)",
                           "v_med3 (sgpr op)", "v_cvt_f32_fp8 sdwa", "v_add_f32 dpp row_ror", "v_add_f32", "v_mul_f32", "v_exp_f32", "v_cmp_lt_f32 -> vcc",
                           "v_max3_f32 abs", "s_nop 0", "v_mov_b32", "v_fmac_f32"};
#define RUN(OP) { float ms = time_ms([&]{ hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 3); }); \
    printf("%-28s %7.3f ns per wave-instruction per SIMD\n", names[OP], ms * 1e6 / ((double)ITERS * UN * 4)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16)
    const char* pn[] = {"v_pk_fma_f32 v,s[2],v", "v_pk_fma_f32 vvv", "v_pk_mul_f32", "v_pk_fma_f32 v,s[2] op_sel_hi"};
#define RUNP(OP) { float ms = time_ms([&]{ hipLaunchKernelGGL(kp<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }); \
    printf("%-28s %7.3f ns per wave-instruction per SIMD\n", pn[OP], ms * 1e6 / ((double)ITERS * UN * 4)); }
    RUNP(0) RUNP(1) RUNP(2) RUNP(3)
    return 0;
}
