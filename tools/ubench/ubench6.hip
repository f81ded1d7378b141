// Sixth micro-benchmark: VALU issue at low occupancy (the matrix row-split kernel runs 2 waves per SIMD at ~240 VGPRs).
// ns per instruction per wave for W = 1, 2, 3, 4 waves per SIMD: plain v_fma_f32, v_pk_fma_f32 / v_pk_mul / v_pk_add (two
// lanes' worth of work per issue slot), v_exp_f32, v_fma_mix_f32, v_cvt_pkrtz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    f2 b = f2{seed * 0.5f, seed * 0.25f};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i + threadIdx.x * 1e-3f, seed};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));
                else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i].x));
                else if (OP == 5) asm volatile("v_fma_mix_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]" : "+v"(a[i].x) : "v"(b.x));
                else if (OP == 6) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                else if (OP == 7) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(b.y));
                else if (OP == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float time_ms(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 24);
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_fma_mix_f32", "v_cvt_pkrtz_f16_f32", "v_med3_f32", "v_add_f32"};
#define RUN(OP, W) { float ms = time_ms([&]{ hipLaunchKernelGGL((k<OP>), dim3(256 * W), dim3(256), 0, 0, out, 1.0f); }); \
    printf("%-22s waves/SIMD=%d : %6.3f ns per instr per wave, %6.3f ns per instr per SIMD (%5.2f cycles @2.4GHz)\n", names[OP], W, ms * 1e6 / ((double)ITERS * 32), ms * 1e6 / ((double)ITERS * 32 * W), ms * 1e6 / ((double)ITERS * 32 * W) * 2.4); }
#define RUNW(OP) RUN(OP, 1) RUN(OP, 2) RUN(OP, 3) RUN(OP, 4)
    RUNW(0) RUNW(1) RUNW(2) RUNW(3) RUNW(4) RUNW(5) RUNW(6) RUNW(7) RUNW(8)
    return 0;
}
