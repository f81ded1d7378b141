// Third micro-benchmark: how much instruction-level parallelism a gfx950 SIMD needs.  ns per wave-instruction
// per SIMD for CH independent dependency chains per wave and W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
template <int OP, int CH>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[CH];
    f2 b = f2{seed * 0.5f, seed * 0.25f};
#pragma unroll
    for (int i = 0; i < CH; ++i) a[i] = f2{seed + i + threadIdx.x * 1e-3f, seed};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8 / CH; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));
                else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i].x));
                else if (OP == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += a[i].x + a[i].y;
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
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_mul_f32"};
#define RUN(OP, CH, W) { float ms = time_ms([&]{ hipLaunchKernelGGL((k<OP, CH>), dim3(256 * W), dim3(256), 0, 0, out, 1.0f); }); \
    printf("%-14s chains=%d waves/SIMD=%d : %6.3f ns per wave-instr per SIMD (%6.3f ns per instr within a wave)\n", names[OP], CH, W, ms * 1e6 / ((double)ITERS * 8 * W), ms * 1e6 / ((double)ITERS * 8)); }
#define RUNW(OP, CH) RUN(OP, CH, 1) RUN(OP, CH, 2) RUN(OP, CH, 3) RUN(OP, CH, 4)
    RUNW(0, 1) RUNW(0, 2) RUNW(0, 4) RUNW(0, 8)
    RUNW(1, 1) RUNW(1, 2) RUNW(1, 4)
    RUNW(2, 1) RUNW(2, 2) RUNW(2, 4)
    RUNW(3, 1) RUNW(3, 2)
    return 0;
}
