// Micro-benchmarks on MI355X: cycles per wave-instruction for the ops the ELBO kernel uses, and
// whether v_mfma_f32_16x16x4_f32 overlaps with VALU work.   hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

#define ITERS 4096
#define UNROLL 8

template <int OP>
__global__ __launch_bounds__(256) void k_valu(float* out, float seed) {
    float a[UNROLL];
    float2v p[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = float2v{a[i], a[i] + 1.f}; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) a[i] = fmaf(a[i], 1.0001f, 0.5f);
            else if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i] * 1e-3f);
            else if (OP == 2) a[i] = __builtin_amdgcn_logf(a[i] + 2.f);
            else if (OP == 3) a[i] = __builtin_amdgcn_rcpf(a[i] + 2.f);
            else if (OP == 4) p[i] = p[i] * float2v{1.0001f, 0.9999f} + float2v{0.5f, 0.25f};
            else if (OP == 5) a[i] = (a[i] > 1.5f) ? a[i] * 0.5f : a[i] + 1.f;      // cmp + cndmask-ish
            else if (OP == 6) a[i] = __builtin_amdgcn_fmed3f(a[i] * 1.01f, -3.f, 3.f);
            else if (OP == 7) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x128, 0xf, 0xf, false));
            else if (OP == 8) a[i] = __builtin_amdgcn_cvt_f32_fp8(__builtin_bit_cast(int, a[i]) | 0x38, 0) + a[i];
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE 0: MFMA only (4 independent accumulators); 1: VALU only; 2: both in the same wave (interleaved);
// 3: even waves MFMA, odd waves VALU
template <int MODE>
__global__ __launch_bounds__(256) void k_mix(float* out, float seed) {
    float4v acc[4];
    float a[8];
    for (int i = 0; i < 4; ++i) acc[i] = float4v{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    const float x = seed + threadIdx.x, y = seed * 2 + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < ITERS; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], 1.0001f, 0.5f);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef short short8 __attribute__((ext_vector_type(8)));
// bf16 MFMA (16x16x32) vs VALU: MODE 0 mfma only, 1 valu only, 2 same wave, 3 even/odd waves
template <int MODE>
__global__ __launch_bounds__(256) void k_mix_bf16(float* out, float seed) {
    float4v acc[4];
    float a[8];
    for (int i = 0; i < 4; ++i) acc[i] = float4v{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    short8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + threadIdx.x + i); y[i] = (short)(0x3f00 + i); }
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < ITERS; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], 1.0001f, 0.5f);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* out; hipMalloc(&out, 256 * 16 * 256 * 4 * sizeof(float));
    const int blocks = 256 * 4;   // 4 blocks of 256 threads per CU = 16 waves/CU = 4 per SIMD
    const double waves_per_simd = 4.0;
    const char* names[] = {"v_fma_f32", "v_exp_f32(+mul)", "v_log_f32(+add)", "v_rcp_f32(+add)", "v_pk_fma_f32", "cmp+cndmask+mul/add", "mul+med3", "dpp mov+add", "or+cvt_fp8+add"};
    const int ninst[] = {1, 2, 2, 2, 1, 3, 2, 2, 3};
#define RUNV(OP) { float ms = time_ms([&]{ hipLaunchKernelGGL(k_valu<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }); \
    double insts = (double)ITERS * UNROLL * waves_per_simd; \
    printf("%-22s %8.3f ms  -> %6.2f ns per wave-op-group on one SIMD (%d instr)  = %5.2f cyc @2.4GHz\n", names[OP], ms, ms * 1e6 / insts, ninst[OP], ms * 1e6 / insts * 2.4); }
    RUNV(0) RUNV(1) RUNV(2) RUNV(3) RUNV(4) RUNV(5) RUNV(6) RUNV(7) RUNV(8)
    const char* mn[] = {"MFMA only (16/iter/wave)", "VALU only (32 fma/iter/wave)", "MFMA+VALU same wave", "even waves MFMA, odd VALU"};
#define RUNM(M) { float ms = time_ms([&]{ hipLaunchKernelGGL(k_mix<M>, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }); printf("%-30s %8.3f ms\n", mn[M], ms); }
    RUNM(0) RUNM(1) RUNM(2) RUNM(3)
    const char* bn[] = {"bf16 MFMA only (4/iter/wave)", "VALU only (32 fma/iter/wave)", "bf16 MFMA+VALU same wave", "even waves bf16 MFMA, odd VALU"};
#define RUNB(M) { float ms = time_ms([&]{ hipLaunchKernelGGL(k_mix_bf16<M>, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }); printf("%-34s %8.3f ms\n", bn[M], ms); }
    RUNB(0) RUNB(1) RUNB(2) RUNB(3)
    return 0;
}
