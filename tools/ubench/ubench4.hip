// Fourth micro-benchmark: does the bf16 matrix pipe run under a VALU stream on gfx950, and what are the operand
// layouts of v_mfma_f32_16x16x32_bf16?
//  (1) loop body = NM independent MFMAs + NV plain VALU ops (+ NT transcendentals), W waves per SIMD:
//      cycles per iteration -> whether MFMA time hides under VALU time (same wave and partner wave);
//  (2) layout probe: random A[16][32], B[32][16] in the hypothesised lane layout
//      A: lane l holds A[l & 15][8 (l >> 4) + 0..7], B: lane l holds B[8 (l >> 4) + 0..7][l & 15],
//      D: lane l holds D[4 (l >> 4) + r][l & 15], checked against a host product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#define ITERS 2048
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NT>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    float v[8], t[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = seed * 0.001f * (i + 1);
    const float c = seed * 0.5f;
    for (int it = 0; it < ITERS; ++it) {
        // interleave: one MFMA, then NV / NM VALU ops
#pragma unroll
        for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
            if (NM > 0) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NV / (NM > 0 ? NM : 1); ++u)
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u & 7]) : "v"(c));
#pragma unroll
            for (int u = 0; u < NT / (NM > 0 ? NM : 1); ++u)
                asm volatile("v_exp_f32 %0, %0" : "+v"(t[u & 3]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += t[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void layout_probe(const unsigned short* A, const unsigned short* B, float* D) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = (short)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = d[r];
}

__global__ void cvt_probe(const float* x, unsigned* out) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x[2 * threadIdx.x]), "v"(x[2 * threadIdx.x + 1]));
    out[threadIdx.x] = r;
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
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    float* out; (void)hipMalloc(&out, 1 << 24);
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk_khz);
#define RUN(NM, NV, NT, W) { float ms = time_ms([&]{ hipLaunchKernelGGL((k<NM, NV, NT>), dim3(256 * W), dim3(256), 0, 0, out, 1.0f); }); \
    printf("mfma=%d valu=%2d trans=%2d waves/SIMD=%d : %7.1f ns per iteration per wave-slot  (%6.1f ns per iteration per SIMD-wave-set)\n", NM, NV, NT, W, ms * 1e6 / ITERS, ms * 1e6 / ITERS / W); }
#define RUNW(NM, NV, NT) RUN(NM, NV, NT, 1) RUN(NM, NV, NT, 2)
    RUNW(4, 0, 0) RUNW(8, 0, 0)
    RUNW(0, 32, 0) RUNW(0, 64, 0) RUNW(0, 32, 8)
    RUNW(4, 32, 0) RUNW(4, 64, 0) RUNW(8, 64, 0) RUNW(4, 32, 8) RUNW(8, 64, 16) RUNW(8, 32, 0) RUNW(8, 16, 0)
    // layout probe
    unsigned short hA[16 * 32], hB[32 * 16];
    srand(7);
    for (int i = 0; i < 512; ++i) { hA[i] = (unsigned short)(0x3f00 + rand() % 256); hB[i] = (unsigned short)(0xbf00 + rand() % 300); }
    unsigned short *dA, *dB; float* dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, 256 * 4);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    float hD[256]; (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)bf2f(hA[i * 32 + kk]) * bf2f(hB[kk * 16 + j]);
        worst = fmax(worst, fabs(s - hD[i * 16 + j]));
    }
    printf("layout probe: max |D - A.B| = %g (%s)\n", worst, worst < 1e-3 ? "layout as hypothesised" : "LAYOUT MISMATCH");
    // cvt probe: rounding mode of v_cvt_pk_bf16_f32
    float hx[128]; for (int i = 0; i < 128; ++i) hx[i] = 1.0f + (float)i * (1.0f / 512.0f) + 1e-4f * i;
    float* dx; unsigned* dr; (void)hipMalloc(&dx, sizeof hx); (void)hipMalloc(&dr, 64 * 4);
    (void)hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dr);
    unsigned hr[64]; (void)hipMemcpy(hr, dr, sizeof hr, hipMemcpyDeviceToHost);
    int rne = 0, trunc = 0;
    for (int i = 0; i < 64; ++i) for (int h = 0; h < 2; ++h) {
        float x = hx[2 * i + h]; unsigned u; memcpy(&u, &x, 4);
        unsigned short got = (unsigned short)(hr[i] >> (16 * h));
        unsigned short t = (unsigned short)(u >> 16);
        unsigned short r = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
        rne += got == r; trunc += got == t;
    }
    printf("cvt_pk_bf16_f32: %d/128 match round-to-nearest-even, %d/128 match truncation (low half = first source)\n", rne, trunc);
    return 0;
}
