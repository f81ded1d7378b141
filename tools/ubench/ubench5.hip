// Fifth micro-benchmark / probe (gfx950):
//  (1) ds_read_b64_tr_b16 semantics: which LDS element does (lane, elem) receive for per-lane addresses;
//  (2) v_mfma_f32_16x16x32_f16: are f16 subnormal inputs honoured or flushed;
//  (3) v_cvt_pkrtz_f16_f32 + hi/lo split residual.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// every lane reads from byte address 8 * lane (64 lanes x 4 halfs = LDS elements 0..255 holding their own index)
__global__ void tr_probe(short* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int off;                                   // in halfs
    if (mode == 0) off = 4 * l;                // lane-linear
    else off = 16 * ((l & 15) >> 2) + 4 * (l & 3) + 64 * (l >> 4);   // [row = (l & 15) >> 2][piece = l & 3] of a [4][16] block per 16-lane group
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

__global__ void denorm_probe(float* out) {
    const int l = threadIdx.x;
    f16x8 a, b;
    // A[m][k]: subnormal 2^-20 in k = 0 for every row; B[k][n]: 1024 in k = 0
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    if ((l >> 4) == 0) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1024.f; }
    f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    out[l] = d[0];
}

__global__ void split_probe(const float* x, float* res) {
    const int l = threadIdx.x;
    const float v0 = x[2 * l], v1 = x[2 * l + 1];
    const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v0, v1));
    const float r0 = v0 - (float)h[0], r1 = v1 - (float)h[1];
    const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    res[2 * l] = r0 - (float)lo[0];
    res[2 * l + 1] = r1 - (float)lo[1];
}

int main() {
    short* dout; (void)hipMalloc(&dout, 256 * 2);
    short hout[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dout, mode);
        (void)hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
        printf("tr16_b64 probe mode %d: lane -> 4 element indices received\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  l%02d: %3d %3d %3d %3d", l, hout[4 * l], hout[4 * l + 1], hout[4 * l + 2], hout[4 * l + 3]);
            if ((l & 3) == 3) printf("\n");
        }
    }
    float* dd; (void)hipMalloc(&dd, 64 * 4);
    hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, dd);
    float hd[64]; (void)hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    printf("f16 MFMA subnormal probe: 2^-20 * 1024 = %g (expect 0.000976562 if subnormals are honoured, 0 if flushed)\n", hd[0]);
    float hx[128], hr[128];
    srand(3);
    for (int i = 0; i < 128; ++i) hx[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * (i < 64 ? 1.f : 1e-3f);
    float *dx, *dr; (void)hipMalloc(&dx, sizeof hx); (void)hipMalloc(&dr, sizeof hr);
    (void)hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_probe, dim3(1), dim3(64), 0, 0, dx, dr);
    (void)hipMemcpy(hr, dr, sizeof hr, hipMemcpyDeviceToHost);
    double w1 = 0, w2 = 0;
    for (int i = 0; i < 64; ++i) w1 = fmax(w1, fabs(hr[i] / hx[i]));
    for (int i = 64; i < 128; ++i) w2 = fmax(w2, fabs(hr[i] / hx[i]));
    printf("f16 hi/lo split (pkrtz): worst relative residual %.3g for |x| ~ 1, %.3g for |x| ~ 1e-3\n", w1, w2);
    return 0;
}
