#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
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
__global__ void k(const float* in, float* o) {
    float v = in[threadIdx.x] * 1.37f + 0.11f;
    float a = v; a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    float b = xor32_add(xor16_add(v));
    o[threadIdx.x] = a; o[64 + threadIdx.x] = b;
}
int main() {
    float h[64], *d, *o, ho[128];
    for (int i = 0; i < 64; ++i) h[i] = (float)(i * i % 37) + 0.3f * i;
    hipMalloc(&d, 256); hipMalloc(&o, 128 * 4); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); hipMemcpy(ho, o, 128 * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) bad += memcmp(&ho[i], &ho[64 + i], 4) != 0;
    printf("mismatches %d  first %g %g\n", bad, ho[0], ho[64]);
}
