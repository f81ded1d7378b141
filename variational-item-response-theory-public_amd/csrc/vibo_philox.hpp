// vibo_philox.hpp -- counter-based N(0,1) noise shared by the trainer kernels (vibo_trainer.hip, vibo_ctrainer.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vibo_device.hpp"

namespace vibo {

// ---------------------------------------------------------------------------
// N(0,1) fill: Philox4x32-10 (Salmon et al. 2011) + Box-Muller, 4 normals per counter
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], const uint32_t k0, const uint32_t k1) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
}

// the 4 normals of counter group g (outputs 4g .. 4g+3 of the stream)
__device__ __forceinline__ float4 philox_normal4(const long long g, const uint32_t step, const uint32_t stream_id, const uint32_t seed_lo,
                                                 const uint32_t seed_hi) {
    uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), step, stream_id};
    uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    // uniforms in (0, 1]; v_sin / v_cos take their argument in revolutions
    const float u0 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * kLn2 * fast_log2(u0)), r1 = sqrtf(-2.0f * kLn2 * fast_log2(u2));
    return float4{r0 * __builtin_amdgcn_cosf(u1), r0 * __builtin_amdgcn_sinf(u1), r1 * __builtin_amdgcn_cosf(u3),
                  r1 * __builtin_amdgcn_sinf(u3)};
}
__device__ __forceinline__ void store_normal4(float* __restrict__ out, const long long n, const long long g, const float4 z) {
    if (4 * g + 3 < n && (((uintptr_t)out & 15) == 0)) {
        reinterpret_cast<float4*>(out)[g] = z;
    } else {
        const float zz[4] = {z.x, z.y, z.z, z.w};
        for (int k = 0; k < 4; ++k)
            if (4 * g + k < n) out[4 * g + k] = zz[k];
    }
}

// entry `idx` of stream `stream_id` (its group of 4 is recomputed by the 4 threads that share it: O(I) work)
__device__ __forceinline__ float philox_normal1(const long long idx, const uint32_t step, const uint32_t stream_id, const uint32_t seed_lo,
                                                const uint32_t seed_hi) {
    const float4 z = philox_normal4(idx >> 2, step, stream_id, seed_lo, seed_hi);
    return (idx & 3) == 0 ? z.x : (idx & 3) == 1 ? z.y : (idx & 3) == 2 ? z.z : z.w;
}

}  // namespace vibo
