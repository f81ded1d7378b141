// vibo_finalize.hpp -- the fixed-order sum over the per-workgroup partial records, shared by finalize_kernel
// (vibo_capi.hip) and by the train epilogue that folds the finalize into its own launch (vibo_trainer.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "vibo_params.hpp"

namespace vibo {

// Slice `slice` of SLICES sums records b0 + slice, b0 + slice + SLICES, ... (< b1) of float `src` in fp64: four independent
// chains keep four record loads in flight.  The caller adds the SLICES slice sums in slice order: bitwise reproducible.
template <int SLICES>
__device__ __forceinline__ double record_slice_sum(const float* __restrict__ partial, const size_t stride, const int src, const int b0,
                                                   const int b1, const int slice) {
    double a4[4] = {0.0, 0.0, 0.0, 0.0};
    int b = b0 + slice;
    for (; b + 3 * SLICES < b1; b += 4 * SLICES) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] += (double)partial[(size_t)(b + u * SLICES) * stride + src];
    }
    for (int u = 0; b < b1; b += SLICES, ++u) a4[u] += (double)partial[(size_t)b * stride + src];
    return (a4[0] + a4[1]) + (a4[2] + a4[3]);
}

}  // namespace vibo
