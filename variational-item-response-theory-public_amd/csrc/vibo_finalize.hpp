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
    // sixteen records per trip, every load issued before the first add (the adds keep the order of the four-at-a-time loop
    // below: chain u takes records u, 4 + u, 8 + u, 12 + u of the trip) -- with one load round trip per four records the 256
    // records of a full launch cost four dependent memory latencies per output
    for (; b + 15 * SLICES < b1; b += 16 * SLICES) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = partial[(size_t)(b + u * SLICES) * stride + src];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] += (double)x[4 * i + u];
    }
    for (; b + 3 * SLICES < b1; b += 4 * SLICES) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] += (double)partial[(size_t)(b + u * SLICES) * stride + src];
    }
    for (int u = 0; b < b1; b += SLICES, ++u) a4[u] += (double)partial[(size_t)b * stride + src];
    return (a4[0] + a4[1]) + (a4[2] + a4[3]);
}

}  // namespace vibo
