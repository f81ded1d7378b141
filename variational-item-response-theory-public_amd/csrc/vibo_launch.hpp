// vibo_launch.hpp -- per-ability-width launchers of the fused ELBO kernel.
// One translation unit per template ability width keeps hipcc compile times
// parallel (see Makefile); vibo_capi.hip dispatches to these.
#pragma once
#include <hip/hip_runtime.h>
#include "vibo_params.hpp"

namespace vibo {

// waves per workgroup: 2 (I <= 144), 4 (I <= 304), 8 (I <= 512), 16 (I <= 1024)
struct LaunchGeom {
    int waves, grid;
    size_t lds_bytes;
};

hipError_t launch_elbo_a1(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a2(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a4(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a8(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);

// wave-per-row kernel (vibo_row_kernel.hip): A in {1,2}, 1PL/2PL, I <= 1024, 16-byte aligned rows
hipError_t launch_elbo_rows(const ElboParams& p, int irt, bool grad, int grid, hipStream_t s);

// row-split kernel (vibo_split_kernel.hip): ability_dim 3..8 (at = 4 | 8), 1PL/2PL, I <= 1024, nq = ceil(I / 256)
hipError_t launch_elbo_split(const ElboParams& p, int at, int irt, bool grad, int nq, int grid, hipStream_t s);

}  // namespace vibo
