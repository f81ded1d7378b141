// vibo_launch.hpp -- per-ability-width launchers of the fused ELBO kernel.
// One translation unit per template ability width keeps hipcc compile times
// parallel (see Makefile); vibo_capi.hip dispatches to these.
#pragma once
#include <hip/hip_runtime.h>
#include "vibo_params.hpp"

namespace vibo {

// geometry: 0 = 8 waves x 2 item slots (I <= 1024), 1 = 8 waves x 4 slots (I <= 2048),
//           2 = 2 waves x 4 slots (I <= 512)
struct LaunchGeom {
    int geo, waves, grid;
    size_t lds_bytes;
};

hipError_t launch_elbo_a1(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a2(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a4(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a8(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);

}  // namespace vibo
