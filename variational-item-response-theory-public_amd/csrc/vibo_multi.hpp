// vibo_multi.hpp -- launch interface of the multi-sample forward kernel (vibo_multi.hip / vibo_multi_kernel.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include "vibo_params.hpp"

namespace vibo {

struct MultiParams {
    ElboParams e;
    long long item_sstride, eps_sstride;      // floats between the samples' prepped item tables / eps blocks
};
// at = template ability width (2 / 4 / 8), sc = samples per pass (1, 2, or 4 with at <= 4), nq = ceil(I / 256)
hipError_t launch_elbo_multi(const MultiParams& mp, int at, int irt, int sc, int nq, int grid, hipStream_t s);

}  // namespace vibo
