// vibo_general.hpp -- launch interface of the general (wave-per-person) ELBO kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vibo {

struct GeneralParams {
    const float* response;
    const void* mask;
    const int64_t* row_index;
    const float* table;       // [2][2A] or [2][I][2A]
    const float* item;        // [I][D] (raw item sample)
    const float* eps;         // [B][A]
    const float* flow;        // [n_flows][2A+1] or null
    float* ability_mu;
    float* ability_logvar;
    float* ability;
    float* ability_k;         // nullable
    float* ability_ladj;      // nullable
    float* grad_table;        // zeroed by the launcher, accumulated with atomics
    float* grad_item;
    float* grad_flow;
    float* acc_scalars;       // [8] zeroed workspace
    float* out_scalars;
    long long resp_stride, mask_stride;
    long long B;
    int I, A, D, irt, conditional, missing_mode, mask_dtype, reg_mode, n_flows, want_grad, item_in_lds;
};

hipError_t launch_elbo_general(const GeneralParams& p, int num_cu, hipStream_t s);

}  // namespace vibo
