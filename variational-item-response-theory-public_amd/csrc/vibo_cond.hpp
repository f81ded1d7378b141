// vibo_cond.hpp -- launch interface of the conditional-posterior kernels (vibo_cond.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vibo_params.hpp"

namespace vibo {

struct CondParams {
    const float* response;
    const void* mask;
    const int64_t* row_index;
    const float* table;        // [2][I_total][2A]
    float* pre_out;            // cond_pre:  [B][2A+1] of this panel
    const float* coef_in;      // cond_post: [coef_panels][B][4A]
    float* partial;            // cond_post: [grid][rec_stride] records of this panel
    long long resp_stride, mask_stride;
    int B, I, I_total, item0, A, mask_dtype, coef_panels, rec_stride;
    uint8_t* codes_out;        // cond_pre on fp32 rows: if set, the rows' 1-byte cell codes go here ([B][codes_stride], minibatch order)
    long long codes_stride;
    int a0;                    // first ability dim of this launch (dims a0 .. a0 + AT - 1: more than 4 dims take two launches)
    int panel_count;           // cond_pre: > 1 = ALL 1024-item panels in this launch (workgroup blockIdx.x = slot * panel_count + panel;
                               // item0 / I / pre_out follow from it): ten launches of a 10 000-item row become one
};

hipError_t launch_cond_pre(const CondParams& p, int at, int nq, int grid, hipStream_t s);
hipError_t launch_cond_post(const CondParams& p, int at, int nq, int grid, hipStream_t s);
// defer != nullptr: nothing is launched, *defer describes the job for the ELBO finalize launch (FinalizeParams::tail)
hipError_t launch_cond_finalize(const float* partial, float* grad_table, int I, int A, int panels, int bpp, int rec_stride,
                                hipStream_t s, CondFinTail* defer = nullptr);

// the same two passes on the matrix pipe, all items in one launch, rows as 1-byte cell codes (vibo_cmean.hip)
size_t cond_mfma_scratch_bytes(long long B, int I, int A);
hipError_t launch_cond_pre_mfma(const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I, int A,
                                const float* table, float* pre, void* scratch, hipStream_t s);
hipError_t launch_cond_pre_mfma_fp32(const float* response, const void* mask, long long resp_stride, long long mask_stride,
                                     const int64_t* row_index, long long B, int I, int A, const float* table, float* pre,
                                     uint8_t* codes_out, long long codes_stride, void* scratch, hipStream_t s);
hipError_t launch_cond_post_mfma(const uint8_t* codes, long long stride, const int64_t* row_index, long long B, int I, int A,
                                 const float* table, const float* coef, float* grad_table, void* scratch, hipStream_t s,
                                 CondFinTail* defer = nullptr);

}  // namespace vibo
