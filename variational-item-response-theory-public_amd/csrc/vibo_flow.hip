// vibo_flow.hip -- a stack of planar flows on the rows of a small matrix, forward and backward
// (flows.py:21-41 PlanarFlow.forward, :58-66 NormalizingFlows.forward; callers models.py:342-348: the item-side stack on
// the [I, D] item sample every step, the ability-side stack on [B, A] for the MLP-decoder models).
//
//   z_{k+1} = z_k + uhat_k tanh(w_k . z_k + b_k),   ladj = sum_k log(|1 + (1 - tanh^2)(w_k . uhat_k)| + 1e-8)
//
// The module path runs this as ~35 tiny PyTorch kernels per flow and direction; here a thread owns a row and walks the whole
// stack (dim <= 18 values in registers), the tanh values are kept for the backward, and the parameter gradients leave as
// one partial record per workgroup in a fixed order (the caller sums them: bitwise reproducible).  uhat is formed from
// (u, w) by the caller (autograd on 2 dim numbers per flow); `packed` is [n_flows][2 dim + 1] = uhat | w | b.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"

namespace vibo {

constexpr int kFlowMaxDim = 18;        // item_feat_dim = ability_dim + 2 <= VIBO_MAX_ABILITY_DIM_WIDE + 2

__device__ __forceinline__ float flow_tanh(float a) {
    // tanh(x) = 1 - 2 / (1 + e^2x), |x| clamped so that e^2x stays finite (tanh(+-15) = +-1 in fp32)
    return 1.0f - 2.0f / (1.0f + expf(2.0f * fminf(fmaxf(a, -15.f), 15.f)));
}

__global__ __launch_bounds__(256) void flow_stack_fwd_kernel(const float* __restrict__ z, const float* __restrict__ packed,
                                                             float* __restrict__ z_out, float* __restrict__ ladj,
                                                             float* __restrict__ tanh_out, int N, int D, int K) {
    __shared__ float par[VIBO_MAX_FLOWS][2 * kFlowMaxDim + 2];      // uhat | w | b | w . uhat
    const int tid = threadIdx.x;
    for (int e = tid; e < K * (2 * D + 1); e += 256) par[e / (2 * D + 1)][e % (2 * D + 1)] = packed[e];
    __syncthreads();
    if (tid < K) {
        float c = 0.f;
        for (int d = 0; d < D; ++d) c = fmaf(par[tid][D + d], par[tid][d], c);
        par[tid][2 * D + 1] = c;
    }
    __syncthreads();
    const int row = blockIdx.x * 256 + tid;
    if (row >= N) return;
    float zz[kFlowMaxDim];
#pragma unroll
    for (int d = 0; d < kFlowMaxDim; ++d) zz[d] = d < D ? z[(size_t)row * D + d] : 0.f;
    float la = 0.f;
    for (int k = 0; k < K; ++k) {
        float a = par[k][2 * D];
#pragma unroll
        for (int d = 0; d < kFlowMaxDim; ++d)
            if (d < D) a = fmaf(par[k][D + d], zz[d], a);
        const float t = flow_tanh(a);
        tanh_out[(size_t)row * K + k] = t;
        la += logf(fabsf(1.0f + (1.0f - t * t) * par[k][2 * D + 1]) + 1e-8f);
#pragma unroll
        for (int d = 0; d < kFlowMaxDim; ++d)
            if (d < D) zz[d] = fmaf(par[k][d], t, zz[d]);
    }
#pragma unroll
    for (int d = 0; d < kFlowMaxDim; ++d)
        if (d < D) z_out[(size_t)row * D + d] = zz[d];
    ladj[row] = la;
}

// backward: g_z_out [N][D], g_ladj [N] -> g_z [N][D]; partials [gridDim.x][K][2 D + 1] = d/d uhat | d/d w | d/d b
__global__ __launch_bounds__(256) void flow_stack_bwd_kernel(const float* __restrict__ z_out, const float* __restrict__ packed,
                                                             const float* __restrict__ tanh_in, const float* __restrict__ g_zout,
                                                             const float* __restrict__ g_ladj, float* __restrict__ g_z,
                                                             float* __restrict__ partials, int N, int D, int K) {
    __shared__ float par[VIBO_MAX_FLOWS][2 * kFlowMaxDim + 2];
    __shared__ float red[4][2 * kFlowMaxDim + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < K * (2 * D + 1); e += 256) par[e / (2 * D + 1)][e % (2 * D + 1)] = packed[e];
    __syncthreads();
    if (tid < K) {
        float c = 0.f;
        for (int d = 0; d < D; ++d) c = fmaf(par[tid][D + d], par[tid][d], c);
        par[tid][2 * D + 1] = c;
    }
    __syncthreads();
    const int row = blockIdx.x * 256 + tid;
    const bool ok = row < N;
    float zz[kFlowMaxDim], gz[kFlowMaxDim];            // output of the flow being backpropagated, gradient w.r.t. it
#pragma unroll
    for (int d = 0; d < kFlowMaxDim; ++d) {
        zz[d] = (ok && d < D) ? z_out[(size_t)row * D + d] : 0.f;
        gz[d] = (ok && d < D) ? g_zout[(size_t)row * D + d] : 0.f;
    }
    const float gl = ok ? g_ladj[row] : 0.f;
    for (int k = K - 1; k >= 0; --k) {
        const float t = ok ? tanh_in[(size_t)row * K + k] : 0.f;
        const float c = par[k][2 * D + 1];
        const float omt = 1.0f - t * t;
        const float psi = 1.0f + omt * c;
        const float dl_dpsi = gl * ((psi >= 0.f) ? 1.0f : -1.0f) / (fabsf(psi) + 1e-8f);
        float g_t = dl_dpsi * (-2.0f * t * c);
#pragma unroll
        for (int d = 0; d < kFlowMaxDim; ++d)
            if (d < D) {
                zz[d] = fmaf(-par[k][d], t, zz[d]);             // the flow's input (to an ulp of the forward's value)
                g_t = fmaf(gz[d], par[k][d], g_t);
            }
        const float g_a = g_t * omt;
        const float g_c = dl_dpsi * omt;
        // parameter gradients of this row, then the block sum
        float pg[2 * kFlowMaxDim + 1];
#pragma unroll
        for (int d = 0; d < kFlowMaxDim; ++d) {
            pg[d] = d < D ? fmaf(g_c, par[k][D + d], gz[d] * t) : 0.f;                  // d/d uhat
            pg[kFlowMaxDim + d] = d < D ? fmaf(g_c, par[k][d], g_a * zz[d]) : 0.f;      // d/d w
        }
        pg[2 * kFlowMaxDim] = g_a;                                                      // d/d b
#pragma unroll
        for (int d = 0; d < kFlowMaxDim; ++d)
            if (d < D) gz[d] = fmaf(g_a, par[k][D + d], gz[d]);
#pragma unroll
        for (int e = 0; e < 2 * kFlowMaxDim + 1; ++e) {
            const int dd = e < kFlowMaxDim ? e : e - kFlowMaxDim;
            if (e < 2 * kFlowMaxDim && dd >= D) continue;
            const float tot = wave_total(pg[e]);
            if (lane == 0) red[wv][e] = tot;
        }
        __syncthreads();
        if (tid < 2 * D + 1) {
            const int e = tid < D ? tid : tid < 2 * D ? kFlowMaxDim + (tid - D) : 2 * kFlowMaxDim;
            partials[((size_t)blockIdx.x * K + k) * (2 * D + 1) + tid] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        }
        __syncthreads();
    }
    if (ok) {
#pragma unroll
        for (int d = 0; d < kFlowMaxDim; ++d)
            if (d < D) g_z[(size_t)row * D + d] = gz[d];
    }
}

}  // namespace vibo

using namespace vibo;

static int flow_check(int n_rows, int dim, int n_flows) {
    if (n_rows < 1) return -2;
    if (dim < 1 || dim > kFlowMaxDim) return -6;
    if (n_flows < 1 || n_flows > VIBO_MAX_FLOWS) return -6;
    return 0;
}

extern "C" int vibo_flow_stack_forward(int n_rows, int dim, int n_flows, const float* z, const float* packed, float* z_out,
                                       float* ladj, float* tanh_out, void* stream) {
    const int rc = flow_check(n_rows, dim, n_flows);
    if (rc) return rc;
    if (!z || !packed || !z_out || !ladj || !tanh_out) return -5;
    hipLaunchKernelGGL(flow_stack_fwd_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, z, packed, z_out, ladj,
                       tanh_out, n_rows, dim, n_flows);
    return (int)hipGetLastError();
}

extern "C" int vibo_flow_stack_backward(int n_rows, int dim, int n_flows, const float* z_out, const float* packed,
                                        const float* tanh_saved, const float* g_zout, const float* g_ladj, float* g_z,
                                        float* partials, void* stream) {
    const int rc = flow_check(n_rows, dim, n_flows);
    if (rc) return rc;
    if (!z_out || !packed || !tanh_saved || !g_zout || !g_ladj || !g_z || !partials) return -5;
    hipLaunchKernelGGL(flow_stack_bwd_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, z_out, packed,
                       tanh_saved, g_zout, g_ladj, g_z, partials, n_rows, dim, n_flows);
    return (int)hipGetLastError();
}
