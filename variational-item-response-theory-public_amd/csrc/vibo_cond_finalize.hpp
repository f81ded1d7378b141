// vibo_cond_finalize.hpp -- the last stage of the conditional posterior's backward (records of the table-gradient pass ->
// grad_table[2 heads][2][I][2A]) as device routines, so that it can ride in the ELBO finalize launch (finalize_kernel's
// workgroups past its own outputs run one of these bodies: one launch less per train step) as well as in a launch of its own.
// Both bodies want 1024 threads and 8 KB of LDS.
#pragma once
#include <hip/hip_runtime.h>
#include "vibo_params.hpp"
#include "vibo_device.hpp"

namespace vibo {

// VALU pass (cond_post_kernel's records): grad_table[head][code][i][j2] = sum over the panel's workgroup records (fp64, fixed order)
// workgroup (bx, by): items 64 bx .. + 63 of output column by = (head * 2 + code) * 2A + j2
__device__ __forceinline__ void cond_finalize_body(const CondFinTail& t, const int bx, const int by, void* smem) {
    double (*part)[64] = reinterpret_cast<double (*)[64]>(smem);          // [16][64]
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int hcj = by;
    const int i = bx * 64 + lane;
    double acc = 0.0;
    if (i < t.I) {
        const int pn = i >> 10, local = i & 1023;
        const float* src = t.rec + (size_t)pn * t.bpp * t.rec_stride + (size_t)hcj * 1024 + local;
        for (int b = slice; b < t.bpp; b += 16) acc += (double)src[(size_t)b * t.rec_stride];
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && i < t.I) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[k][lane];
        const int hc = hcj / (2 * t.A), j2 = hcj % (2 * t.A);
        t.grad_table[((size_t)hc * t.I + i) * 2 * t.A + j2] = (float)s;
    }
}

// matrix-pipe pass (cm_backward_kernel's records): S1, S2 per (head, code, item, dim) -> grad_table[head][code][I][mu dims | logvar dims]
//   d/d mu = S1 tau,  d/d logvar = -(S1 mu + S2) tau^2 exp(logvar)       (the chain through utils.py:105-113)
// One workgroup per (stripe S = bx, quarter g' = by of its 128 record columns = 16 items x 2 codes): the records' 32 x N block is
// summed over the person ranges with whole-row loads (fixed order), the chain rule runs on the sums in LDS.
__device__ __forceinline__ void cm_cond_finalize_body(const CondFinTail& t, const int bx, const int by, void* smem) {
    float* part = reinterpret_cast<float*>(smem);          // [1024]
    float* sum = part + 1024;                              // [32 * 32]
    const int S = bx, gq = by, tid = threadIdx.x;
    const int I = t.I, A = t.A, nR = t.nR, N = t.N;
    const int E = 32 * N, nsl = 1024 / E;            // E = 512 or 1024 record values of this quarter; nsl person-range slices
    {
        const int e = tid % E, sl = tid / E;
        const float* q = t.rec + ((size_t)S * nR * 128 + 32 * gq) * N + e;
        const size_t rs = (size_t)128 * N;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = sl;
        for (; r + 7 * nsl < nR; r += 8 * nsl) {         // 8 loads in flight per thread
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += q[(size_t)(r + u * nsl) * rs];
        }
        for (; r < nR; r += nsl) a[0] += q[(size_t)r * rs];
        part[tid] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();
    if (tid < E) {
        float v = part[tid];
        for (int sl = 1; sl < nsl; ++sl) v += part[sl * E + tid];
        sum[tid] = v;
    }
    __syncthreads();
    for (int e = tid; e < 32 * A; e += 1024) {
        const int a = e % A, col = e / A;                 // col = 8 j + kk: item 64 S + 16 g' + 4 j + (kk >> 1), code kk & 1
        const int item = 64 * S + 16 * gq + 4 * (col >> 3) + ((col & 7) >> 1), c = col & 1;
        if (item >= I) continue;
        const float* sp = sum + col * N;
        const int pc = t.packed_cols;
        auto val = [&](const int k) { return pc > 0 ? (sp[k] + sp[pc + k]) + sp[2 * pc + k] : sp[k]; };
        const float* te = t.table + ((size_t)c * I + item) * 2 * A;
        const float es = expf(te[A + a]), tau = 1.0f / (es + kPoeEps), mu = te[a];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float s1 = val((2 * h) * A + a), s2 = val((2 * h + 1) * A + a);
            float* go = t.grad_table + ((size_t)(h * 2 + c) * I + item) * 2 * A;
            go[a] = s1 * tau;
            go[A + a] = -(s1 * mu + s2) * tau * tau * es;
        }
    }
}

__device__ __forceinline__ void cond_fin_tail_body(const CondFinTail& t, const int block, void* smem) {
    const int bx = block % t.gx, by = block / t.gx;
    if (t.kind == 1) cond_finalize_body(t, bx, by, smem);
    else cm_cond_finalize_body(t, bx, by, smem);
}

}  // namespace vibo
