// vibo_cmean.hip -- --ability-merge mean WITH --conditional-posterior (models.py:664-710 with _forward_mean :631-650):
// the per-term feature depends on the item, h[c, i, :] = elu(mlp1([c, item_i])) in R^H, and a person's encoder input is the
// mean of h over its observed cells.  The sum over the cells is the one genuinely dense contraction of the VIBO encoder:
//
//     S[p, :]  = sum_i [cell (p, i) observed] h[code_pi, i, :]          = onehot(codes) [B, 2I] x h [2I, H]
//     dh[c, i, :] = sum_p [code_pi == c] G[p, :]                         = onehot(codes)^T [2I, B] x G [B, H]
//
// Both run on the matrix pipe (v_mfma_f32_16x16x32_f16) straight from the 1-byte cell codes (0 wrong / 1 right / 2 missing):
// the one-hot operand is exact in f16, the dense operand (h, or the upstream gradient G) goes in as hi + lo f16 pieces
// (round toward zero, |x - hi - lo| <= 2^-22 |x|) with fp32 accumulation -- fp32-grade, like the matrix ELBO kernel.
// Round 2 did this as two rocBLAS GEMMs on materialised fp32 indicator matrices (three extra passes over the rows).
//
// Operand layouts (16x16x32: A lane (row = lane & 15, g = lane >> 4) holds k = 8 g .. 8 g + 7; B lane (col = lane & 15, g)
// likewise; D lane (col = lane & 15, g) holds rows 4 g .. 4 g + 3):
//   * K order of the forward, per super-step S of 64 items: K-step j in 0..3, lane group g, kk in 0..7 <-> item
//     64 S + 16 g + 4 j + (kk >> 1), code kk & 1 -- so that a lane's four K-steps are the four dwords of ONE 16-byte load of
//     its person's code row (a wave reads 16 rows x 64 contiguous bytes per load);
//   * the dense operands are pre-arranged once per call into "images" whose 16-byte lane pieces are contiguous per wave load.
// H = 64 (the reference's --hidden-dim default); other widths stay on the GEMM path of the caller.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"

namespace vibo {

typedef _Float16 cm_half8 __attribute__((ext_vector_type(8)));
typedef float cm_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kCmH = 64;

__device__ __forceinline__ void cm_split(float x, _Float16& hi, _Float16& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x, 0.f));
    hi = a[0];
    lo = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x - (float)a[0], 0.f))[0];
}
// 4 cell codes (one dword) -> the 8 one-hot halfs [c0 == 0, c0 == 1, c1 == 0, c1 == 1, ...] (f16 1.0 = 0x3C00)
__device__ __forceinline__ cm_half8 cm_onehot(const uint32_t w) {
    const uint32_t e1 = w & 0x01010101u;                                 // code == 1
    const uint32_t e0 = ~(w | (w >> 1)) & 0x01010101u;                   // code == 0
    const uint32_t t0 = __builtin_amdgcn_perm(0u, 0x00003C00u, e0);      // bytes 0x3C where the cell is a 0
    const uint32_t t1 = __builtin_amdgcn_perm(0u, 0x00003C00u, e1);
    // cell b -> dword [0x00, t0.b, 0x00, t1.b]  (perm: S0 = t1 -> bytes 4..7, S1 = t0 -> bytes 0..3, 0x0c = constant 0)
    uint32_t d[4];
    d[0] = __builtin_amdgcn_perm(t1, t0, 0x040c000cu);
    d[1] = __builtin_amdgcn_perm(t1, t0, 0x050c010cu);
    d[2] = __builtin_amdgcn_perm(t1, t0, 0x060c020cu);
    d[3] = __builtin_amdgcn_perm(t1, t0, 0x070c030cu);
    return __builtin_bit_cast(cm_half8, uint4{d[0], d[1], d[2], d[3]});
}
__device__ __forceinline__ cm_f32x4 cm_mfma(const cm_half8 a, const cm_half8 b, const cm_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ---- images -----------------------------------------------------------------------------------------------------------
// forward B operand: img[(((S * 4 + j) * 4 + nt) * 2 + plane) * 64 + lane] = half8 over kk of h[kk & 1][item][16 nt + (lane & 15)]
__global__ __launch_bounds__(256) void cm_table_image_kernel(const float* __restrict__ h /* [2][I][64] */, cm_half8* __restrict__ img, int I, int nS) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nS * 4 * 4 * 64) return;
    const int lane = t & 63, nt = (t >> 6) & 3, j = (t >> 8) & 3, S = t >> 10;
    const int n = lane & 15, g = lane >> 4;
    cm_half8 hi, lo;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int item = 64 * S + 16 * g + 4 * j + (kk >> 1), c = kk & 1;
        const float v = item < I ? h[((size_t)c * I + item) * kCmH + 16 * nt + n] : 0.f;
        _Float16 a, b;
        cm_split(v, a, b);
        hi[kk] = a; lo[kk] = b;
    }
    const size_t o = ((size_t)((S * 4 + j) * 4 + nt) * 2) * 64 + lane;
    img[o] = hi;
    img[o + 64] = lo;
}
// backward B operand: gimg[((c32 * 4 + nt) * 2 + plane) * 64 + lane] = half8 over kk of G[32 c32 + 8 g + kk][16 nt + (lane & 15)]
__global__ __launch_bounds__(256) void cm_grad_image_kernel(const float* __restrict__ G /* [B][64] */, cm_half8* __restrict__ gimg, long long B, long long n32) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n32 * 4 * 64) return;
    const int lane = (int)(t & 63), nt = (int)((t >> 6) & 3);
    const long long c32 = t >> 8;
    const int n = lane & 15, g = lane >> 4;
    cm_half8 hi, lo;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const long long p = 32 * c32 + 8 * g + kk;
        const float v = p < B ? G[p * kCmH + 16 * nt + n] : 0.f;
        _Float16 a, b;
        cm_split(v, a, b);
        hi[kk] = a; lo[kk] = b;
    }
    const size_t o = ((size_t)(c32 * 4 + nt) * 2) * 64 + lane;
    gimg[o] = hi;
    gimg[o + 64] = lo;
}

// the 16 code bytes of person `p` at items [64 S + 16 g, +16): missing (2) beyond the matrix / the row's end
__device__ __forceinline__ uint4 cm_load_codes(const uint8_t* __restrict__ codes, long long stride, long long B, int I, long long p, int S, int g) {
    uint4 w = uint4{0x02020202u, 0x02020202u, 0x02020202u, 0x02020202u};
    const int i0 = 64 * S + 16 * g;
    if (p < B && i0 < I) {
        const uint8_t* rp = codes + p * stride + i0;
        if (i0 + 16 <= I) {
            w = *reinterpret_cast<const uint4*>(rp);                      // (rows are 4-byte aligned with a stride % 4 == 0: 16-byte
        } else {                                                          //  alignment holds when stride % 16 == 0, else dwords)
            uint32_t d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = 0x02020202u;
                const int left = I - (i0 + 4 * k);
                if (left >= 4) d[k] = *reinterpret_cast<const uint32_t*>(rp + 4 * k);
                else if (left > 0) {
                    uint32_t v = 0x02020202u;
                    for (int b = 0; b < left; ++b) v = (v & ~(0xffu << (8 * b))) | ((uint32_t)rp[4 * k + b] << (8 * b));
                    d[k] = v;
                }
            }
            w = uint4{d[0], d[1], d[2], d[3]};
        }
    }
    return w;
}
__device__ __forceinline__ uint4 cm_load_codes_any(const uint8_t* __restrict__ codes, long long stride, long long B, int I, long long p, int S, int g,
                                                   bool aligned16) {
    if (aligned16) return cm_load_codes(codes, stride, B, I, p, S, g);
    // rows only 4-byte aligned: four dword loads
    uint32_t d[4] = {0x02020202u, 0x02020202u, 0x02020202u, 0x02020202u};
    const int i0 = 64 * S + 16 * g;
    if (p < B) {
        const uint8_t* rp = codes + p * stride + i0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int left = I - (i0 + 4 * k);
            if (left >= 4) d[k] = *reinterpret_cast<const uint32_t*>(rp + 4 * k);
            else if (left > 0) {
                uint32_t v = 0x02020202u;
                for (int b = 0; b < left; ++b) v = (v & ~(0xffu << (8 * b))) | ((uint32_t)rp[4 * k + b] << (8 * b));
                d[k] = v;
            }
        }
    }
    return uint4{d[0], d[1], d[2], d[3]};
}

// ---- forward: S[p][:] = sum_i onehot . h ; a wave owns 64 persons (4 M-tiles) x all 64 hidden units ---------------------
__global__ __launch_bounds__(256) void cm_forward_kernel(const uint8_t* __restrict__ codes, long long stride, long long B, int I, int nS,
                                                         const cm_half8* __restrict__ img, float* __restrict__ Sout, int aligned16) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const long long p0 = ((long long)blockIdx.x * 4 + wv) * 64;
    if (p0 >= B) return;
    cm_f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = cm_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int S = 0; S < nS; ++S) {
        uint4 w[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) w[mt] = cm_load_codes_any(codes, stride, B, I, p0 + 16 * mt + m, S, g, aligned16 != 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cm_half8 a[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[mt] = cm_onehot(j == 0 ? w[mt].x : j == 1 ? w[mt].y : j == 2 ? w[mt].z : w[mt].w);
            const cm_half8* ip = img + ((size_t)(S * 4 + j) * 4 * 2) * 64 + lane;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const cm_half8 bh = ip[(size_t)(nt * 2) * 64], bl = ip[(size_t)(nt * 2 + 1) * 64];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[mt][nt] = cm_mfma(a[mt], bh, acc[mt][nt]);
                    acc[mt][nt] = cm_mfma(a[mt], bl, acc[mt][nt]);
                }
            }
        }
    }
    // D: lane (col n = m, g) holds rows 4 g + jj of the M-tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const long long p = p0 + 16 * mt + 4 * g + jj;
            if (p < B) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) Sout[p * kCmH + 16 * nt + m] = acc[mt][nt][jj];
            }
        }
}

// ---- backward: dh = onehot^T G ; a wave (= workgroup) owns a stripe of 64 items x a range of persons -------------------
constexpr int kCmTileStride = 136;      // halfs per person row of the one-hot tile (128 + 8: 16-byte aligned rows, staggered banks)
__global__ __launch_bounds__(64) void cm_backward_kernel(const uint8_t* __restrict__ codes, long long stride, long long B, int I,
                                                         const cm_half8* __restrict__ gimg, float* __restrict__ rec, int nR, long long per_r,
                                                         int aligned16) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[64 * kCmTileStride];
    const int lane = threadIdx.x;
    const int m = lane & 15, g = lane >> 4;
    const int S = blockIdx.x / nR, r = blockIdx.x % nR;
    const long long pa = (long long)r * per_r, pb = pa + per_r < B ? pa + per_r : B;        // per_r is a multiple of 64
    cm_f32x4 acc[8][4];
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[T][nt] = cm_f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long p0 = pa; p0 < pb; p0 += 64) {
        // one-hot tile [64 persons][128 (item, code) columns]: lane (m, g) of M-tile mt writes its 4 x 8 halfs at columns 32 g + 8 j
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const uint4 w = cm_load_codes_any(codes, stride, B, I, p0 + 16 * mt + m, S, g, aligned16 != 0);
            _Float16* row = tile + (16 * mt + m) * kCmTileStride + 32 * g;
            *reinterpret_cast<cm_half8*>(row) = cm_onehot(w.x);
            *reinterpret_cast<cm_half8*>(row + 8) = cm_onehot(w.y);
            *reinterpret_cast<cm_half8*>(row + 16) = cm_onehot(w.z);
            *reinterpret_cast<cm_half8*>(row + 24) = cm_onehot(w.w);
        }
        // the upstream gradient of these 64 persons as B operands: [2 K-chunks of 32 persons][4 N-tiles][hi | lo]
        cm_half8 bg[2][4][2];
        const long long c32 = p0 >> 5;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const cm_half8* gp = gimg + ((size_t)((c32 + c) * 4 + nt) * 2) * 64 + lane;
                bg[c][nt][0] = gp[0];
                bg[c][nt][1] = gp[64];
            }
        __syncthreads();
#pragma unroll
        for (int T = 0; T < 8; ++T) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // A operand: row = column 16 T + m of the tile, k = persons 32 c + 8 g + kk
                cm_half8 a;
                const _Float16* col = tile + (32 * c + 8 * g) * kCmTileStride + 16 * T + m;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) a[kk] = col[kk * kCmTileStride];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[T][nt] = cm_mfma(a, bg[c][nt][0], acc[T][nt]);
                    acc[T][nt] = cm_mfma(a, bg[c][nt][1], acc[T][nt]);
                }
            }
        }
        __syncthreads();
    }
    // record [128 columns = 32 g' + 8 j + kk][64 hidden]; D: lane (col n = m, g) holds rows 4 g + jj of tile T
    float* out = rec + (size_t)blockIdx.x * 128 * kCmH;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) out[(size_t)(16 * T + 4 * g + jj) * kCmH + 16 * nt + m] = acc[T][nt][jj];
}
// dh[c][item][n] = sum over the person ranges of the records, fixed order.  column (g', j, kk) <-> item 64 S + 16 g' + 4 j + (kk >> 1), code kk & 1
__global__ __launch_bounds__(256) void cm_backward_reduce_kernel(const float* __restrict__ rec, float* __restrict__ dh, int I, int nR) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * I * kCmH) return;
    const int n = t % kCmH, item = (t / kCmH) % I, c = t / (kCmH * I);
    const int S = item >> 6, il = item & 63;
    const int gq = il >> 4, j = (il >> 2) & 3, kk = ((il & 3) << 1) | c;
    const int colm = 32 * gq + 8 * j + kk;
    const float* rp = rec + ((size_t)S * nR * 128 + colm) * kCmH + n;
    float a = 0.f;
    for (int r = 0; r < nR; ++r) a += rp[(size_t)r * 128 * kCmH];
    dh[t] = a;
}

}  // namespace vibo

using namespace vibo;

static int cm_ranges(long long B, int nS, long long* per_r) {
    // ~2048 workgroups: nS item stripes x nR person ranges of a multiple of 64 persons
    int nR = 2048 / (nS > 0 ? nS : 1);
    if (nR < 1) nR = 1;
    long long per = (B + nR - 1) / nR;
    per = (per + 63) / 64 * 64;
    if (per < 64) per = 64;
    *per_r = per;
    return (int)((B + per - 1) / per);
}

extern "C" {

// bytes of scratch the two calls need (images + partial records), 256-byte aligned
size_t vibo_code_table_scratch_bytes(int64_t num_person, int num_item, int hidden_dim) {
    if (num_person < 1 || num_item < 1 || hidden_dim != kCmH) return 0;
    const int nS = (num_item + 63) / 64;
    long long per_r;
    const int nR = cm_ranges(num_person, nS, &per_r);
    const size_t timg = ((size_t)nS * 4 * 4 * 2 * 64 * 16 + 255) & ~(size_t)255;
    const size_t gimg = ((size_t)((num_person + 63) / 64 * 2) * 4 * 2 * 64 * 16 + 255) & ~(size_t)255;
    const size_t rec = (size_t)nS * nR * 128 * kCmH * 4;
    return timg + gimg + rec + 256;
}

/* S [B][64] = sum over the observed cells of feature[code][item][:]   (feature = elu(mlp1([c, item_i])), [2][I][64]) */
int vibo_code_table_sum_forward(int64_t num_person, int num_item, int hidden_dim, const uint8_t* codes, int64_t codes_row_stride,
                                const float* feature, float* out_sum, void* scratch, size_t scratch_bytes, void* stream) {
    if (num_person < 1 || num_item < 1) return -3;
    if (hidden_dim != kCmH) return -6;
    if (!codes || !feature || !out_sum || !scratch) return -5;
    if (codes_row_stride % 4 != 0 || ((uintptr_t)codes & 3) || ((uintptr_t)scratch & 255)) return -8;
    if (scratch_bytes < vibo_code_table_scratch_bytes(num_person, num_item, hidden_dim)) return -7;
    const int nS = (num_item + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    cm_half8* img = static_cast<cm_half8*>(scratch);
    hipLaunchKernelGGL(cm_table_image_kernel, dim3((nS * 4 * 4 * 64 + 255) / 256), dim3(256), 0, s, feature, img, num_item, nS);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int aligned16 = (codes_row_stride % 16 == 0 && ((uintptr_t)codes & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(cm_forward_kernel, dim3((unsigned)((num_person + 255) / 256)), dim3(256), 0, s, codes, (long long)codes_row_stride,
                       (long long)num_person, num_item, nS, (const cm_half8*)img, out_sum, aligned16);
    return (int)hipGetLastError();
}

/* d feature [2][I][64] = sum over the persons of [code == c] grad_sum[p][:] */
int vibo_code_table_sum_backward(int64_t num_person, int num_item, int hidden_dim, const uint8_t* codes, int64_t codes_row_stride,
                                 const float* grad_sum, float* grad_feature, void* scratch, size_t scratch_bytes, void* stream) {
    if (num_person < 1 || num_item < 1) return -3;
    if (hidden_dim != kCmH) return -6;
    if (!codes || !grad_sum || !grad_feature || !scratch) return -5;
    if (codes_row_stride % 4 != 0 || ((uintptr_t)codes & 3) || ((uintptr_t)scratch & 255)) return -8;
    if (scratch_bytes < vibo_code_table_scratch_bytes(num_person, num_item, hidden_dim)) return -7;
    const int nS = (num_item + 63) / 64;
    long long per_r;
    const int nR = cm_ranges(num_person, nS, &per_r);
    hipStream_t s = (hipStream_t)stream;
    const size_t timg = ((size_t)nS * 4 * 4 * 2 * 64 * 16 + 255) & ~(size_t)255;
    char* base = static_cast<char*>(scratch) + timg;
    cm_half8* gimg = reinterpret_cast<cm_half8*>(base);
    const long long n32 = (num_person + 63) / 64 * 2;
    const size_t gbytes = ((size_t)n32 * 4 * 2 * 64 * 16 + 255) & ~(size_t)255;
    float* rec = reinterpret_cast<float*>(base + gbytes);
    hipLaunchKernelGGL(cm_grad_image_kernel, dim3((unsigned)((n32 * 4 * 64 + 255) / 256)), dim3(256), 0, s, grad_sum, gimg, (long long)num_person, n32);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int aligned16 = (codes_row_stride % 16 == 0 && ((uintptr_t)codes & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(cm_backward_kernel, dim3(nS * nR), dim3(64), 0, s, codes, (long long)codes_row_stride, (long long)num_person, num_item,
                       (const cm_half8*)gimg, rec, nR, per_r, aligned16);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cm_backward_reduce_kernel, dim3((2 * num_item * kCmH + 255) / 256), dim3(256), 0, s, (const float*)rec, grad_feature, num_item, nR);
    return (int)hipGetLastError();
}

}  // extern "C"
