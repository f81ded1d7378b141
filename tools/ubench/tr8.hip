// RESULT (MI355X): with lane i of a 16-lane group addressing row (i & 7), 8-byte piece (i >> 3), lane i receives rows
// {0,2,4,6} (i < 8) or {1,3,5,7} (i >= 8) of column (i & 7) followed by the same four rows of column (i & 7) + 8 -- the fp8 MFMA
// operand layout.  It is NOT an 8-rows-of-one-column transpose, so it cannot feed a bf16 operand built from 8 persons of one item.
// what ds_read_b64_tr_b8 returns: LDS holds byte value = its own byte offset (mod 256) in a 64 x 64 byte matrix; every lane passes an
// address and prints the 8 bytes it gets
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint8_t m[64 * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64; i += 64) m[i] = (uint8_t)(((i / 64) << 4) | ((i % 64) & 15));   // high nibble = row (mod 16), low = col (mod 16)
    __syncthreads();
    // mode 0: lane i of a 16-lane group addresses row (i >> 1), 8-byte piece (i & 1) of a [8 rows][16 cols] block, group g -> rows + 8 g
    // mode 1: lane i addresses row (i & 7), piece (i >> 3)
    int row, piece;
    const int i = lane & 15, g = lane >> 4;
    if (mode == 0) { row = 8 * g + (i >> 1); piece = i & 1; } else { row = 8 * g + (i & 7); piece = i >> 3; }
    const uint8_t* p = m + row * 64 + 8 * piece;
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)p);
    out[2 * lane] = r[0]; out[2 * lane + 1] = r[1];
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 8);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        uint32_t h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (each byte printed as row:col nibbles)\n", mode);
        for (int l = 0; l < 64; ++l) {
            if (l % 16 == 0 || l % 16 == 1 || l % 16 == 2 || l % 16 == 8 || l % 16 == 15) {
                printf(" lane %2d:", l);
                for (int b = 0; b < 8; ++b) printf(" %02x", (h[2 * l + b / 4] >> (8 * (b & 3))) & 0xff);
                printf("\n");
            }
        }
    }
    return 0;
}
