// Probe (diagnostic): semantics of ds_read_b64_tr_b8 on gfx950.  LDS holds a byte matrix M[row][col] with a
// distinct value per cell; each lane supplies an address and we print what every lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(unsigned char* out, int pitch, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 256];
    for (int i = threadIdx.x; i < 64 * 256; i += 64) lds[i] = (unsigned char)(((i / pitch) & 15) << 4 | ((i % pitch) & 15));  // hi nibble=row, lo=col
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, w = l & 15;
    int addr;
    if (mode == 0) addr = (w >> 1) * pitch + (w & 1) * 8 + g * 16;        // lane 2q+p -> row q, cols 8p.. (+16 cols per group)
    else addr = (w & 7) * pitch + (w >> 3) * 8 + g * 16;                  // lane p*8+q -> row q, cols 8p..
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + addr));
    ((v2i*)out)[l] = r;
}
int main() {
    unsigned char* d; hipMalloc(&d, 64 * 8);
    unsigned char h[64 * 8];
    for (int mode = 0; mode < 2; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 64, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (byte = row<<4 | col&15), lanes 0-17 and 32-33:\n", mode);
        for (int l = 0; l < 64; l++) {
            if (l > 17 && !(l == 32 || l == 33)) continue;
            printf(" lane %2d:", l);
            for (int b = 0; b < 8; b++) printf(" %02x", h[l * 8 + b]);
            printf("\n");
        }
    }
    return 0;
}
