// Probe (diagnostic): is an integer contraction exact on the MX FP6 (E3M2) MFMA?
// A[32][64], B[64][32] with entries in -8..8 -> E3M2 codes, 32 six-bit fields per lane (6 dwords),
// lane l = (row/col l&31, k block l>>5), field j = k 32*(l>>5)+j; scale exponent 127 (2^0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__host__ __device__ inline unsigned e3m2(int v) {
    const unsigned tab[9] = {0, 12, 16, 18, 20, 21, 22, 23, 24};   // (exp<<2)|mant, bias 3
    return (v < 0 ? 32u : 0u) | tab[v < 0 ? -v : v];
}
__global__ void k(const int* A, const int* B, float* D, int fmt) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    unsigned long long bits[2][3] = {{0, 0, 0}, {0, 0, 0}};   // 192 bits each
    v8i a = {}, b = {};
    unsigned wa[6] = {0, 0, 0, 0, 0, 0}, wb[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 32; j++) {
        const unsigned ca = e3m2(A[r * 64 + 32 * h + j]), cb = e3m2(B[(32 * h + j) * 32 + r]);
        const int bit = 6 * j, w = bit >> 5, s = bit & 31;
        wa[w] |= ca << s; wb[w] |= cb << s;
        if (s > 26) { wa[w + 1] |= ca >> (32 - s); wb[w + 1] |= cb >> (32 - s); }
    }
    for (int q = 0; q < 6; q++) { a[q] = (int)wa[q]; b[q] = (int)wb[q]; }
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 3, 3, 0, 127, 0, 127);
    for (int g = 0; g < 16; g++) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
    (void)bits; (void)fmt;
}
int main() {
    std::vector<int> A(32 * 64), B(64 * 32);
    srand(3);
    for (auto& v : A) v = rand() % 17 - 8;
    for (auto& v : B) v = rand() % 17 - 8;
    int *dA, *dB; float* dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 3);
    std::vector<float> D(32 * 32);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0; double maxabs = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            long ref = 0;
            for (int kk = 0; kk < 64; kk++) ref += (long)A[i * 64 + kk] * B[kk * 32 + j];
            if ((double)D[i * 32 + j] != (double)ref) { if (bad < 5) printf("mismatch [%d][%d] got %g want %ld\n", i, j, D[i * 32 + j], ref); bad++; }
            if (fabs((double)ref) > maxabs) maxabs = fabs((double)ref);
        }
    printf("fp6 e3m2 32x32x64: %d mismatches of 1024 (max |ref| %g)\n", bad, maxabs);
    return bad != 0;
}
