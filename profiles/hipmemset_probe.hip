// Diagnostic (round 4): does hipMemset return before its fill has run, and can a kernel on a hipStreamNonBlocking stream overtake it?
// (profiles/r04/pmc_fault_diagnosis.txt: the zero-fill of freshly allocated descriptors landed after the kernel that wrote them.)
//   hipcc -O2 --offload-arch=gfx950 hipmemset_probe.hip -o hipmemset_probe && ./hipmemset_probe
// A spin kernel keeps the NULL stream busy for ~50 ms; then hipMemset(buf) is timed, a kernel on a non-blocking stream writes
// 1s into buf and is waited for, and after a device-wide wait the buffer is read back: zeros mean the fill ran after the writer.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (cycles < 0) *sink = 1;
}
__global__ void write_ones(int* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int n = 1024;
    int *buf, *sink, host[1024];
    hipStream_t nb;
    CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
    CK(hipMalloc((void**)&buf, n * sizeof(int)));
    CK(hipMalloc((void**)&sink, sizeof(int)));
    for (int busy = 0; busy < 2; busy++) {
        CK(hipDeviceSynchronize());
        if (busy) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 5000000LL, sink);      // ~50 ms at 100 MHz wall clock, on the null stream
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipMemset(buf, 0, n * sizeof(int)));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        hipLaunchKernelGGL(write_ones, dim3(4), dim3(256), 0, nb, buf, n);
        CK(hipStreamSynchronize(nb));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost));
        int zeros = 0;
        for (int i = 0; i < n; i++) zeros += host[i] == 0;
        printf("null stream %s: hipMemset returned after %.1f us; words of the later writer that were zeroed under it: %d of %d\n",
               busy ? "busy for ~50 ms" : "idle           ", us, zeros, n);
    }
    return 0;
}
