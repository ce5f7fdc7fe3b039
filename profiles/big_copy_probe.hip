// Diagnostic (round 4): does a large device-to-host hipMemcpyAsync on one stream hold up kernels or HIP calls on another?
// (config 5 through the blocks stalls for ~3 ms whenever CorrAcc publishes its 383 MB long integration: profiles/r04/blocks_gpu_idle.txt)
//   hipcc -O2 --offload-arch=gfx950 big_copy_probe.hip -o big_copy_probe -lpthread && ./big_copy_probe [MB] [chunks] [repeats] [0: hipMemcpyAsync | 1: copy kernel]
// Thread A launches a ~20 us kernel + event record per iteration on a non-blocking stream for ~40 ms and logs, per iteration, the
// host time of the launch call, of the record call, and the kernel's own start clock (written by the kernel).  The main thread
// starts the copy (in `chunks` pieces) on another non-blocking stream 10 ms in, and waits for it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void work(long long cycles, unsigned long long* stamp) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *stamp = (unsigned long long)t0;
    while (wall_clock64() - t0 < cycles) {}
}
__global__ void copy_out(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) dst[k] = src[k];
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 383;
    const int chunks = argc > 2 ? atoi(argv[2]) : 1;
    const size_t nbytes = mb << 20;
    void *dev, *host;
    CK(hipMalloc(&dev, nbytes));
    CK(hipHostMalloc(&host, nbytes, hipHostMallocDefault));
    CK(hipMemset(dev, 1, nbytes));
    CK(hipDeviceSynchronize());
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int NIT = 20000;
    unsigned long long* stamps;
    CK(hipHostMalloc((void**)&stamps, NIT * sizeof(unsigned long long), hipHostMallocDefault));
    std::vector<double> t_launch(NIT), d_launch(NIT), d_record(NIT), d_query(NIT);
    std::atomic<int> nit{0};
    std::atomic<bool> stop{false};
    double t_begin = now_us();
    std::thread a([&] {
        CK(hipSetDevice(0));
        hipEvent_t ev[8];
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        int i = 0;
        for (; i < NIT && !stop.load(); i++) {
            if (i >= 8) CK(hipEventSynchronize(ev[i & 7]));          // at most eight kernels in flight
            const double t0 = now_us();
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, 2000LL, stamps + i);     // ~20 us at the 100 MHz wall clock
            const double t1 = now_us();
            CK(hipEventRecord(ev[i & 7], s1));
            const double t2 = now_us();
            (void)hipEventQuery(ev[(i + 1) & 7]);
            const double t3 = now_us();
            t_launch[i] = t0 - t_begin; d_launch[i] = t1 - t0; d_record[i] = t2 - t1; d_query[i] = t3 - t2;
        }
        nit = i;
        CK(hipStreamSynchronize(s1));
    });
    const int reps = argc > 3 ? atoi(argv[3]) : 4;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;          // 0: hipMemcpyAsync (device -> pinned), 1: a copy KERNEL that stores into the pinned buffer
    hipEvent_t cev;
    CK(hipEventCreateWithFlags(&cev, hipEventDisableTiming));
    std::vector<double> C0(reps), C1(reps), C2(reps);
    for (int r = 0; r < reps; r++) {
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
        C0[r] = now_us() - t_begin;
        const size_t piece = nbytes / chunks;
        for (int k = 0; k < chunks; k++) {
            if (mode == 0) CK(hipMemcpyAsync((char*)host + k * piece, (char*)dev + k * piece, piece, hipMemcpyDeviceToHost, s2));
            else hipLaunchKernelGGL(copy_out, dim3(64), dim3(256), 0, s2, (const uint4*)((char*)dev + k * piece), (uint4*)((char*)host + k * piece), piece / 16);
        }
        C1[r] = now_us() - t_begin;
        CK(hipEventRecord(cev, s2));
        CK(hipEventSynchronize(cev));
        C2[r] = now_us() - t_begin;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
    stop = true;
    a.join();
    const int n = nit.load();
    printf("%zu MB device -> pinned host, %s, %d piece(s), %d times 10 ms apart\n", mb, mode ? "by a 64-group copy kernel" : "hipMemcpyAsync", chunks, reps);
    auto report = [&](const char* what, double lo, double hi) {
        double ml = 0, mr = 0, mq = 0, gap = 0, kgap = 0;
        int cnt = 0;
        for (int i = 1; i < n; i++) {
            if (t_launch[i] < lo || t_launch[i] >= hi) continue;
            cnt++;
            ml = std::max(ml, d_launch[i]); mr = std::max(mr, d_record[i]); mq = std::max(mq, d_query[i]);
            gap = std::max(gap, t_launch[i] - t_launch[i - 1]);
            kgap = std::max(kgap, (double)(stamps[i] - stamps[i - 1]) / 100.0);      // 100 MHz -> us
        }
        printf("  %-16s %5d iterations of the other thread: longest launch call %7.1f us, record %7.1f us, query %5.1f us; longest gap between its kernel starts on the GPU %7.1f us\n",
               what, cnt, ml, mr, mq, kgap);
    };
    report("before any copy", 2000, C0[0]);
    for (int r = 0; r < reps; r++) {
        printf(" copy %d: enqueue calls %7.1f us, complete after %6.2f ms (%5.1f GB/s)\n", r, C1[r] - C0[r], (C2[r] - C0[r]) / 1e3, nbytes / ((C2[r] - C0[r]) * 1e-6) / 1e9);
        report("  during it", C0[r], C2[r]);
    }
    return 0;
}
