// Probe: host cost of hipLaunchKernel in a tight C loop (no synchronisation inside), empty kernels and a
// kernel with a 200-byte argument block, plus event record / wait pairs.  tools/probe/launchrate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { long long a[25]; };
__global__ void k_empty() {}
__global__ void k_args(Big b, float* out) { if (b.a[0] == 12345) out[0] = 1.f; }
int main() {
    hipStream_t s[2]; hipStreamCreate(&s[0]); hipStreamCreate(&s[1]);
    hipEvent_t ev[2]; hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
    float* d; hipMalloc(&d, 4);
    Big b{}; 
    for (int i = 0; i < 100; ++i) k_empty<<<1, 64, 0, s[0]>>>();
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        const int n = 2000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) k_empty<<<256, 256, 0, s[0]>>>();
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) k_args<<<256, 256, 0, s[0]>>>(b, d);
        auto t3 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t4 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {          // two streams alternating with an event pair per launch
            const int k = i & 1;
            hipStreamWaitEvent(s[k], ev[1 - k], 0);
            k_args<<<256, 256, 0, s[k]>>>(b, d);
            hipEventRecord(ev[k], s[k]);
        }
        auto t5 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t6 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        printf("empty kernel: host %.2f us/launch (%.2f incl. drain); 200-byte args: %.2f (%.2f); wait+launch+record on two streams: %.2f (%.2f)\n",
               us(t0, t1) / n, us(t0, t2) / n, us(t2, t3) / n, us(t2, t4) / n, us(t4, t5) / n, us(t4, t6) / n);
    }
    return 0;
}
