// Per-CU ingest rate HBM/L2 -> LDS on gfx950: how many bytes per clock ONE CU can pull with
//   (a) LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction), W loader waves, D instructions in flight per wave,
//   (b) plain global_load_dwordx4 into registers + ds_write_b128,
// with G of the chip's CUs streaming at once (one block per CU, 160 KB of LDS requested so that nothing else fits).
// Each block streams its own 8 MB window of a big buffer (HBM-resident, no reuse).  hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst, bool nt) {
    if (nt) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(gsrc), "s"(lds_dst) : "memory");
    else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int D, bool NT>
__global__ __launch_bounds__(1024) void dma_kernel(const char* src, size_t window, int iters, unsigned long long* st, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * window;
    const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + wave * (D * 1024);
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    size_t off = (size_t)wave * 1024 + lane * 16;
    const size_t stride = (size_t)nw * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            glds16(base + (off % window), __builtin_amdgcn_readfirstlane(l0 + d * 1024), NT);
            off += stride;
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }
    if (sink && threadIdx.x == 12345) sink[0] = *reinterpret_cast<float*>(lds);
}

template <int D>
__global__ __launch_bounds__(1024) void reg_kernel(const char* src, size_t window, int iters, unsigned long long* st, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * window;
    char* const dst = lds + threadIdx.x * 16;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    size_t off = (size_t)threadIdx.x * 16;
    const size_t stride = (size_t)nw * 1024;
    uint4 v[D];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            v[d] = *reinterpret_cast<const uint4*>(base + (off % window));
            off += stride;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) *reinterpret_cast<uint4*>(dst + (d & 1) * 16384) = v[d];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }
    if (sink && threadIdx.x == 12345) sink[0] = *reinterpret_cast<float*>(lds);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const size_t window = 8u << 20;
    char* src; hipMalloc(&src, window * 256); hipMemset(src, 1, window * 256);
    unsigned long long* st; hipMalloc(&st, 256 * 16);
    auto report = [&](const char* name, int G, int W, int D, int iters) {
        std::vector<unsigned long long> h(2 * G);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> bpc;
        double ghz = 0;
        for (int b = 0; b < G; ++b) { bpc.push_back((double)iters * D * W * 1024 / (double)h[2 * b]); ghz += (double)h[2 * b] / h[2 * b + 1] / 10.0 / G; }
        std::sort(bpc.begin(), bpc.end());
        printf("%-34s CUs %3d  waves %2d  depth %2d : %6.1f B/clk/CU (p10 %5.1f p90 %5.1f)  = %6.2f TB/s aggregate at %.2f GHz\n", name, G, W, D,
               bpc[G / 2], bpc[G / 10], bpc[G * 9 / 10], bpc[G / 2] * G * ghz / 1e3, ghz);
    };
#define RUN_DMA(D_, NT_, G, W, name) do { const int iters = 4096 / D_; \
        hipFuncSetAttribute((const void*)dma_kernel<D_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        dma_kernel<D_, NT_><<<G, 64 * W, 160 * 1024>>>(src, window, iters, st, nullptr); hipDeviceSynchronize(); \
        dma_kernel<D_, NT_><<<G, 64 * W, 160 * 1024>>>(src, window, iters, st, nullptr); hipDeviceSynchronize(); report(name, G, W, D_, iters); } while (0)
#define RUN_REG(D_, G, W, name) do { const int iters = 4096 / D_; \
        hipFuncSetAttribute((const void*)reg_kernel<D_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        reg_kernel<D_><<<G, 64 * W, 160 * 1024>>>(src, window, iters, st, nullptr); hipDeviceSynchronize(); \
        reg_kernel<D_><<<G, 64 * W, 160 * 1024>>>(src, window, iters, st, nullptr); hipDeviceSynchronize(); report(name, G, W, D_, iters); } while (0)
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    for (int G : {128, 256}) {
        RUN_DMA(4, false, G, 1, "LDS-DMA"); RUN_DMA(8, false, G, 1, "LDS-DMA"); RUN_DMA(8, false, G, 2, "LDS-DMA"); RUN_DMA(8, false, G, 4, "LDS-DMA");
        RUN_DMA(8, false, G, 8, "LDS-DMA"); RUN_DMA(8, false, G, 16, "LDS-DMA"); RUN_DMA(16, false, G, 4, "LDS-DMA");
        RUN_DMA(8, true, G, 4, "LDS-DMA nt"); RUN_DMA(8, true, G, 8, "LDS-DMA nt");
        RUN_REG(4, G, 4, "global_load x4 -> ds_write_b128"); RUN_REG(8, G, 4, "global_load x4 -> ds_write_b128");
        RUN_REG(8, G, 8, "global_load x4 -> ds_write_b128"); RUN_REG(8, G, 16, "global_load x4 -> ds_write_b128");
    }
    return 0;
}
