// Micro-benchmark: raw v_mfma_f32_32x32x2_f32 issue rate (no memory), to calibrate the GEMM.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, const char* name) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("%-28s blocks=%4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(256, "1 acc, 1 wave/SIMD");
    run<2>(256, "2 acc, 1 wave/SIMD");
    run<4>(256, "4 acc, 1 wave/SIMD");
    run<1>(512, "1 acc, 2 waves/SIMD");
    run<2>(512, "2 acc, 2 waves/SIMD");
    run<1>(1024, "1 acc, 4 waves/SIMD");
    return 0;
}
