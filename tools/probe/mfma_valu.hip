// Probe: do v_mfma_f32_16x16x4_f32 and plain VALU instructions of ONE wave overlap?  cycles (s_memtime) per loop trip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV, int BIG>
__global__ void probe(float* out, long long* cyc, int iters, float seed) {
    f32x4 z0 = {0, 0, 0, 0}, z1 = {0, 0, 0, 0};
    f32x16 w0 = {0}, w1 = {0};
    float a = seed + threadIdx.x, b = seed * 2.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE & 1) {
                if (BIG) {
                    w0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, w0, 0, 0, 0);
                    w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, w1, 0, 0, 0);
                } else {
                    z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, z0, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, z1, 0, 0, 0);
                }
            }
            if (MODE & 2) {
#pragma unroll
                for (int k = 0; k < NV; ++k) v[k & 7] = fmaf(v[k & 7], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV - NV / 2, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = z0[0] + z1[1] + w0[0] + w1[3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int NV, int BIG>
void run(const char* name, int waves) {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    const int iters = 2000;
    probe<MODE, NV, BIG><<<256, 64 * waves>>>(out, cyc, iters, 1.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<MODE, NV, BIG><<<256, 64 * waves>>>(out, cyc, iters, 1.f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/block %d: %8.1f counter ticks per trip (16 MFMA, %3d VALU), %7.1f ns per trip -> %.2f ticks/ns\n", name, waves,
           (double)c / iters, 8 * NV, ms * 1e6 / iters, (double)c / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int waves : {1, 4, 8}) {
        run<1, 0, 0>("mfma16x16x4 only", waves);
        run<2, 8, 0>("valu only (64)", waves);
        run<3, 8, 0>("mfma16 + 64 valu", waves);
        run<2, 16, 0>("valu only (128)", waves);
        run<3, 16, 0>("mfma16 + 128 valu", waves);
        run<1, 0, 1>("mfma32x32x2 only", waves);
        run<3, 8, 1>("mfma32 + 64 valu", waves);
        run<3, 16, 1>("mfma32 + 128 valu", waves);
    }
    return 0;
}
