// Probe: v_mfma_f32_4x4x1_16b_f32 -- (1) issue cost per instruction (cycles, one wave and two waves per SIMD) against
// v_mfma_f32_16x16x4_f32; (2) semantics of the CBSZ / ABID broadcast modifiers: with cbsz = n the A operand of block
// (j & ~(2^n - 1)) + abid is used by all 2^n blocks of its group.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void rate(float* out, long long* cyc, int iters, float seed) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 2.f + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 0, 0, 0);
                if (KIND == 1) acc[k] = (k & 1) ? __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 3, 1, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 3, 2, 0);
                if (KIND == 2) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
            }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void sem(float* out) {
    const int l = threadIdx.x;
    // A = 100 + lane, B = 1 at one lane: D[lane][reg] = sum over the (broadcast) A block x B block
    int idx = 0;
    for (int cb = 0; cb <= 4; ++cb)
        for (int ab = 0; ab < (1 << cb) && ab < 4; ++ab)
            for (int sel : {0, 5, 21, 42}) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                const float a = 100.f + l, b = l == sel ? 1.f : 0.f;
                switch (cb * 4 + ab) {
#define CASE(CB, AB) case CB * 4 + AB: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CB, AB, 0); break;
                    CASE(0, 0) CASE(1, 0) CASE(1, 1) CASE(2, 0) CASE(2, 1) CASE(2, 2) CASE(2, 3)
                    CASE(3, 0) CASE(3, 1) CASE(3, 2) CASE(3, 3) CASE(4, 0) CASE(4, 1) CASE(4, 2) CASE(4, 3)
                }
                for (int r = 0; r < 4; ++r) out[(idx * 64 + l) * 4 + r] = c[r];
                ++idx;
            }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 16 << 20); hipMalloc(&cyc, 8);
    const char* names[] = {"4x4x1_16b", "4x4x1_16b cbsz=3 abid=k", "16x16x4"};
    for (int kind = 0; kind < 3; ++kind)
        for (int waves : {4, 8}) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) rate<0><<<256, 64 * waves>>>(out, cyc, iters, 1.f);
                if (kind == 1) rate<1><<<256, 64 * waves>>>(out, cyc, iters, 1.f);
                if (kind == 2) rate<2><<<256, 64 * waves>>>(out, cyc, iters, 1.f);
                hipDeviceSynchronize();
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-28s %d wave(s)/SIMD: %.2f cycles per instruction in one wave's stream\n", names[kind], waves / 4,
                   (double)c / (32.0 * iters));
        }
    sem<<<1, 64>>>(out);
    static float h[60 * 64 * 4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int idx = 0;
    for (int cb = 0; cb <= 4; ++cb)
        for (int ab = 0; ab < (1 << cb) && ab < 4; ++ab)
            for (int sel : {0, 5, 21, 42}) {
                printf("cbsz %d abid %d, B = e(lane %2d):", cb, ab, sel);
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 4; ++r) {
                        const float v = h[(idx * 64 + l) * 4 + r];
                        if (v != 0.f) printf(" D[l%d][r%d]=A(l%d)", l, r, (int)v - 100);
                    }
                printf("\n");
                ++idx;
            }
    return 0;
}
