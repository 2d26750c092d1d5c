// Probe: (1) accuracy of an fp32 product emulated by bf16 triplets on v_mfma_f32_32x32x16_bf16 (6 or 9 partial
// products, fp32 accumulate) against the fp32 MFMA and a float64 host result; (2) do bf16 MFMAs overlap with VALU work?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float a, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned ua = __float_as_uint(a);
    const float fh = __uint_as_float(ua & 0xffff0000u);
    const float r1 = a - fh;
    const unsigned u1 = __float_as_uint(r1);
    const float fm = __uint_as_float(u1 & 0xffff0000u);
    const float r2 = r1 - fm;
    h = ua >> 16; m = u1 >> 16; l = __float_as_uint(r2) >> 16;
}

// A: [32][K] row-major, B: [K][32]; one wave.  mode 0: fp32 mfma, 6: bf16x3 six products, 9: nine products
__global__ void prod(const float* A, const float* B, float* D, int K, int mode) {
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
    f32x16 acc = {0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + hi], B[(k + hi) * 32 + r], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            s16x8 a3[3], b3[3];
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 8 * hi + j;
                unsigned short h, m, l;
                split3(k < K ? A[r * K + k] : 0.f, h, m, l);
                a3[0][j] = h; a3[1][j] = m; a3[2][j] = l;
                split3(k < K ? B[k * 32 + r] : 0.f, h, m, l);
                b3[0][j] = h; b3[1][j] = m; b3[2][j] = l;
            }
            // small terms first
            for (int s = (mode == 9 ? 4 : 2); s >= 0; --s)
                for (int i = 0; i < 3; ++i) {
                    const int j = s - i;
                    if (j < 0 || j > 2) continue;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3[i]),
                                                                  __builtin_bit_cast(bf16x8, b3[j]), acc, 0, 0, 0);
                }
        }
    }
    for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * hi) * 32 + r] = acc[q];
}

template <int MODE, int NV>
__global__ void overlap(float* out, long long* cyc, int iters, float seed) {
    f32x16 w0 = {0}, w1 = {0};
    s16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(threadIdx.x + j); b[j] = (short)(j * 3); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE & 1) {
                w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), w0, 0, 0, 0);
                w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), w1, 0, 0, 0);
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
    float s = w0[0] + w1[3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int NV>
void run(const char* name, int waves) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    overlap<MODE, NV><<<256, 64 * waves>>>(out, cyc, iters, 1.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    overlap<MODE, NV><<<256, 64 * waves>>>(out, cyc, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-24s waves/block %d: %8.1f ticks per trip (16 MFMA bf16 32x32x16, %3d VALU), %7.1f ns per trip\n", name, waves,
           (double)c / iters, 8 * NV, ms * 1e6 / iters);
}

int main() {
    const int K = 720;
    std::vector<float> A(32 * K), B(K * 32);
    srand(1);
    auto u = []() { return (float)rand() / RAND_MAX; };
    for (auto& x : A) { float t = u() * 2.f - .7f; x = t > 0 ? t : 0.f; }            // relu-like activations
    for (auto& x : B) x = (u() + u() + u() - 1.5f) * 0.1f;                            // weights
    std::vector<double> ref(1024, 0.0);
    double maxref = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 32 + j];
        ref[i * 32 + j] = s; maxref = fmax(maxref, fabs(s));
    }
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dD, 4096);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode : {0, 6, 9}) {
        prod<<<1, 64>>>(dA, dB, dD, K, mode);
        std::vector<float> D(1024);
        (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double mx = 0, rms = 0, mrel = 0;
        for (int i = 0; i < 1024; ++i) {
            const double e = fabs(D[i] - ref[i]); mx = fmax(mx, e); rms += e * e;
            mrel = fmax(mrel, e / fmax(fabs(ref[i]), 1e-3 * maxref));
        }
        printf("mode %d (%s): max abs err %.3e (max |ref| %.3f), rms %.3e, max rel %.3e\n", mode,
               mode == 0 ? "fp32 mfma" : (mode == 6 ? "bf16x3, 6 products" : "bf16x3, 9 products"), mx, maxref, sqrt(rms / 1024), mrel);
    }
    for (int waves : {1, 4, 8}) {
        run<1, 0>("bf16 mfma only", waves);
        run<2, 8>("valu only (64)", waves);
        run<3, 8>("bf16 mfma + 64 valu", waves);
        run<2, 16>("valu only (128)", waves);
        run<3, 16>("bf16 mfma + 128 valu", waves);
    }
    return 0;
}
