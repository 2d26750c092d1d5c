// Effective shader clock inside matrix-core loops: s_memtime (shader cycles, __builtin_readcyclecounter) against
// s_memrealtime (wall_clock64: a constant 100 MHz) around a bare MFMA loop with no memory traffic, fp32 (32x32x2) and
// fp16 (32x32x16), one and two waves per SIMD, every CU busy.  Prints GHz = d(cycles) / d(ticks) / 10 and TFLOP/s from the
// HIP events around the launch, so that the "of the 2.4 GHz peak" fractions in DESIGN.md can be read against what the chip
// actually clocks at under this kind of load (DESIGN.md lesson 15).  hipcc --offload-arch=gfx950 -O3 clock_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// RND: every lane's operands come from a buffer of random numbers (the chip clocks to its POWER budget and constant
// operands toggle no bits: the same loop runs ~20 % faster on constants -- MI355X_MICROARCH.md "DVFS give-back")
template <bool F16, int NACC, bool RND>
__global__ __launch_bounds__(256) void loop(float* out, unsigned long long* st, int iters, float a, float b, const float* rnd) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 ha, hb;
    for (int r = 0; r < 8; ++r) { ha[r] = (_Float16)a; hb[r] = (_Float16)b; }
    if (RND) {
        const float* q = rnd + 32 * (size_t)(blockIdx.x * 256 + threadIdx.x);
        a = q[0]; b = q[1];
        for (int r = 0; r < 8; ++r) { ha[r] = (_Float16)q[2 + r]; hb[r] = (_Float16)q[10 + r]; }
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* d = st + 2 * (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6));
        d[0] = c1 - c0; d[1] = w1 - w0;
    }
}

template <bool F16, int NACC, bool RND = true>
void run(int blocks, int iters, const char* name) {
    float* out; unsigned long long* st; float* rnd;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&st, (size_t)blocks * 4 * 2 * 8);
    std::vector<float> hr((size_t)blocks * 256 * 32);
    unsigned s_ = 12345u;
    for (auto& v : hr) { s_ = s_ * 1664525u + 1013904223u; v = ((s_ >> 8) & 0xffff) / 65536.f - .5f; }
    hipMalloc(&rnd, hr.size() * 4);
    hipMemcpy(rnd, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) loop<F16, NACC, RND><<<blocks, 256>>>(out, st, iters, 1.f, 2.f, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop<F16, NACC, RND><<<blocks, 256>>>(out, st, iters, 1.f, 2.f, rnd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (size_t w = 0; w < (size_t)blocks * 4; ++w) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] / 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double kflop = F16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    const double flops = (double)blocks * 4 * iters * 8 * NACC * kflop;
    const double peak = F16 ? 2500.0 : 157.3;
    printf("%-44s blocks %4d  %8.3f ms  %8.1f TFLOP/s = %.3f of %.1f   shader clock: median %.3f GHz (p10 %.3f, p90 %.3f)\n", name, blocks, ms,
           flops / ms / 1e9, flops / ms / 1e9 / peak, peak, ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10]);
    hipFree(out); hipFree(st); hipFree(rnd);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int cu = p.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        run<false, 4>(cu, 4000, "fp32 32x32x2, random operands, 1 wave/SIMD");
        run<false, 2>(2 * cu, 4000, "fp32 32x32x2, random operands, 2 waves/SIMD");
        run<true, 4>(cu, 16000, "fp16 32x32x16, random operands, 1 wave/SIMD");
        run<true, 2>(2 * cu, 16000, "fp16 32x32x16, random operands, 2 waves/SIMD");
        run<false, 4>(cu, 40000, "fp32 32x32x2, random operands, 10x longer");
        run<false, 4, false>(cu, 4000, "fp32 32x32x2, CONSTANT operands, 1 wave/SIMD");
        run<true, 4, false>(cu, 16000, "fp16 32x32x16, CONSTANT operands, 1 wave/SIMD");
    }
    return 0;
}
