// Probe: issue cost (shader cycles per wave64 instruction, one wave alone on its SIMD and 2 / 4 waves per SIMD) of the
// vector instructions the VALU-bound kernels of the mnist.prms step are made of: v_fma_f32 (VGPR and SGPR operand),
// v_pk_fma_f32 (VGPR and SGPR-pair operand), v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 (Philox), v_xor.
// Each body = 64 instructions over 8 independent chains, inline asm so the compiler cannot rewrite them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { FMA_V, FMA_S, PK_V, PK_S, MUL_LO, MUL_HI, MAD64, XOR, FMAC };

template <int KIND>
__global__ void probe(float* out, long long* cyc, int iters, float seed, unsigned useed) {
    float v[8];
    f32x2 p[8];
    unsigned u[8];
    unsigned long long q[8];
    for (int i = 0; i < 8; ++i) {
        v[i] = seed + i + threadIdx.x;
        p[i] = f32x2{v[i], v[i] + 1.f};
        u[i] = useed + i * 77u + threadIdx.x;
        q[i] = u[i];
    }
    float s0 = seed * 1.0001f;                       // wave-uniform -> SGPR
    f32x2 sp = {seed * 1.0001f, seed * 0.9999f};
    s0 = __builtin_amdgcn_readfirstlane(s0);
    sp.x = __builtin_amdgcn_readfirstlane(sp.x);
    sp.y = __builtin_amdgcn_readfirstlane(sp.y);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == FMA_V) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(v[(k + 1) & 7]));
                if (KIND == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[k]) : "v"(v[(k + 1) & 7]), "v"(v[(k + 2) & 7]));
                if (KIND == FMA_S) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k]) : "s"(s0));
                if (KIND == PK_V) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[k]) : "v"(p[(k + 1) & 7]));
                if (KIND == PK_S) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[k]) : "s"(sp));
                if (KIND == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (KIND == MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (KIND == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[k]) : "v"(u[k]), "v"(u[(k + 1) & 7]) : "vcc");
                if (KIND == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + (float)u[i] + (float)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int waves : {4, 8, 16}) {                   // per block of one CU: 1, 2, 4 waves per SIMD
        probe<KIND><<<256, 64 * waves>>>(out, cyc, iters, 1.f, 12345u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<KIND><<<256, 64 * waves>>>(out, cyc, iters, 1.f, 12345u);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_simd = waves / 4.0 * 64 * iters;      // wave-instructions issued per SIMD
        printf("%-28s %d wave(s)/SIMD: %6.2f cycles per instruction in one wave's stream, %6.2f ns*2.4 per SIMD-instruction (wall)\n",
               name, waves / 4, (double)c / (64.0 * iters), ms * 1e6 * 2.4 / per_simd);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<FMA_V>("v_fma_f32 vgpr");
    run<FMAC>("v_fmac_f32 vgpr");
    run<FMA_S>("v_fma_f32 sgpr operand");
    run<PK_V>("v_pk_fma_f32 vgpr");
    run<PK_S>("v_pk_fma_f32 sgpr pair");
    run<MUL_LO>("v_mul_lo_u32");
    run<MUL_HI>("v_mul_hi_u32");
    run<MAD64>("v_mad_u64_u32");
    run<XOR>("v_xor_b32");
    return 0;
}
