// How fast can W waves per SIMD feed v_mfma_f32_32x32x16_f16 from LDS through ds_read_b64_tr_b16?  One block per CU, 4*W waves,
// each wave loops: R transposing reads (software-pipelined one step ahead) + M products on M accumulators.  Prints cycles per
// step per wave and the matrix-pipe utilisation (M * 32 * W / cycles per step).  hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half4v tr16(const char* l) {
    return __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)l));
}
template <int R, int M, bool B64>
__global__ __launch_bounds__(1024) void k(unsigned long long* st, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 120 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    f32x16 acc[M];
    for (int m = 0; m < M; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    // c8-style addresses: group of 16 lanes = 4 pixels x 2 planes x 2 halves; planes 64 (mod 256) bytes apart
    const int grp = lane >> 4, r4 = (lane >> 2) & 3, q8 = lane & 3;
    const char* base = lds + (wave % 3) * 2112 * 8 + (2 * (grp & 1) + (q8 >> 1)) * 2112 + (q8 & 1) * 8 + (8 * (grp >> 1) + r4) * 16;
    half4v v[2][R > 0 ? R : 1];
    auto load = [&](int b, int step) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const char* p = base + (step & 7) * 256 + (r >> 1) * 16 + (r & 1) * 64 + (r >> 3) * 4224;
            if (B64) v[b][r] = *reinterpret_cast<const half4v*>(p);
            else v[b][r] = tr16(p);
        }
    };
    const unsigned long long c0 = __builtin_readcyclecounter();
    load(0, 0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            load(1 - h, it + h + 1);
            __builtin_amdgcn_sched_barrier(0);
            half8 a, b;
            if (R >= 2) { a = half8{v[h][0][0], v[h][0][1], v[h][0][2], v[h][0][3], v[h][1][0], v[h][1][1], v[h][1][2], v[h][1][3]}; }
            else { for (int e = 0; e < 8; ++e) a[e] = (_Float16)(lane * 0.001f); }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if (R >= 2 * m + 4) b = half8{v[h][2 * m + 2][0], v[h][2 * m + 2][1], v[h][2 * m + 2][2], v[h][2 * m + 2][3], v[h][2 * m + 3][0], v[h][2 * m + 3][1], v[h][2 * m + 3][2], v[h][2 * m + 3][3]};
                else b = a;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < M; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + t] = s;
    if (lane == 0) st[blockIdx.x * 16 + wave] = c1 - c0;
}
template <int R, int M, bool B64>
void run(int W, const char* name) {
    const int G = 128, iters = 4000;
    unsigned long long* st; float* out;
    hipMalloc(&st, G * 16 * 8); hipMalloc(&out, G * 1024 * 4);
    hipFuncSetAttribute((const void*)k<R, M, B64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) { k<R, M, B64><<<G, 256 * W, 160 * 1024>>>(st, out, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(G * 16);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < G; ++b) for (int w = 0; w < 4 * W; ++w) c.push_back((double)h[b * 16 + w] / iters);
    std::sort(c.begin(), c.end());
    const double med = c[c.size() / 2], worst = c[c.size() - 1];
    printf("%-26s reads/step %2d  products/step %d  waves/SIMD %d : %7.1f cycles per step (slowest wave %7.1f)  matrix pipe %4.0f %%   LDS %5.1f B/clk/CU\n",
           name, R, M, W, med, worst, 100.0 * M * 32 * W / worst, R * 512.0 * 4 * W / worst);
    hipFree(st); hipFree(out);
}
int main() {
    for (int W = 1; W <= 4; ++W) {
        run<0, 3, false>(W, "no LDS reads");
        run<8, 3, false>(W, "ds_read_b64_tr_b16");
        run<8, 3, true>(W, "ds_read_b64 (plain)");
        run<4, 3, false>(W, "ds_read_b64_tr_b16");
        run<14, 6, false>(W, "ds_read_b64_tr_b16");
    }
    return 0;
}
