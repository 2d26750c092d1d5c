// Max-pool (stride = window, ceil/floor mode), global mean, dropout mask utilities.
// Semantics: theanet/layer/convpool.py:97-144, theanet/layer/dropout.py:9-31.
// All HBM-bound streaming kernels: one thread per output (fwd) / per input (bwd) element,
// consecutive threads on consecutive addresses.
#include "common.h"

__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x,
                                                      float* __restrict__ y, long long total,
                                                      int H, int Wd, int p, int Ho, int Wo) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int HoWo = Ho * Wo;
    const long long nc = t / HoWo;
    const int q = (int)(t - nc * HoWo);
    const int i = q / Wo, j = q - i * Wo;
    const float* xc = x + nc * (long long)H * Wd;
    const int y1 = min(H, (i + 1) * p), x1 = min(Wd, (j + 1) * p);
    float m = -INFINITY;
    for (int yy = i * p; yy < y1; ++yy)
        for (int xx = j * p; xx < x1; ++xx) m = fmaxf(m, xc[yy * Wd + xx]);
    y[t] = m;
}

// dx = (x == max of its window) ? dy[window] : 0 ; then * act'(x) of the producing layer
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ y,
                                                      const float* __restrict__ dy,
                                                      float* __restrict__ dx, long long total, int H,
                                                      int Wd, int p, int Ho, int Wo, int prev_act,
                                                      float prev_prm) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int HW = H * Wd;
    const long long nc = t / HW;
    const int q = (int)(t - nc * HW);
    const int yy = q / Wd, xx = q - yy * Wd;
    const int i = yy / p, j = xx / p;
    float r = 0.f;
    const float xv = x[t];
    if (i < Ho && j < Wo) {
        const long long o = nc * (long long)(Ho * Wo) + i * Wo + j;
        if (xv == y[o]) r = dy[o];
    }
    if (prev_act != TN_ACT_LINEAR) r *= tn_act_grad_from_out(xv, prev_act, prev_prm);
    dx[t] = r;
}

__global__ __launch_bounds__(64) void mean_fwd_kernel(const float* __restrict__ x,
                                                     float* __restrict__ y, int HW) {
    const float* xc = x + (size_t)blockIdx.x * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 64) s += xc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) y[blockIdx.x] = s / HW;
}

__global__ __launch_bounds__(256) void mean_bwd_kernel(const float* __restrict__ dy,
                                                      float* __restrict__ dx, long long total, int HW,
                                                      const float* __restrict__ prev_a, int prev_act,
                                                      float prev_prm) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    float r = dy[t / HW] / HW;
    if (prev_a) r *= tn_act_grad_from_out(prev_a[t], prev_act, prev_prm);
    dx[t] = r;
}

// 4 mask bytes per thread from one Philox call; idx = global element index / 4
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, size_t n,
                                                          float pdrop, uint32_t k0, uint32_t k1,
                                                          uint32_t step, const uint32_t* d_step,
                                                          uint64_t elem0) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;   // local quad index
    const size_t i0 = q * 4;
    if (i0 >= n) return;
    const uint32_t st = step + (d_step ? *d_step : 0u);
    // counters are keyed by the GLOBAL element index so that sharding does not change the mask:
    // element e uses word (e & 3) of philox(counter = e >> 2).  elem0 need not be 4-aligned.
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = i0 + k;
        if (i >= n) break;
        const uint64_t e = elem0 + i;
        const uint64_t cq = e >> 2;
        const u32x4 r = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), st, TN_STREAM_DROPOUT, k0, k1);
        const uint32_t w = ((e & 3) == 0) ? r.x : ((e & 3) == 1) ? r.y : ((e & 3) == 2) ? r.z : r.w;
        mask[i] = tn_u01(w) >= pdrop ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void scale_mask_kernel(const float* __restrict__ x,
                                                        const uint8_t* __restrict__ mask, float scale,
                                                        float* __restrict__ y, size_t n,
                                                        const float* __restrict__ prev_a, int prev_act,
                                                        float prev_prm) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        float r = x[i] * scale;
        if (mask) r *= (float)mask[i];
        if (prev_a) r *= tn_act_grad_from_out(prev_a[i], prev_act, prev_prm);
        y[i] = r;
    }
}

extern "C" {

int tn_pool_fwd(tn_ctx* ctx, const float* x, float* y, int NC, int H, int Wd, int p, int Ho, int Wo) {
    TN_REQUIRE(NC > 0 && p > 0 && Ho > 0 && Wo > 0 && (Ho - 1) * p < H && (Wo - 1) * p < Wd,
               "tn_pool_fwd: bad geometry H=%d W=%d p=%d Ho=%d Wo=%d", H, Wd, p, Ho, Wo);
    const long long total = (long long)NC * Ho * Wo;
    pool_fwd_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(x, y, total, H, Wd, p, Ho, Wo);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_pool_bwd(tn_ctx* ctx, const float* x, const float* y, const float* dy, float* dx, int NC, int H,
                int Wd, int p, int Ho, int Wo, int prev_act, float prev_act_param) {
    TN_REQUIRE(NC > 0 && p > 0 && Ho > 0 && Wo > 0, "tn_pool_bwd: bad geometry");
    const long long total = (long long)NC * H * Wd;
    pool_bwd_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(x, y, dy, dx, total, H, Wd, p, Ho, Wo,
                                                              prev_act, prev_act_param);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_mean_fwd(tn_ctx* ctx, const float* x, float* y, int NC, int HW) {
    mean_fwd_kernel<<<NC, 64, 0, ctx->stream>>>(x, y, HW);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_mean_bwd(tn_ctx* ctx, const float* dy, float* dx, int NC, int HW, const float* prev_a,
                int prev_act, float prev_act_param) {
    const long long total = (long long)NC * HW;
    mean_bwd_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(dy, dx, total, HW, prev_a, prev_act,
                                                              prev_act_param);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_dropout_mask(tn_ctx* ctx, uint8_t* mask, size_t n, float pdrop, uint64_t seed, uint32_t step,
                    const uint32_t* d_step, uint64_t elem0) {
    if (!n) return TN_OK;
    const size_t quads = (n + 3) / 4;
    dropout_mask_kernel<<<cdiv(quads, 256), 256, 0, ctx->stream>>>(
        mask, n, pdrop, (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, elem0);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_scale_mask(tn_ctx* ctx, const float* x, const uint8_t* mask, float scale, float* y, size_t n,
                  const float* prev_a, int prev_act, float prev_act_param) {
    if (!n) return TN_OK;
    int blocks = cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    scale_mask_kernel<<<blocks, 256, 0, ctx->stream>>>(x, mask, scale, y, n, prev_a, prev_act,
                                                      prev_act_param);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
