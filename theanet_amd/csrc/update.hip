// Momentum SGD (+L1/L2 gradient terms) and max-norm projection.
// Semantics: theanet/layer/layer.py:70-107 -- simultaneous Theano updates:
//   v' = m*v + (1-m)*g ;  p' = p - rate*lr*v   (the OLD velocity moves p) ; maxnorm(p').
#include "common.h"
#include "update_body.h"

__global__ __launch_bounds__(256) void sgd_update_kernel(float* __restrict__ p, float* __restrict__ v,
                                                        const float* __restrict__ g, size_t n,
                                                        float momentum, float rate,
                                                        const float* __restrict__ d_lr, float L1,
                                                        float L2, float gscale) {
    const float step = rate * d_lr[0];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const float pv = p[i], vv = v[i];
        float gg = g[i] * gscale;
        if (L1 != 0.f) gg += L1 * ((pv > 0.f) - (pv < 0.f));
        if (L2 != 0.f) gg += 2.f * L2 * pv;
        v[i] = momentum * vv + (1.f - momentum) * gg;
        p[i] = pv - step * vv;
    }
}

// all parameter tensors of the net in ONE launch: blockIdx.y selects the segment descriptor
__global__ __launch_bounds__(256) void sgd_update_multi_kernel(const tn_sgd_seg* __restrict__ segs,
                                                              int nseg, const float* __restrict__ d_lr,
                                                              float gscale, uint32_t* d_step_inc,
                                                              const float* __restrict__ rowloss,
                                                              int nrow, float cost_scale,
                                                              float* __restrict__ d_cost) {
    __shared__ float red[4];
    sgd_update_multi_block(segs, nseg, d_lr, gscale, d_step_inc, rowloss, nrow, cost_scale, d_cost,
                           blockIdx.x, blockIdx.y, gridDim.x, red);
}

__global__ __launch_bounds__(256) void clip_kernel(float* __restrict__ p, size_t n, float mx) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = fminf(fmaxf(p[i], -mx), mx);
}

// ndim 4: one block per leading index d0, contiguous 'rest' elements
__global__ __launch_bounds__(256) void maxnorm_rows_kernel(float* __restrict__ p, int rest, float mx) {
    __shared__ float red[4];
    __shared__ float scale_s;
    float* row = p + (size_t)blockIdx.x * rest;
    float s = 0.f;
    for (int i = threadIdx.x; i < rest; i += 256) s += row[i] * row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
        scale_s = (1e-7f + fminf(fmaxf(nrm, 0.f), mx)) / (1e-7f + nrm);
    }
    __syncthreads();
    const float sc = scale_s;
    for (int i = threadIdx.x; i < rest; i += 256) row[i] *= sc;
}

// ndim 2 (rows x cols, row-major): per-COLUMN norm.  Block = 64 columns x 4 row-lanes.
__global__ __launch_bounds__(256) void maxnorm_cols_kernel(float* __restrict__ p, int rows, int cols,
                                                          float mx) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r0 = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols)
        for (int r = r0; r < rows; r += 4) {
            const float v = p[(size_t)r * cols + c];
            s += v * v;
        }
    red[r0][threadIdx.x & 63] = s;
    __syncthreads();
    const int cl = threadIdx.x & 63;
    const float nrm = sqrtf(red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
    const float sc = (1e-7f + fminf(fmaxf(nrm, 0.f), mx)) / (1e-7f + nrm);
    if (c < cols)
        for (int r = r0; r < rows; r += 4) p[(size_t)r * cols + c] *= sc;
}

extern "C" {

int tn_sgd_update(tn_ctx* ctx, float* p, float* v, const float* g, size_t n, float momentum,
                  float rate, const float* d_lr, float L1, float L2, float gscale) {
    if (!n) return TN_OK;
    TN_REQUIRE(d_lr != nullptr, "tn_sgd_update: d_lr is NULL");
    int blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    sgd_update_kernel<<<blocks, 256, 0, ctx->stream>>>(p, v, g, n, momentum, rate, d_lr, L1, L2,
                                                      gscale);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_sgd_update_multi(tn_ctx* ctx, const tn_sgd_seg* d_segs, int nseg, size_t max_n,
                        const float* d_lr, float gscale, uint32_t* d_step_inc) {
    return tn_sgd_update_multi_cost(ctx, d_segs, nseg, max_n, d_lr, gscale, d_step_inc, nullptr, 0, 0.f,
                                    nullptr);
}

int tn_sgd_update_multi_cost(tn_ctx* ctx, const tn_sgd_seg* d_segs, int nseg, size_t max_n,
                             const float* d_lr, float gscale, uint32_t* d_step_inc,
                             const float* rowloss, int nrow, float cost_scale, float* d_cost) {
    const bool rider = rowloss != nullptr;
    if (nseg <= 0 && !rider) return TN_OK;
    TN_REQUIRE(nseg <= 0 || (d_segs != nullptr && d_lr != nullptr), "tn_sgd_update_multi: NULL argument");
    TN_REQUIRE(!rider || (d_cost != nullptr && nrow > 0), "tn_sgd_update_multi_cost: bad cost arguments");
    if (nseg < 0) nseg = 0;
    int bx = cdiv(max_n, 1024);
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    sgd_update_multi_kernel<<<dim3(bx, nseg + (rider ? 1 : 0)), 256, 0, ctx->stream>>>(
        d_segs, nseg, d_lr, gscale, d_step_inc, rowloss, nrow, cost_scale, d_cost);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_maxnorm(tn_ctx* ctx, float* p, int ndim, int d0, int rest, float maxnorm) {
    if (maxnorm == 0.f) return TN_OK;
    if (ndim == 1) {
        clip_kernel<<<cdiv(d0, 256), 256, 0, ctx->stream>>>(p, (size_t)d0, maxnorm);
    } else if (ndim == 2) {
        maxnorm_cols_kernel<<<cdiv(rest, 64), 256, 0, ctx->stream>>>(p, d0, rest, maxnorm);
    } else if (ndim == 4) {
        maxnorm_rows_kernel<<<d0, 256, 0, ctx->stream>>>(p, rest, maxnorm);
    } else {
        return tn_fail(ctx, TN_E_ARG, "tn_maxnorm: ndim %d unsupported (1, 2 or 4)", ndim);
    }
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
