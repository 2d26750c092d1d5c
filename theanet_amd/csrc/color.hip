// ColorLayer (theanet/layer/color.py:9-52): per-(image, channel) colour balance and gamma jitter, and
// the backward passes of the input-distortion layers when they sit in the MIDDLE of a net
// (neuralnet.py:132-142 allows ElasticLayer / ColorLayer anywhere; Theano differentiates through them).
//
//   out = x / maxval ; out *= b ; out = clip(out, 0, 1) ; out **= g1 ; out = 1 - (1 - out) ** g2 ; out *= maxval
//   b = exp(ln(balance) * u0), g1 = exp(ln(gamma) * u1), g2 = exp(ln(gamma) * u2), u_k ~ U(-1, 1) per (n, c):
//   three random variables of shape (N, C) in the reference (one srs.uniform per pos_rand call).
// The uniforms come from Philox keyed by (seed, step, GLOBAL image index * C + c) -- sharding-proof like
// every other stream here -- or are injected (parity tests replay the oracle's draws); the three factors
// of every (n, c) are kept for the backward pass.  HBM-bound elementwise work: one pass over the tensor.
#include "common.h"

enum { TN_STREAM_COLOR = 5 };

__global__ __launch_bounds__(256) void color_factors_kernel(float* __restrict__ fac, int NC, int C, double lnb, double lng,
                                                           const float* __restrict__ draws, uint32_t k0, uint32_t k1,
                                                           uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // n*C + c
    if (i >= NC) return;
    float u[3];
    if (draws) {
        u[0] = draws[i]; u[1] = draws[NC + i]; u[2] = draws[2 * NC + i];
    } else {
        const uint64_t e = (uint64_t)row_global0 * C + (uint64_t)i;
        const u32x4 r = philox4x32((uint32_t)e, (uint32_t)(e >> 32), step + (d_step ? *d_step : 0u), TN_STREAM_COLOR, k0, k1);
        u[0] = -1.f + 2.f * tn_u01(r.x); u[1] = -1.f + 2.f * tn_u01(r.y); u[2] = -1.f + 2.f * tn_u01(r.z);
    }
    // tt.exp(np.log(a) * uniform).astype(floatX): float64 log constant times a float32 draw -> float64 exp
    fac[3 * i + 0] = (float)exp(lnb * (double)u[0]);
    fac[3 * i + 1] = (float)exp(lng * (double)u[1]);
    fac[3 * i + 2] = (float)exp(lng * (double)u[2]);
}

template <bool BWD>
__global__ __launch_bounds__(256) void color_kernel(const float* __restrict__ x, const float* __restrict__ fac,
                                                   const float* __restrict__ g, float* __restrict__ out, long long total,
                                                   int hw, float maxval, const float* __restrict__ prev_a, int prev_act,
                                                   float prev_prm) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long img = t / hw;                          // n*C + c
    const float xv = x[t];
    const float b = fac[3 * img], g1 = fac[3 * img + 1], g2 = fac[3 * img + 2];
    const float o1 = xv / maxval * b;
    const float o2 = fminf(fmaxf(o1, 0.f), 1.f);
    const float o3 = powf(o2, g1);
    if (!BWD) {
        out[t] = (1.f - powf(1.f - o3, g2)) * maxval;
    } else {
        // d out / d x = g2 (1-o3)^(g2-1) * g1 o2^(g1-1) * [0 <= o1 <= 1] * b     (Theano's Clip gradient is inclusive)
        float d = g2 * powf(1.f - o3, g2 - 1.f) * g1 * powf(o2, g1 - 1.f) * b;
        if (!(o1 >= 0.f && o1 <= 1.f)) d = 0.f;
        float v = g[t] * d;
        if (prev_a && prev_act != TN_ACT_LINEAR) v *= tn_act_grad_from_out(prev_a[t], prev_act, prev_prm);
        out[t] = v;
    }
}

// Backward of tn_elastic_apply: dx[n,c,src] += g[n,c,p] * weight(p, src) * (invert ? -1) * (flipped(p) ? -1),
// one block per (image, channel), the scatter goes through an LDS tile (float atomics stay on-chip).
__global__ __launch_bounds__(256) void elastic_apply_bwd_kernel(
    const float* __restrict__ g, float* __restrict__ dx, int C, int hw, int w, int invert, int nearest,
    const int32_t* __restrict__ map_idx, const float* __restrict__ map_fy, const float* __restrict__ map_fx,
    float pflip, const uint8_t* __restrict__ flipmask, uint32_t k0, uint32_t k1, uint32_t step, const uint32_t* d_step,
    int64_t row_global0, const float* __restrict__ prev_a, int prev_act, float prev_prm) {
    extern __shared__ float tile[];
    const long long img = blockIdx.x;                      // n*C + c
    for (int i = threadIdx.x; i < hw; i += 256) tile[i] = 0.f;
    __syncthreads();
    const uint32_t st = step + (d_step ? *d_step : 0u);
    for (int p = threadIdx.x; p < hw; p += 256) {
        const size_t t = (size_t)img * hw + p;
        float gv = g[t];
        bool flip = false;
        if (flipmask) {
            flip = flipmask[t] != 0;
        } else if (pflip > 0.f) {
            const uint64_t e = (uint64_t)row_global0 * C * hw + (uint64_t)t;
            const uint64_t cq = e >> 2;
            const u32x4 r = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), st, TN_STREAM_FLIP, k0, k1);
            const uint32_t wd = ((e & 3) == 0) ? r.x : ((e & 3) == 1) ? r.y : ((e & 3) == 2) ? r.z : r.w;
            flip = tn_u01(wd) < pflip;
        }
        if (flip) gv = -gv;
        if (invert) gv = -gv;
        if (!map_idx) {
            atomicAdd(&tile[p], gv);
        } else if (nearest) {
            atomicAdd(&tile[map_idx[p]], gv);
        } else {
            const int i00 = map_idx[p];
            const float fy = map_fy[p], fx = map_fx[p];
            atomicAdd(&tile[i00], gv * (1.f - fy) * (1.f - fx));
            atomicAdd(&tile[i00 + 1], gv * (1.f - fy) * fx);
            atomicAdd(&tile[i00 + w], gv * fy * (1.f - fx));
            atomicAdd(&tile[i00 + w + 1], gv * fy * fx);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hw; i += 256) {
        float v = tile[i];
        const size_t t = (size_t)img * hw + i;
        if (prev_a && prev_act != TN_ACT_LINEAR) v *= tn_act_grad_from_out(prev_a[t], prev_act, prev_prm);
        dx[t] = v;
    }
}

// ---- aux-input layers (theanet/layer/auxiliary.py:14-160): the two elementwise pieces around their tiny MLPs ----
enum { TN_STREAM_AUX = 6 };
// LocationInfo's input mix (:27-36): train: a[n,0,:]*u_n + a[n,1,:]*(1-u_n), u_n ~ U(0,1) per row; test: mean of the two
__global__ __launch_bounds__(256) void aux_mix_kernel(const float* __restrict__ aux, float* __restrict__ out, int B, int d,
                                                     float boost, int train, const float* __restrict__ u_inj, uint32_t k0,
                                                     uint32_t k1, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * d) return;
    const int n = i / d, j = i - n * d;
    const float a0 = aux[(size_t)n * 2 * d + j], a1 = aux[(size_t)n * 2 * d + d + j];
    float v;
    if (train) {
        float u;
        if (u_inj) u = u_inj[n];
        else {
            const uint64_t e = (uint64_t)row_global0 + (uint64_t)n;
            u = tn_u01(philox4x32((uint32_t)e, (uint32_t)(e >> 32), step + (d_step ? *d_step : 0u), TN_STREAM_AUX, k0, k1).x);
        }
        v = a0 * u + a1 * (1.f - u);
    } else {
        v = (a0 + a1) / 2.f;
    }
    out[i] = v * boost;
}

// dst[n, col_dst + j] = src[n, col_src + j] (* act'(prev_a[n, col_dst + j])), j < ncols: concatenation and its split
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int ld_src, int col_src,
                                                       float* __restrict__ dst, int ld_dst, int col_dst, int ncols, int B,
                                                       const float* __restrict__ prev_a, int prev_act, float prm) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * ncols) return;
    const int n = (int)(i / ncols), j = (int)(i - (long long)n * ncols);
    float v = src[(size_t)n * ld_src + col_src + j];
    const size_t o = (size_t)n * ld_dst + col_dst + j;
    if (prev_a && prev_act != TN_ACT_LINEAR) v *= tn_act_grad_from_out(prev_a[o], prev_act, prm);
    dst[o] = v;
}

extern "C" {

int tn_aux_mix(tn_ctx* ctx, const float* aux, int64_t row0, float* out, int B, int d, float boost, int train,
               const float* u_inj, uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    TN_REQUIRE(aux && out && B > 0 && d > 0, "tn_aux_mix: bad arguments");
    aux_mix_kernel<<<cdiv((long long)B * d, 256), 256, 0, ctx->stream>>>(aux + (size_t)row0 * 2 * d, out, B, d, boost, train,
                                                                        u_inj, (uint32_t)seed, (uint32_t)(seed >> 32), step,
                                                                        d_step, row_global0);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_copy_cols(tn_ctx* ctx, const float* src, int ld_src, int col_src, float* dst, int ld_dst, int col_dst, int ncols,
                 int B, const float* prev_a, int prev_act, float prev_act_param) {
    TN_REQUIRE(src && dst && B > 0 && ncols > 0 && col_src + ncols <= ld_src && col_dst + ncols <= ld_dst,
               "tn_copy_cols: bad arguments");
    copy_cols_kernel<<<cdiv((long long)B * ncols, 256), 256, 0, ctx->stream>>>(src, ld_src, col_src, dst, ld_dst, col_dst,
                                                                              ncols, B, prev_a, prev_act, prev_act_param);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_color_factors(tn_ctx* ctx, float* fac, int N, int C, double balance, double gamma, const float* draws,
                     uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    TN_REQUIRE(fac && N > 0 && C > 0 && balance > 0 && gamma > 0, "tn_color_factors: bad arguments");
    color_factors_kernel<<<cdiv(N * C, 256), 256, 0, ctx->stream>>>(fac, N * C, C, log(balance), log(gamma), draws,
                                                                   (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step,
                                                                   row_global0);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_color_apply(tn_ctx* ctx, const float* x, int64_t x_row0, const float* fac, float* out,
                   int N, int C, int hw, float maxval) {
    TN_REQUIRE(x && fac && out && N > 0 && C > 0 && hw > 0 && maxval > 0, "tn_color_apply: bad arguments");
    const long long total = (long long)N * C * hw;
    color_kernel<false><<<cdiv(total, 256), 256, 0, ctx->stream>>>(x + (size_t)x_row0 * C * hw, fac, nullptr, out,
                                                                  total, hw, maxval, nullptr, 0, 0.f);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_color_apply_bwd(tn_ctx* ctx, const float* x, int64_t x_row0, const float* fac, const float* g, float* dx, int N,
                       int C, int hw, float maxval, const float* prev_a, int prev_act, float prev_act_param) {
    TN_REQUIRE(x && fac && g && dx && N > 0 && C > 0 && hw > 0 && maxval > 0, "tn_color_apply_bwd: bad arguments");
    const long long total = (long long)N * C * hw;
    color_kernel<true><<<cdiv(total, 256), 256, 0, ctx->stream>>>(x + (size_t)x_row0 * C * hw, fac, g, dx, total,
                                                                 hw, maxval, prev_a, prev_act, prev_act_param);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_elastic_apply_bwd(tn_ctx* ctx, const float* g, float* dx, int N, int C, int h, int w, int invert, int nearest,
                         const int32_t* map_idx, const float* map_fy, const float* map_fx, float pflip,
                         const uint8_t* flipmask, uint64_t seed, uint32_t step, const uint32_t* d_step,
                         int64_t row_global0, const float* prev_a, int prev_act, float prev_act_param) {
    TN_REQUIRE(g && dx && N > 0 && C > 0 && h > 0 && w > 0, "tn_elastic_apply_bwd: bad arguments");
    TN_REQUIRE(!map_idx || nearest || (map_fy && map_fx), "tn_elastic_apply_bwd: bilinear needs map_fy/map_fx");
    const size_t lds = (size_t)h * w * sizeof(float);
    TN_REQUIRE(lds <= 64 * 1024, "tn_elastic_apply_bwd: image %dx%d too large", h, w);
    elastic_apply_bwd_kernel<<<N * C, 256, lds, ctx->stream>>>(g, dx, C, h * w, w, invert, nearest, map_idx, map_fy, map_fx,
                                                              pflip, flipmask, (uint32_t)seed, (uint32_t)(seed >> 32),
                                                              step, d_step, row_global0, prev_a, prev_act, prev_act_param);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
