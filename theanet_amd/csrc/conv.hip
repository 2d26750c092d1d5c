// Direct (VALU) NCHW convolution kernels for small channel counts (C*f*f below the MFMA
// break-even, SURVEY.md section 7 "hard parts"): thread = one output pixel, KT output maps per
// thread, weights fetched with wave-uniform (scalar) loads.  HBM/L2-bound by design: x is
// read once from HBM (the f*f re-reads hit L1), a is written once.
//
// Semantics: theanet/layer/convpool.py:54-72 (true convolution, W is flipped).
#include "common.h"

// ------------------------------------------------------------------------------------
// forward: a = act(conv(x, W) + b)
// ------------------------------------------------------------------------------------
template <int F, int KT>
__global__ __launch_bounds__(256) void conv_fwd_direct(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    float* __restrict__ a, int N, int C, int H, int Wd, int K, int f_rt, int stride, int pad,
    int Ho, int Wo, int act, float prm) {
    const int f = F ? F : f_rt;
    const int HoWo = Ho * Wo;
    const long long M = (long long)N * HoWo;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * KT;
    if (m >= M) return;
    const int n = (int)(m / HoWo);
    const int p = (int)(m - (long long)n * HoWo);
    const int i = p / Wo, j = p - i * Wo;
    const int y0 = i * stride - pad, x0 = j * stride - pad;

    float acc[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) acc[kk] = (k0 + kk < K) ? b[k0 + kk] : 0.f;

    const int ff = f * f;
    for (int c = 0; c < C; ++c) {
        const float* xc = x + ((size_t)n * C + c) * H * Wd;
        const float* wc = W + ((size_t)k0 * C + c) * ff;
#pragma unroll
        for (int u = 0; u < f; ++u) {
            const int yy = y0 + u;
            const bool yok = (yy >= 0) && (yy < H);
#pragma unroll
            for (int v = 0; v < f; ++v) {
                const int xx = x0 + v;
                float xv = 0.f;
                if (yok && xx >= 0 && xx < Wd) xv = xc[yy * Wd + xx];
                const int widx = (f - 1 - u) * f + (f - 1 - v);
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) {
                    // wave-uniform address -> s_load
                    const float wv = (k0 + kk < K) ? wc[(size_t)kk * C * ff + widx] : 0.f;
                    acc[kk] = fmaf(xv, wv, acc[kk]);
                }
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        if (k0 + kk < K)
            a[((size_t)n * K + k0 + kk) * HoWo + p] = tn_act_fwd(acc[kk], act, prm);
    }
}

// ------------------------------------------------------------------------------------
// wgrad: partial[blk][k][c][u][v] (correlation order), db partial; second kernel reduces
// over blocks deterministically and writes the flipped dW.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int F, int KT, int CT>
__global__ __launch_bounds__(256) void conv_wgrad_direct(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    float* __restrict__ dbpartial, int N, int C, int H, int Wd, int K, int stride, int pad,
    int Ho, int Wo) {
    constexpr int FF = F * F;
    __shared__ float red[4][KT * CT * FF + KT];
    const int HoWo = Ho * Wo;
    const long long M = (long long)N * HoWo;
    const int k0 = blockIdx.y * KT, c0 = blockIdx.z * CT;

    float acc[KT][CT][FF];
    float accb[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        accb[kk] = 0.f;
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
#pragma unroll
            for (int t = 0; t < FF; ++t) acc[kk][cc][t] = 0.f;
    }

    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M;
         m += (long long)gridDim.x * 256) {
        const int n = (int)(m / HoWo);
        const int p = (int)(m - (long long)n * HoWo);
        const int i = p / Wo, j = p - i * Wo;
        const int y0 = i * stride - pad, x0 = j * stride - pad;
        float dzv[KT];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            dzv[kk] = (k0 + kk < K) ? dz[((size_t)n * K + k0 + kk) * HoWo + p] : 0.f;
            accb[kk] += dzv[kk];
        }
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) {
            if (c0 + cc < C) {
                const float* xc = x + ((size_t)n * C + c0 + cc) * H * Wd;
#pragma unroll
                for (int u = 0; u < F; ++u) {
                    const int yy = y0 + u;
                    const bool yok = (yy >= 0) && (yy < H);
#pragma unroll
                    for (int v = 0; v < F; ++v) {
                        const int xx = x0 + v;
                        float xv = 0.f;
                        if (yok && xx >= 0 && xx < Wd) xv = xc[yy * Wd + xx];
#pragma unroll
                        for (int kk = 0; kk < KT; ++kk)
                            acc[kk][cc][u * F + v] = fmaf(dzv[kk], xv, acc[kk][cc][u * F + v]);
                    }
                }
            }
        }
    }

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
#pragma unroll
            for (int t = 0; t < FF; ++t) {
                float s = wave_sum_f(acc[kk][cc][t]);
                if (lane == 0) red[wave][(kk * CT + cc) * FF + t] = s;
            }
        float sb = wave_sum_f(accb[kk]);
        if (lane == 0) red[wave][KT * CT * FF + kk] = sb;
    }
    __syncthreads();
    const int KCFF = K * C * FF;
    for (int t = threadIdx.x; t < KT * CT * FF + KT; t += 256) {
        const float s = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        if (t < KT * CT * FF) {
            const int kk = t / (CT * FF), r = t - kk * CT * FF, cc = r / FF, uv = r - cc * FF;
            if (k0 + kk < K && c0 + cc < C)
                partial[(size_t)blockIdx.x * KCFF + ((size_t)(k0 + kk) * C + c0 + cc) * FF + uv] = s;
        } else if (blockIdx.z == 0) {
            const int kk = t - KT * CT * FF;
            if (k0 + kk < K) dbpartial[(size_t)blockIdx.x * K + k0 + kk] = s;
        }
    }
}

// generic fallback (any f): one block per (k, c, u, v), threads stride over pixels
__global__ __launch_bounds__(256) void conv_wgrad_generic(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ dW,
    int N, int C, int H, int Wd, int K, int f, int stride, int pad, int Ho, int Wo) {
    __shared__ float red[4];
    const int ff = f * f;
    int t = blockIdx.x;
    const int uv = t % ff;
    t /= ff;
    const int c = t % C, k = t / C;
    const int u = uv / f, v = uv % f;
    const int HoWo = Ho * Wo;
    const long long M = (long long)N * HoWo;
    float acc = 0.f;
    for (long long m = threadIdx.x; m < M; m += 256) {
        const int n = (int)(m / HoWo);
        const int p = (int)(m - (long long)n * HoWo);
        const int i = p / Wo, j = p - i * Wo;
        const int yy = i * stride - pad + u, xx = j * stride - pad + v;
        if (yy >= 0 && yy < H && xx >= 0 && xx < Wd)
            acc = fmaf(dz[((size_t)n * K + k) * HoWo + p],
                       x[(((size_t)n * C + c) * H + yy) * Wd + xx], acc);
    }
    acc = wave_sum_f(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        dW[((size_t)k * C + c) * ff + (f - 1 - u) * f + (f - 1 - v)] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void conv_bgrad_generic(const float* __restrict__ dz,
                                                         float* __restrict__ db, int N, int K, int HoWo) {
    __shared__ float red[4];
    const int k = blockIdx.x;
    const long long M = (long long)N * HoWo;
    float acc = 0.f;
    for (long long m = threadIdx.x; m < M; m += 256) {
        const int n = (int)(m / HoWo);
        const int p = (int)(m - (long long)n * HoWo);
        acc += dz[((size_t)n * K + k) * HoWo + p];
    }
    acc = wave_sum_f(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) db[k] = red[0] + red[1] + red[2] + red[3];
}


// ------------------------------------------------------------------------------------
// dgrad: dx[n,c,y,x] = sum_{k,u,v} dz[n,k,(y+pad-u)/s,(x+pad-v)/s] * W[k,c,f-1-u,f-1-v]
// optional fused activation gradient of the layer below (its output prev_a has dx's shape)
// ------------------------------------------------------------------------------------
template <int F, int CT>
__global__ __launch_bounds__(256) void conv_dgrad_direct(
    const float* __restrict__ dz, const float* __restrict__ W, float* __restrict__ dx, int N, int C,
    int H, int Wd, int K, int f_rt, int stride, int pad, int Ho, int Wo,
    const float* __restrict__ prev_a, int prev_act, float prev_prm) {
    const int f = F ? F : f_rt;
    const int ff = f * f;
    const int HW = H * Wd, HoWo = Ho * Wo;
    const long long M = (long long)N * HW;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c0 = blockIdx.y * CT;
    if (m >= M) return;
    const int n = (int)(m / HW);
    const int p = (int)(m - (long long)n * HW);
    const int y = p / Wd, xq = p - y * Wd;

    float acc[CT];
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) acc[cc] = 0.f;

    if (F > 0 && stride == 1) {
        // stride-1 fast path: branch-free, fully unrolled taps (independent loads in flight)
        for (int k = 0; k < K; ++k) {
            const float* dzk = dz + ((size_t)n * K + k) * HoWo;
            const float* wk = W + ((size_t)k * C + c0) * ff;
#pragma unroll
            for (int u = 0; u < f; ++u) {
                const int i = y + pad - u;
                const bool iok = (unsigned)i < (unsigned)Ho;
                const int ic = min(max(i, 0), Ho - 1);
#pragma unroll
                for (int v = 0; v < f; ++v) {
                    const int j = xq + pad - v;
                    const bool ok = iok && ((unsigned)j < (unsigned)Wo);
                    const int jc = min(max(j, 0), Wo - 1);
                    const float t = dzk[ic * Wo + jc];
                    const float g = ok ? t : 0.f;
                    const int widx = (f - 1 - u) * f + (f - 1 - v);
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc) {
                        const float wv = (c0 + cc < C) ? wk[(size_t)cc * ff + widx] : 0.f;
                        acc[cc] = fmaf(g, wv, acc[cc]);
                    }
                }
            }
        }
    } else {
        for (int k = 0; k < K; ++k) {
            const float* dzk = dz + ((size_t)n * K + k) * HoWo;
            const float* wk = W + ((size_t)k * C + c0) * ff;
            for (int u = 0; u < f; ++u) {
                const int ty = y + pad - u;
                if (ty < 0) continue;
                const int i = ty / stride;
                if (i * stride != ty || i >= Ho) continue;
                for (int v = 0; v < f; ++v) {
                    const int tx = xq + pad - v;
                    if (tx < 0) continue;
                    const int j = tx / stride;
                    if (j * stride != tx || j >= Wo) continue;
                    const float g = dzk[i * Wo + j];
                    const int widx = (f - 1 - u) * f + (f - 1 - v);
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc) {
                        const float wv = (c0 + cc < C) ? wk[(size_t)cc * ff + widx] : 0.f;
                        acc[cc] = fmaf(g, wv, acc[cc]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
        if (c0 + cc < C) {
            const size_t o = ((size_t)n * C + c0 + cc) * HW + p;
            float r = acc[cc];
            if (prev_a) r *= tn_act_grad_from_out(prev_a[o], prev_act, prev_prm);
            dx[o] = r;
        }
    }
}

// LDS-staged dgrad (stride 1).  A block stages dz of G whole images in LDS with a ZERO HALO of
// f-1 pixels around every map, plus the weights as one float4 (4 input channels) per (k, tap).
// Then thread = one input pixel: every tap is an unconditional LDS read (no bounds selects, so
// the reads pipeline), the weights are wave-broadcast b128 reads, 4 FMAs per tap.
//   dx[n,c,y,x] = sum_{k,u',v'} dzh[n,k,y+pad+u',x+pad+v'] * W[k,c,u',v']      (dzh = haloed dz)
template <int F>
__global__ __launch_bounds__(256) void conv_dgrad_lds(
    const float* __restrict__ dz, const float* __restrict__ W, float* __restrict__ dx, int N, int C,
    int H, int Wd, int K, int pad, int Ho, int Wo, int G, const float* __restrict__ prev_a,
    int prev_act, float prev_prm) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int FF = F * F;
    const int Hh = Ho + 2 * (F - 1), Wh = Wo + 2 * (F - 1);
    const int plane = Hh * Wh, HoWo = Ho * Wo, HW = H * Wd;
    float4* sW = reinterpret_cast<float4*>(sm);                  // [K][FF]
    float* sdz = sm + (size_t)K * FF * 4;                        // [G][K][Hh][Wh]
    const int n0 = blockIdx.x * G;
    const int gcnt = min(G, N - n0);
    const int c0 = blockIdx.y * 4;
    for (int t = threadIdx.x; t < K * FF; t += 256) {
        const int k = t / FF, uv = t - k * FF;
        float4 w;
        w.x = (c0 + 0 < C) ? W[((size_t)k * C + c0 + 0) * FF + uv] : 0.f;
        w.y = (c0 + 1 < C) ? W[((size_t)k * C + c0 + 1) * FF + uv] : 0.f;
        w.z = (c0 + 2 < C) ? W[((size_t)k * C + c0 + 2) * FF + uv] : 0.f;
        w.w = (c0 + 3 < C) ? W[((size_t)k * C + c0 + 3) * FF + uv] : 0.f;
        sW[t] = w;
    }
    const float* src = dz + (size_t)n0 * K * HoWo;
    for (int t = threadIdx.x; t < gcnt * K * plane; t += 256) {
        const int gk = t / plane, r = t - gk * plane;
        const int ii = r / Wh - (F - 1), jj = r % Wh - (F - 1);
        const bool in = ((unsigned)ii < (unsigned)Ho) && ((unsigned)jj < (unsigned)Wo);
        const float v = src[(size_t)gk * HoWo + min(max(ii, 0), Ho - 1) * Wo + min(max(jj, 0), Wo - 1)];
        sdz[t] = in ? v : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < gcnt * HW; t += 256) {
        const int gi = t / HW;
        const int p = t - gi * HW;
        const int y = p / Wd, xq = p - y * Wd;
        const float* base = sdz + (size_t)gi * K * plane + (y + pad) * Wh + xq + pad;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < K; ++k) {
            const float* bk = base + k * plane;
            const float4* wk = sW + k * FF;
#pragma unroll
            for (int u = 0; u < F; ++u)
#pragma unroll
                for (int v = 0; v < F; ++v) {
                    const float gv = bk[u * Wh + v];
                    const float4 w = wk[u * F + v];
                    acc.x = fmaf(gv, w.x, acc.x);
                    acc.y = fmaf(gv, w.y, acc.y);
                    acc.z = fmaf(gv, w.z, acc.z);
                    acc.w = fmaf(gv, w.w, acc.w);
                }
        }
        const float r4[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            if (c0 + cc < C) {
                const size_t o = ((size_t)(n0 + gi) * C + c0 + cc) * HW + p;
                float r = r4[cc];
                if (prev_a) r *= tn_act_grad_from_out(prev_a[o], prev_act, prev_prm);
                dx[o] = r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------

// MFMA implicit-GEMM path (conv_mfma.hip)
int tn_conv_mfma_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int N,
                     int C, int H, int Wd, int K, int f, int pad, int Ho, int Wo, int act, float prm);
int tn_conv_mfma_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int N, int C, int H,
                       int Wd, int K, int f, int pad, int Ho, int Wo, const float* prev_a, int act,
                       float prm);
int tn_conv_mfma_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N,
                       int C, int H, int Wd, int K, int f, int pad, int Ho, int Wo);
extern "C" int tn_conv_mfma_supported(int C, int K, int f, int stride);
int tn_conv_tile_smallc_ok(const float* x, const float* dz, int N, int C, int H, int Wd, int K, int f, int pad,
                           int Ho, int Wo);
int tn_conv_tile_smallc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N, int C,
                              int H, int Wd, int K);
int tn_conv_tile_ok(const float* x, int N, int C, int H, int Wd, int K, int f, int pad, int Ho, int Wo);
int tn_conv_tile_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int N, int C,
                     int H, int Wd, int K, int pad, int Ho, int Wo, int act, float prm);

int tn_conv_wgrad_finish(tn_ctx* ctx, const float* partial, const float* dbpartial, float* dW,
                         float* db, int nblk, int K, int C, int f) {
    // slabs are in correlation layout: the sum flips every f x f block back (reduce.hip)
    int rc = tn_red_push(ctx, partial, dW, (uint32_t)(K * C * f * f), (uint32_t)nblk,
                         (uint32_t)(K * C * f * f), (uint32_t)(f * f));
    if (rc) return rc;
    rc = tn_red_push(ctx, dbpartial, db, (uint32_t)K, (uint32_t)nblk, (uint32_t)K, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

extern "C" {

int tn_conv2d_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int N,
                  int C, int H, int Wd, int K, int f, int stride, int pad_lo, int Ho, int Wo,
                  int act, float act_param) {
    TN_REQUIRE(N > 0 && C > 0 && K > 0 && f > 0 && stride > 0 && Ho > 0 && Wo > 0,
               "tn_conv2d_fwd: bad shape");
    TN_REQUIRE(!ctx->mm_f16, "tn_conv2d_fwd: fp32 tensors in DTYPE float16 mode (the mode's entry points are tn_c8_*)");
    if (tn_conv_mfma_supported(C, K, f, stride))
        return tn_conv_mfma_fwd(ctx, x, W, b, a, N, C, H, Wd, K, f, pad_lo, Ho, Wo, act, act_param);
    // few input channels (first layers): still worth the matrix core when the LDS-tile kernel applies
    if (stride == 1 && K >= 16 && tn_conv_tile_ok(x, N, C, H, Wd, K, f, pad_lo, Ho, Wo))
        return tn_conv_tile_fwd(ctx, x, W, b, a, N, C, H, Wd, K, pad_lo, Ho, Wo, act, act_param);
    const long long M = (long long)N * Ho * Wo;
    const int gx = cdiv(M, 256);
#define LAUNCH_FWD(F_, KT_)                                                                     \
    conv_fwd_direct<F_, KT_><<<dim3(gx, cdiv(K, KT_)), 256, 0, ctx->stream>>>(                   \
        x, W, b, a, N, C, H, Wd, K, f, stride, pad_lo, Ho, Wo, act, act_param)
    const bool k8 = (K % 8 == 0) || K > 32;
    if (f == 3) {
        if (k8) LAUNCH_FWD(3, 8); else LAUNCH_FWD(3, 4);
    } else if (f == 5) {
        if (k8) LAUNCH_FWD(5, 8); else LAUNCH_FWD(5, 4);
    } else {
        if (k8) LAUNCH_FWD(0, 8); else LAUNCH_FWD(0, 4);
    }
#undef LAUNCH_FWD
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_conv2d_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N, int C,
                    int H, int Wd, int K, int f, int stride, int pad_lo, int Ho, int Wo) {
    TN_REQUIRE(N > 0 && C > 0 && K > 0 && f > 0 && stride > 0, "tn_conv2d_wgrad: bad shape");
    TN_REQUIRE(!ctx->mm_f16, "tn_conv2d_wgrad: fp32 tensors in DTYPE float16 mode (the mode's entry points are tn_c8_*)");
    if (tn_conv_mfma_supported(C, K, f, stride))
        return tn_conv_mfma_wgrad(ctx, x, dz, dW, db, N, C, H, Wd, K, f, pad_lo, Ho, Wo);
    if (stride == 1 && tn_conv_tile_smallc_ok(x, dz, N, C, H, Wd, K, f, pad_lo, Ho, Wo))   // first layers
        return tn_conv_tile_smallc_wgrad(ctx, x, dz, dW, db, N, C, H, Wd, K);
    const long long M = (long long)N * Ho * Wo;
    if (f == 3 || f == 5 || f == 1 || f == 2) {
        int nblk = cdiv(M, 256 * 16);
        if (nblk > 1024) nblk = 1024;
        if (nblk < 1) nblk = 1;
        const size_t KCFF = (size_t)K * C * f * f;
        const size_t need = ((size_t)nblk * (KCFF + K)) * sizeof(float);
        float* partial;
        int rc = tn_scratch_get(ctx, need, &partial);
        if (rc) return rc;
        float* dbpartial = partial + (size_t)nblk * KCFF;
#define LAUNCH_WG(F_, KT_, CT_)                                                                  \
    conv_wgrad_direct<F_, KT_, CT_><<<dim3(nblk, cdiv(K, KT_), cdiv(C, CT_)), 256, 0, ctx->stream>>>( \
        x, dz, partial, dbpartial, N, C, H, Wd, K, stride, pad_lo, Ho, Wo)
        if (f == 3) {
            if (C >= 4) LAUNCH_WG(3, 4, 4);
            else if (C >= 2) LAUNCH_WG(3, 4, 2);
            else LAUNCH_WG(3, 4, 1);
        } else if (f == 5) {
            if (C >= 2) LAUNCH_WG(5, 2, 2); else LAUNCH_WG(5, 4, 1);
        } else if (f == 2) {
            if (C >= 4) LAUNCH_WG(2, 4, 4); else LAUNCH_WG(2, 4, 1);
        } else {
            if (C >= 4) LAUNCH_WG(1, 8, 4); else LAUNCH_WG(1, 8, 1);
        }
#undef LAUNCH_WG
        TN_LAUNCH_CHECK();
        rc = tn_conv_wgrad_finish(ctx, partial, dbpartial, dW, db, nblk, K, C, f);
        if (rc) return rc;
    } else {
        conv_wgrad_generic<<<K * C * f * f, 256, 0, ctx->stream>>>(x, dz, dW, N, C, H, Wd, K, f,
                                                                  stride, pad_lo, Ho, Wo);
        TN_LAUNCH_CHECK();
        conv_bgrad_generic<<<K, 256, 0, ctx->stream>>>(dz, db, N, K, Ho * Wo);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}

int tn_conv2d_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int N, int C, int H,
                    int Wd, int K, int f, int stride, int pad_lo, int Ho, int Wo, const float* prev_a,
                    int prev_act, float prev_act_param) {
    TN_REQUIRE(N > 0 && C > 0 && K > 0 && f > 0 && stride > 0, "tn_conv2d_dgrad: bad shape");
    TN_REQUIRE(!ctx->mm_f16, "tn_conv2d_dgrad: fp32 tensors in DTYPE float16 mode (the mode's entry points are tn_c8_*)");
    if (tn_conv_mfma_supported(K, C, f, stride))     // reduction K*f*f, rows = C input maps
        return tn_conv_mfma_dgrad(ctx, dz, W, dx, N, C, H, Wd, K, f, pad_lo, Ho, Wo, prev_a, prev_act,
                                  prev_act_param);
    if (f == 3 && stride == 1) {
        const size_t per = (size_t)K * (Ho + 4) * (Wo + 4) * sizeof(float);
        const size_t wbytes = (size_t)K * 9 * 16;
        if (per + wbytes <= 60 * 1024) {
            int G = (int)((60 * 1024 - wbytes) / per);
            if (G > 4) G = 4;
            const size_t lds = wbytes + (size_t)G * per;
            conv_dgrad_lds<3><<<dim3(cdiv(N, G), cdiv(C, 4)), 256, lds, ctx->stream>>>(
                dz, W, dx, N, C, H, Wd, K, pad_lo, Ho, Wo, G, prev_a, prev_act, prev_act_param);
            TN_LAUNCH_CHECK();
            return TN_OK;
        }
    }
    const long long M = (long long)N * H * Wd;
    const int gx = cdiv(M, 256);
#define LAUNCH_DG(F_, CT_)                                                                       \
    conv_dgrad_direct<F_, CT_><<<dim3(gx, cdiv(C, CT_)), 256, 0, ctx->stream>>>(                  \
        dz, W, dx, N, C, H, Wd, K, f, stride, pad_lo, Ho, Wo, prev_a, prev_act, prev_act_param)
    if (f == 3) {
        if (C >= 8) LAUNCH_DG(3, 8); else LAUNCH_DG(3, 4);
    } else {
        if (C >= 8) LAUNCH_DG(0, 8); else LAUNCH_DG(0, 4);
    }
#undef LAUNCH_DG
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
