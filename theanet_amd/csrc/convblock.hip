// LDS-resident backward of a conv + act + 2x2-max-pool block with MANY filter elements
// (K*C*f*f of a few hundred, e.g. mnist.prms conv2: 20 x 4 x 3 x 3 = 720), where the
// register-accumulator scheme of convpool.hip runs out of registers.
//
// A block owns G whole images at a time (persistent loop over image groups):
//   stage  : x tile of the G images -> LDS (zero padded for mode 'same')
//   phase 1: item = (pooling window, group of 4 filters): recompute the window's conv outputs
//            from the LDS patch, act, max (every tie gets the gradient), dz = g * act'(a);
//            dz goes into an LDS tile with a zero halo of f-1 pixels -- it never touches HBM.
//   phase 2: dgrad, thread = input pixel: unconditional LDS taps x broadcast float4 weights.
//   phase 3: wgrad, thread = (k, c, u): slides along the rows of dz / x in LDS and keeps its
//            f accumulators in registers across ALL groups of the block; one partial slab
//            per block is written at the end and reduced (fixed order) by conv_wgrad_finish.
// HBM traffic: x, g read once, dx written once.  Semantics: theanet/layer/convpool.py:54-72,
// :106-112; Theano MaxPoolGrad tie rule.
#include <cstdlib>

#include "common.h"

int tn_conv_wgrad_finish(tn_ctx* ctx, const float* partial, const float* dbpartial, float* dW,
                         float* db, int nblk, int K, int C, int f);
// convblock_mfma.hip: matrix-core variant of the same backward
int tn_convblock_mfma_supported(int C, int K, int f, int stride, int p, int H, int Wd, int pad_lo,
                                int Ho, int Wo, int Hp, int Wp);
int tn_convblock_mfma_bwd(tn_ctx* ctx, const float* x, const float* W, const float* b, const float* g,
                          float* dx, float* dW, float* db, int N, int C, int H, int Wd, int K,
                          int pad_lo, int Ho, int Wo, int Hp, int Wp, int act, float act_param);

template <int ACT>
__device__ __forceinline__ float cb_act(float z, int act, float prm) {
    if (ACT == TN_ACT_LEAKY) return fmaxf(0.f, z) + fminf(0.f, z) * prm;
    return tn_act_fwd(z, act, prm);
}
template <int ACT>
__device__ __forceinline__ float cb_actg(float a, int act, float prm) {
    if (ACT == TN_ACT_LEAKY) return a > 0.f ? 1.f : (a < 0.f ? prm : (prm > 0.f ? 1.f + prm : 0.f));
    return tn_act_grad_from_out(a, act, prm);
}

// LDS rows are padded so that wide (b64 / b128) reads are naturally aligned:
//   x  tile: rows of Wx = 4*ceil(Wo/4) + 4 floats (data at col 0..Wo+F-2, zero beyond)
//   dz tile: rows of Wh = 4*ceil(Wo/4) + 4 floats, interior at col CB_LP = 4 (16-byte aligned);
//            the left pad doubles as the previous row's right halo (rows are contiguous).
#define CB_LP 4
struct CbGeom {
    int N, H, Wd, K, pad, Ho, Wo, Hp, Wp, G;
    int Hx, Wx;      // x tile:  Hx = Ho+F-1 rows of Wx floats
    int Hh, Wh;      // dz tile: Hh = Ho+2(F-1) rows of Wh floats
    int dbg;         // ablation (TN_CB_DBG): 1 skip phase 1, 2 skip phase 2, 4 skip phase 3, 8 skip staging
};

#define CB_NT 1024     // 16 waves per block: four per SIMD hide the LDS latency of the phases

template <int F, int C, int ACT>
__global__ __launch_bounds__(CB_NT) void convblock_bwd_lds(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    const float* __restrict__ g, float* __restrict__ dx, float* __restrict__ partial,
    float* __restrict__ dbpartial, CbGeom q, int act, float prm) {
    constexpr int FF = F * F, KQ = 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = q.K, G = q.G;
    const int xplane = q.Hx * q.Wx, hplane = q.Hh * q.Wh, HpWp = q.Hp * q.Wp, HW = q.H * q.Wd;
    float4* sW4 = reinterpret_cast<float4*>(sm);             // [K][FF]   W[k][c0..3][u][v] (dgrad)
    float* sWc = sm + (size_t)K * FF * 4;                    // [K][C][F][4] flipped (conv), v padded to 4
    float* sb = sWc + (size_t)K * C * F * 4;                 // [K]
    float* sx = sb + ((K + 3) & ~3);                         // [G][C][Hx][Wx]
    float* sdz = sx + (((size_t)G * C * xplane + 3) & ~(size_t)3);   // [G][K][Hh][Wh] (+8 zero floats)
    const int tid = threadIdx.x;

    for (int t = tid; t < K * FF; t += CB_NT) {
        const int k = t / FF, uv = t - k * FF;
        float4 w;
        w.x = W[((size_t)k * C + 0) * FF + uv];
        w.y = (C > 1) ? W[((size_t)k * C + (C > 1 ? 1 : 0)) * FF + uv] : 0.f;
        w.z = (C > 2) ? W[((size_t)k * C + (C > 2 ? 2 : 0)) * FF + uv] : 0.f;
        w.w = (C > 3) ? W[((size_t)k * C + (C > 3 ? 3 : 0)) * FF + uv] : 0.f;
        sW4[t] = w;
    }
    for (int t = tid; t < K * C * F * 4; t += CB_NT) {
        const int v = t & 3, u = (t >> 2) % F, kc = t / (4 * F);
        sWc[t] = (v < F) ? W[(size_t)kc * FF + (F - 1 - u) * F + (F - 1 - v)] : 0.f;   // flipped
    }
    for (int t = tid; t < K; t += CB_NT) sb[t] = b[t];
    for (int t = tid; t < G * K * hplane + 8; t += CB_NT) sdz[t] = 0.f;   // halo / pads / never-covered cells stay 0

    // wgrad item of this thread: (image slot wg, k, c, u)
    const int KCF = K * C * F;
    const int wg = tid / KCF, wr = tid - wg * KCF;
    const int wk = wr / (C * F), wc = (wr / F) % C, wu = wr % F;
    const bool wlive = wg < G;
    float wacc[F];
#pragma unroll
    for (int v = 0; v < F; ++v) wacc[v] = 0.f;
    float bacc = 0.f;

    const int ngroups = (q.N + G - 1) / G;
    const int KG = (K + KQ - 1) / KQ;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int n0 = grp * G;
        const int gcnt = min(G, q.N - n0);
        __syncthreads();                       // previous group's readers are done
        // ---- stage x (zero padded) ------------------------------------------------------
        for (int t = tid; t < ((q.dbg & 8) ? 0 : gcnt * C * xplane); t += CB_NT) {
            const int gc = t / xplane, r = t - gc * xplane;
            const int yy = r / q.Wx - q.pad, xx = r % q.Wx - q.pad;     // cols beyond the data are zero
            const bool in = ((unsigned)yy < (unsigned)q.H) && ((unsigned)xx < (unsigned)q.Wd);
            const float v = x[((size_t)n0 * C + gc) * HW + min(max(yy, 0), q.H - 1) * q.Wd +
                              min(max(xx, 0), q.Wd - 1)];
            sx[t] = in ? v : 0.f;
        }
        __syncthreads();
        // ---- phase 1: dz into the haloed LDS tile ------------------------------------------
        const int nwin = gcnt * HpWp;
        for (int it = tid; it < ((q.dbg & 1) ? 0 : nwin * KG); it += CB_NT) {
            const int kg = it / nwin, w = it - kg * nwin;     // window fastest: a wave shares kg
            const int gi = w / HpWp, wq = w - gi * HpWp;
            const int pi = wq / q.Wp, pj = wq - pi * q.Wp;
            const int k0 = kg * KQ;
            float gk[KQ];
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk)
                gk[kk] = g[((size_t)(n0 + gi) * K + min(k0 + kk, K - 1)) * HpWp + wq];
            // one channel's (F+1)^2 patch at a time (16 registers), 4 filters x 4 positions of
            // accumulators; rows/cols beyond the padded tile only feed invalid outputs
            int ro[F + 1];
#pragma unroll
            for (int r = 0; r <= F; ++r) ro[r] = min(2 * pi + r, q.Hx - 1) * q.Wx + 2 * pj;
            const float* sxg = sx + (size_t)gi * C * xplane;
            float z[KQ][2][2];
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const float bias = sb[min(k0 + kk, K - 1)];
                z[kk][0][0] = z[kk][0][1] = z[kk][1][0] = z[kk][1][1] = bias;
            }
#pragma unroll 1
            for (int c = 0; c < C; ++c) {
                float pt[F + 1][F + 1];      // F == 3: 4 x 4 patch, rows are two aligned b64 reads
#pragma unroll
                for (int r = 0; r <= F; ++r) {
                    const float2* pr = reinterpret_cast<const float2*>(sxg + c * xplane + ro[r]);
                    const float2 lo = pr[0], up = pr[1];
                    pt[r][0] = lo.x; pt[r][1] = lo.y; pt[r][2] = up.x; pt[r][3] = up.y;
                }
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) {
                    const float4* wk_ = reinterpret_cast<const float4*>(sWc) +
                                        ((size_t)min(k0 + kk, K - 1) * C + c) * F;
#pragma unroll
                    for (int u = 0; u < F; ++u) {
                        const float4 w4 = wk_[u];
                        const float wrow[3] = {w4.x, w4.y, w4.z};
#pragma unroll
                        for (int v = 0; v < F; ++v) {
                            const float wv = wrow[v];
                            z[kk][0][0] = fmaf(pt[u][v], wv, z[kk][0][0]);
                            z[kk][0][1] = fmaf(pt[u][v + 1], wv, z[kk][0][1]);
                            z[kk][1][0] = fmaf(pt[u + 1][v], wv, z[kk][1][0]);
                            z[kk][1][1] = fmaf(pt[u + 1][v + 1], wv, z[kk][1][1]);
                        }
                    }
                }
            }
            const bool v01 = (2 * pj + 1 < q.Wo), v10 = (2 * pi + 1 < q.Ho);
            const bool vld[2][2] = {{true, v01}, {v10, v10 && v01}};
            float* dzg = sdz + ((size_t)gi * K) * hplane + (2 * pi + F - 1) * q.Wh + 2 * pj + CB_LP;
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int k = k0 + kk;
                if (k < K) {
                    float m = -INFINITY;
#pragma unroll
                    for (int di = 0; di < 2; ++di)
#pragma unroll
                        for (int dj = 0; dj < 2; ++dj) {
                            z[kk][di][dj] = cb_act<ACT>(z[kk][di][dj], act, prm);
                            m = vld[di][dj] ? fmaxf(m, z[kk][di][dj]) : m;
                        }
                    float* dzk = dzg + (size_t)k * hplane;
#pragma unroll
                    for (int di = 0; di < 2; ++di)
#pragma unroll
                        for (int dj = 0; dj < 2; ++dj) {
                            // invalid positions fall on the (zero) halo: storing their 0 is harmless,
                            // and the unconditional store keeps the phase branch-free
                            const float d = (vld[di][dj] && z[kk][di][dj] == m)
                                                ? gk[kk] * cb_actg<ACT>(z[kk][di][dj], act, prm) : 0.f;
                            dzk[di * q.Wh + dj] = d;
                        }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: dgrad --------------------------------------------------------------
        if (dx && !(q.dbg & 2)) {
            for (int t = tid; t < gcnt * HW; t += CB_NT) {
                const int gi = t / HW, p = t - gi * HW;
                const int y = p / q.Wd, xq = p - y * q.Wd;
                const float* base = sdz + (size_t)gi * K * hplane + (y + q.pad) * q.Wh + xq + q.pad +
                                    (CB_LP - (F - 1));
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = 0; k < K; ++k) {
                    const float* bk = base + k * hplane;
                    const float4* w4 = sW4 + k * FF;
#pragma unroll
                    for (int u = 0; u < F; ++u)
#pragma unroll
                        for (int v = 0; v < F; ++v) {
                            const float gv = bk[u * q.Wh + v];
                            const float4 w = w4[u * F + v];
                            acc.x = fmaf(gv, w.x, acc.x);
                            acc.y = fmaf(gv, w.y, acc.y);
                            acc.z = fmaf(gv, w.z, acc.z);
                            acc.w = fmaf(gv, w.w, acc.w);
                        }
                }
                float* o = dx + (size_t)(n0 + gi) * C * HW + p;
                o[0] = acc.x;
                if (C > 1) o[(size_t)HW] = acc.y;
                if (C > 2) o[(size_t)2 * HW] = acc.z;
                if (C > 3) o[(size_t)3 * HW] = acc.w;
            }
        }
        // ---- phase 3: wgrad (registers persist over the groups) ------------------------------
        if (wlive && wg < gcnt && !(q.dbg & 4)) {
            {
                const int gi = wg;
                const float* dzp = sdz + ((size_t)gi * K + wk) * hplane + (F - 1) * q.Wh + CB_LP;
                const float* xp = sx + ((size_t)gi * C + wc) * xplane + wu * q.Wx;
                const int nq = (q.Wo + 3) >> 2;
                for (int i = 0; i < q.Ho; ++i) {
                    const float4* dr = reinterpret_cast<const float4*>(dzp + i * q.Wh);
                    const float4* xr = reinterpret_cast<const float4*>(xp + i * q.Wx);
                    float4 xa = xr[0];
                    for (int jq = 0; jq < nq; ++jq) {      // 4 outputs: 1 dz + 1 x b128 read, 12 FMAs
                        const float4 d = dr[jq];           // cols beyond Wo are zero
                        const float4 xb = xr[jq + 1];
                        const float xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                        const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
#pragma unroll
                            for (int v = 0; v < F; ++v) wacc[v] = fmaf(dd[e], xs[e + v], wacc[v]);
                        }
                        if (wc == 0 && wu == 0) bacc += (dd[0] + dd[1]) + (dd[2] + dd[3]);
                        xa = xb;
                    }
                }
            }
        }
    }
    if (wlive) {        // one partial slab per (block, image slot); conv_wgrad_finish sums them
        const size_t slab = (size_t)blockIdx.x * G + wg;
        float* p = partial + slab * K * C * FF + ((size_t)wk * C + wc) * FF + wu * F;
#pragma unroll
        for (int v = 0; v < F; ++v) p[v] = wacc[v];
        if (wc == 0 && wu == 0) dbpartial[slab * K + wk] = bacc;
    }
}

static size_t cb_lds_bytes(int F, int C, int K, int G, int Ho, int Wo) {
    const int FF = F * F;
    const int roww = 4 * ((Wo + 3) / 4) + 4;
    const size_t xplane = (size_t)(Ho + F - 1) * (roww + 4), hplane = (size_t)(Ho + 2 * F - 2) * roww;
    size_t fl = (size_t)K * FF * 4 + (size_t)K * C * F * 4 + ((K + 3) & ~3) +
                (((size_t)G * C * xplane + 3) & ~(size_t)3) + (size_t)G * K * hplane + 8;
    return fl * sizeof(float);
}

// returns the group size G (>0) if the LDS-resident backward applies, else 0
extern "C" int tn_convblock_supported(int C, int K, int f, int stride, int p, int Ho, int Wo) {
    if (f != 3 || stride != 1 || p != 2 || C < 1 || C > 4) return 0;
    if (K * C * f > 256 || K * C * f * f < 128) return 0;   // phase-3 threads; small nets keep convpool_bwd
    for (int G = 4; G >= 1; --G)
        if (G * K * C * f <= 1024 && cb_lds_bytes(f, C, K, G, Ho, Wo) <= 100 * 1024) return G;
    return 0;
}

template <int C>
static int launch_cb(tn_ctx* ctx, const float* x, const float* W, const float* b, const float* g,
                     float* dx, float* dW, float* db, CbGeom q, int act, float prm) {
    constexpr int F = 3;
    const size_t lds = cb_lds_bytes(F, C, q.K, q.G, q.Ho, q.Wo);
    const int ngroups = cdiv(q.N, q.G);
    int nblk = ctx->num_cus;
    if (nblk > ngroups) nblk = ngroups;
    const size_t KCFF = (size_t)q.K * C * F * F;
    const int nslab = nblk * q.G;
    float* partial;
    int rc = tn_scratch_get(ctx, (size_t)nslab * (KCFF + q.K) * sizeof(float), &partial);
    if (rc) return rc;
    float* dbpartial = partial + (size_t)nslab * KCFF;
    if (act == TN_ACT_LEAKY) {
        auto kern = convblock_bwd_lds<F, C, TN_ACT_LEAKY>;
        static size_t set_for = 0;      // the attribute call is not a stream op: do it once per size
        if (set_for < lds) {
            TN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            set_for = lds;
        }
        kern<<<nblk, CB_NT, lds, ctx->stream>>>(x, W, b, g, dx, partial, dbpartial, q, act, prm);
    } else {
        auto kern = convblock_bwd_lds<F, C, -1>;
        static size_t set_for = 0;
        if (set_for < lds) {
            TN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            set_for = lds;
        }
        kern<<<nblk, CB_NT, lds, ctx->stream>>>(x, W, b, g, dx, partial, dbpartial, q, act, prm);
    }
    TN_LAUNCH_CHECK();
    return tn_conv_wgrad_finish(ctx, partial, dbpartial, dW, db, nslab, q.K, C, F);
}

// Same contract as tn_convpool_bwd, but produces dx (may be NULL) directly instead of dz.
extern "C" int tn_convblock_bwd(tn_ctx* ctx, const float* x, const float* W, const float* b,
                                const float* g, float* dx, float* dW, float* db, int N, int C, int H,
                                int Wd, int K, int f, int pad_lo, int Ho, int Wo, int p, int Hp, int Wp,
                                int act, float act_param) {
    const int G = tn_convblock_supported(C, K, f, 1, p, Ho, Wo);
    TN_REQUIRE(G > 0, "tn_convblock_bwd: unsupported C=%d K=%d f=%d p=%d", C, K, f, p);
    if (tn_convblock_mfma_supported(C, K, f, 1, p, H, Wd, pad_lo, Ho, Wo, Hp, Wp))
        return tn_convblock_mfma_bwd(ctx, x, W, b, g, dx, dW, db, N, C, H, Wd, K, pad_lo, Ho, Wo, Hp, Wp,
                                     act, act_param);
    CbGeom q;
    q.N = N; q.H = H; q.Wd = Wd; q.K = K; q.pad = pad_lo; q.Ho = Ho; q.Wo = Wo; q.Hp = Hp; q.Wp = Wp;
    q.G = G;
    {
        static int dbg = -1;
        if (dbg < 0) {
            const char* e = getenv("TN_CB_DBG");
            dbg = e ? atoi(e) : 0;
        }
        q.dbg = dbg;
    }
    q.Hx = Ho + f - 1; q.Hh = Ho + 2 * f - 2;
    q.Wh = 4 * ((Wo + 3) / 4) + 4;
    q.Wx = q.Wh + 4;      // x rows 4 floats wider: de-phases the (c, u) b128 reads of phase 3
    switch (C) {
        case 1: return launch_cb<1>(ctx, x, W, b, g, dx, dW, db, q, act, act_param);
        case 2: return launch_cb<2>(ctx, x, W, b, g, dx, dW, db, q, act, act_param);
        case 3: return launch_cb<3>(ctx, x, W, b, g, dx, dW, db, q, act, act_param);
        default: return launch_cb<4>(ctx, x, W, b, g, dx, dW, db, q, act, act_param);
    }
}
