// Implicit-im2col convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), stride 1.
// Used when the reduction C*f*f is deep enough to feed MFMA tiles (C*f*f >= 32): the wide
// layers of cifar_like / wide6.  No im2col buffer ever exists: the B operand of the GEMM is
// gathered from x on the fly while it is staged into LDS.
//
//   forward : out[k][m] = act( b[k] + sum_kk A[k][kk] * Bm[kk][m] )
//             GEMM rows = filters k, GEMM cols = pixels m = (n, i, j) -- so a wave's 32 lanes
//             store 32 consecutive pixels of one map (coalesced NCHW stores);
//             A[k][kk] = W[k][c][u'][v'] in W's NATIVE order (kk = (c, u', v'), rows contiguous);
//             Bm[kk][m] = xpad[n, c, i + f-1-u', j + f-1-v']   (the flip of the true convolution,
//             theanet/layer/convpool.py:54-56, is applied on the gather side).
//   dgrad   : the same kernel on dz with the transposed + flipped weights Wt[c][k][u][v] =
//             W[k][c][f-1-u][f-1-v] (built by a tiny kernel) and padding f-1-pad_lo, with the
//             activation gradient of the layer below fused in the epilogue.
//   wgrad   : dWf[k][kk] = sum_m dz[k][m] * Bm[kk][m]: GEMM rows = filters, cols = (c,u,v),
//             reduction over pixels, split over 8 K-slabs (one per XCD), deterministic reduce.
//
// Tiling / pipeline as in gemm.hip: 64x64 block tile, BK = 16, LDS rows [row][k] (stride 20
// floats), two-tile register look-ahead, branch-free hot loop (all gather addresses are clamped,
// the zero padding of mode 'same' is a select issued after every load of the tile is in flight).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CM_BK 16
#define CM_LDK 20


struct ConvMG {
    const float* x;        // gathered operand source (N, C, H, Wd)
    const float* W;        // A operand: rows of length Kd = C*f*f (fwd/dgrad) ; dz for wgrad
    float* out;
    const float* bias;     // fwd
    const float* prev_a;   // dgrad epilogue: act'(prev_a) (same shape as out) or NULL
    int N, C, H, Wd;       // gathered tensor
    int K;                 // GEMM rows (filters / output maps)
    int f, pad;            // window, zero padding (lo)
    int Ho, Wo;            // spatial size of the pixel dimension m
    int Kd;                // C*f*f
    int M;                 // N*Ho*Wo
    int act;
    float prm;
    int a_vec;             // A rows 16-byte loadable
    int KT, MT;            // tiles along filters / pixels (fwd), filters / kk (wgrad)
    int S, mchunk;         // wgrad: pixel slabs
    float* ws;             // wgrad: [S][K][Kd] partial slabs
    float* dbws;           // wgrad: [S][K] partial bias gradients
};

// ---- pixel decode (once per thread: its pixels do not change across K-tiles) ----------------
struct Pix {
    int off;        // n*C*H*Wd + (i-pad)*Wd + (j-pad)   (may point outside for padded taps)
    int iy, jx;     // i - pad, j - pad
    bool ok;        // m < M
};
__device__ __forceinline__ Pix decode_pix(const ConvMG& g, int m) {
    Pix p;
    p.ok = m < g.M;
    const int mm = min(m, g.M - 1);
    const int HoWo = g.Ho * g.Wo;
    const int n = mm / HoWo, r = mm - n * HoWo;
    const int i = r / g.Wo, j = r - i * g.Wo;
    p.iy = i - g.pad;
    p.jx = j - g.pad;
    p.off = n * g.C * g.H * g.Wd + p.iy * g.Wd + p.jx;
    return p;
}

// Forward / dgrad kernel.  PADDED: zero padding present (mode 'same' / dgrad halo).
template <bool PADDED, bool DGRAD>
__global__ __launch_bounds__(256) void conv_mfma_fwd_kernel(ConvMG g) {
    __shared__ __attribute__((aligned(16))) float As[2][64][CM_LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][64][CM_LDK];
    // XCD-aware decode: all filter tiles of a pixel tile on one XCD (they share the gathered x)
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int mt = (idx / g.KT) * 8 + xcd, kt = idx % g.KT;
    if (mt >= g.MT) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int k0 = kt * 64, m0 = mt * 64;
    const int ntile = (g.Kd + CM_BK - 1) / CM_BK;
    const int ff = g.f * g.f, HW = g.H * g.Wd;

    // A staging (weights, KC): row = t>>2, kk = 4*(t&3)
    const int a_r = t >> 2, a_k = 4 * (t & 3);
    const int arow = min(k0 + a_r, g.K - 1);
    const float* pA = g.W + (size_t)arow * g.Kd + a_k;
    // B staging (gather): kk = t&15, 4 pixels m0 + 4*(t>>4) + e
    const int b_k = t & 15, b_r = 4 * (t >> 4);
    Pix px[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) px[e] = decode_pix(g, m0 + b_r + e);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float4 a0, a1;
    float b0[4], b1[4];
#define CM_GLOAD(RA, RB, TILE)                                                                   \
    {                                                                                            \
        const int kb_ = min((TILE), ntile - 1) * CM_BK;                                          \
        /* A: clamp the 4-group inside the row (values beyond Kd are multiplied by B == 0) */   \
        if (g.a_vec) { /* Kd % 4 == 0: a 4-group is wholly inside or wholly beyond the row */    \
            RA = *reinterpret_cast<const float4*>(pA + min(kb_, g.Kd - 4 - a_k));                \
        } else {                                                                                 \
            float w_[4];                                                                         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                        \
                w_[e] = g.W[(size_t)arow * g.Kd + min(kb_ + a_k + e, g.Kd - 1)];                 \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                        \
                w_[e] = (kb_ + a_k + e < g.Kd) ? w_[e] : 0.f;                                    \
            RA = make_float4(w_[0], w_[1], w_[2], w_[3]);                                        \
        }                                                                                        \
        /* B: tap of reduction index kk = kb_ + b_k */                                           \
        const int kk_ = min(kb_ + b_k, g.Kd - 1);                                                \
        const bool kok_ = kb_ + b_k < g.Kd;                                                      \
        const int c_ = kk_ / ff, r_ = kk_ - c_ * ff;                                             \
        const int du_ = g.f - 1 - r_ / g.f, dv_ = g.f - 1 - r_ % g.f;                            \
        const int tap_ = c_ * HW + du_ * g.Wd + dv_;                                             \
        bool ok_[4];                                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                          \
            ok_[e] = px[e].ok && kok_;                                                           \
            int off_ = px[e].off + tap_;                                                         \
            if (PADDED) {                                                                        \
                const bool in_ = ((unsigned)(px[e].iy + du_) < (unsigned)g.H) &&                 \
                                 ((unsigned)(px[e].jx + dv_) < (unsigned)g.Wd);                  \
                ok_[e] = ok_[e] && in_;                                                          \
                off_ = in_ ? off_ : 0;                                                           \
            }                                                                                    \
            RB[e] = g.x[off_];                                                                   \
        }                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) RB[e] = ok_[e] ? RB[e] : 0.f;              \
    }
#define CM_LSTORE(RA, RB, BUF)                                                                   \
    {                                                                                            \
        *reinterpret_cast<float4*>(&As[BUF][a_r][a_k]) = RA;                                     \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) Bs[BUF][b_r + e][b_k] = RB[e];             \
    }
    const int ar = wm * 32 + (lane & 31), br = wn * 32 + (lane & 31), hi = lane >> 5;
#define CM_COMPUTE(BUF)                                                                          \
    {                                                                                            \
        const float4* pa_ = reinterpret_cast<const float4*>(&As[BUF][ar][8 * hi]);               \
        const float4* pb_ = reinterpret_cast<const float4*>(&Bs[BUF][br][8 * hi]);               \
        const float4 al = pa_[0], au = pa_[1], bl = pb_[0], bu = pb_[1];                         \
        const float av[8] = {al.x, al.y, al.z, al.w, au.x, au.y, au.z, au.w};                    \
        const float bv[8] = {bl.x, bl.y, bl.z, bl.w, bu.x, bu.y, bu.z, bu.w};                    \
        _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_)                                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s_], bv[s_], acc, 0, 0, 0);            \
    }

    CM_GLOAD(a0, b0, 0);
    CM_GLOAD(a1, b1, 1);
    CM_LSTORE(a0, b0, 0);
    __syncthreads();
    CM_GLOAD(a0, b0, 2);
    int tile = 0;
    for (; tile + 1 < ntile; tile += 2) {
        CM_COMPUTE(0);
        CM_LSTORE(a1, b1, 1);
        __syncthreads();
        CM_GLOAD(a1, b1, tile + 3);
        CM_COMPUTE(1);
        CM_LSTORE(a0, b0, 0);
        __syncthreads();
        CM_GLOAD(a0, b0, tile + 4);
    }
    if (ntile & 1) CM_COMPUTE(0);
#undef CM_GLOAD
#undef CM_LSTORE
#undef CM_COMPUTE

    // ---- epilogue: lane <-> pixel (coalesced along a map), registers <-> filters -----------
    const int m = m0 + wn * 32 + (lane & 31);
    if (m < g.M) {
        const int HoWo = g.Ho * g.Wo;
        const int n = m / HoWo, p = m - n * HoWo;
        float pa[16];
        if (DGRAD && g.prev_a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = min(k0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, g.K - 1);
                pa[r] = g.prev_a[((size_t)n * g.K + k) * HoWo + p];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k < g.K) {
                float v = acc[r];
                if (DGRAD) {
                    if (g.prev_a) v *= tn_act_grad_from_out(pa[r], g.act, g.prm);
                } else {
                    v = tn_act_fwd(v + g.bias[k], g.act, g.prm);
                }
                g.out[((size_t)n * g.K + k) * HoWo + p] = v;
            }
        }
    }
}

// Wt[c][k][u][v] = W[k][c][f-1-u][f-1-v]
__global__ void conv_wt_kernel(const float* __restrict__ W, float* __restrict__ Wt, int K, int C, int f) {
    const int ff = f * f, total = K * C * ff;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int uv = t % ff, kc = t / ff, k = kc % K, c = kc / K;
    Wt[t] = W[((size_t)k * C + c) * ff + (ff - 1 - uv)];
}

// ---- wgrad: dWf[k][kk] = sum_m dz[k][m] * Bm[kk][m] ; rows = filters, cols = kk, K = pixels ---
template <bool PADDED>
__global__ __launch_bounds__(256) void conv_mfma_wgrad_kernel(ConvMG g) {
    __shared__ __attribute__((aligned(16))) float As[2][64][CM_LDK];   // [k][m]
    __shared__ __attribute__((aligned(16))) float Bs[2][64][CM_LDK];   // [kk][m]
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int per = g.KT * g.MT;
    const int z = (idx / per) * 8 + xcd;          // one pixel slab per XCD
    if (z >= g.S) return;
    const int rem = idx % per, kt = rem / g.MT, ct = rem % g.MT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int k0 = kt * 64, c0 = ct * 64;          // filter tile, kk tile
    const int mbeg = z * g.mchunk, mend = min(g.M, mbeg + g.mchunk);
    const int ntile = max((mend - mbeg + CM_BK - 1) / CM_BK, 1);
    const int ff = g.f * g.f, HW = g.H * g.Wd, HoWo = g.Ho * g.Wo;

    // A staging: dz[k][m]: thread -> (row k = t>>2, 4 pixels m = mb + 4*(t&3) + e)
    const int a_r = t >> 2, a_m = 4 * (t & 3);
    const int arow = min(k0 + a_r, g.K - 1);
    const bool arow_ok = k0 + a_r < g.K;
    // B staging: thread -> (row kk = t>>2, the same 4 pixels)
    const int b_r = t >> 2;
    const int kk = min(c0 + b_r, g.Kd - 1);
    const bool kk_ok = c0 + b_r < g.Kd;
    const int c_ = kk / ff, r_ = kk - c_ * ff;
    const int du = g.f - 1 - r_ / g.f, dv = g.f - 1 - r_ % g.f;      // kk = (c, u', v') native order
    const int tap = c_ * HW + du * g.Wd + dv;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a0[4], a1[4], b0[4], b1[4];
    // running pixel coordinates of this thread's 4 pixels; each load advances them by one tile
    // (16 pixels) with carries instead of divisions.  Loads are issued for tiles 0,1,2,3,... in
    // order; beyond the last tile they just re-read a clamped pixel (masked to zero).
    int pn[4], pi[4], pj[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m_ = min(mbeg + a_m + e, g.M - 1);
        pn[e] = m_ / HoWo;
        const int p_ = m_ - pn[e] * HoWo;
        pi[e] = p_ / g.Wo;
        pj[e] = p_ - pi[e] * g.Wo;
    }
    int next_m = mbeg + a_m;          // pixel index of element 0 of the NEXT load
    // dz rows of 4 pixels never straddle a plane when the plane size is a multiple of 4 (mbeg and the
    // tile step are multiples of 4 too) and the tensor is 16-byte aligned
    const bool avec = (HoWo & 3) == 0 && (g.mchunk & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.W) & 15) == 0);
    float asum = 0.f;                 // row sum of dz (bias gradient), first kk-tile only
#define CW_GLOAD(RA, RB, TILE)                                                                   \
    {                                                                                            \
        bool aok_[4], bok_[4];                                                                   \
        if (avec) {      /* 4 consecutive pixels of one (image, filter) plane: one 16-byte load */ \
            const float4 a4_ = *reinterpret_cast<const float4*>(                                 \
                g.W + ((size_t)min(pn[0], g.N - 1) * g.K + arow) * HoWo + pi[0] * g.Wo + pj[0]); \
            RA[0] = a4_.x; RA[1] = a4_.y; RA[2] = a4_.z; RA[3] = a4_.w;                          \
        }                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                          \
            const bool mok_ = next_m + e < mend;                                                 \
            const int n_ = min(pn[e], g.N - 1), i_ = pi[e], j_ = pj[e];                          \
            const int p_ = i_ * g.Wo + j_;                                                       \
            if (!avec) RA[e] = g.W[((size_t)n_ * g.K + arow) * HoWo + p_];                       \
            aok_[e] = mok_ && arow_ok;                                                           \
            int off_ = n_ * g.C * HW + (i_ - g.pad) * g.Wd + (j_ - g.pad) + tap;                 \
            bok_[e] = mok_ && kk_ok;                                                             \
            if (PADDED) {                                                                        \
                const bool in_ = ((unsigned)(i_ - g.pad + du) < (unsigned)g.H) &&                \
                                 ((unsigned)(j_ - g.pad + dv) < (unsigned)g.Wd);                 \
                bok_[e] = bok_[e] && in_;                                                        \
                off_ = in_ ? off_ : 0;                                                           \
            }                                                                                    \
            RB[e] = g.x[off_];                                                                   \
        }                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                          \
            RA[e] = aok_[e] ? RA[e] : 0.f;                                                       \
            RB[e] = bok_[e] ? RB[e] : 0.f;                                                       \
            asum += RA[e];                                                                       \
        }                                                                                        \
        next_m += CM_BK;                                                                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {   /* advance by 16 pixels with carries */ \
            pj[e] += CM_BK;                                                                      \
            while (pj[e] >= g.Wo) { pj[e] -= g.Wo; ++pi[e]; }                                    \
            while (pi[e] >= g.Ho) { pi[e] -= g.Ho; ++pn[e]; }                                    \
        }                                                                                        \
    }
#define CW_LSTORE(RA, RB, BUF)                                                                   \
    {                                                                                            \
        *reinterpret_cast<float4*>(&As[BUF][a_r][a_m]) = make_float4(RA[0], RA[1], RA[2], RA[3]); \
        *reinterpret_cast<float4*>(&Bs[BUF][b_r][a_m]) = make_float4(RB[0], RB[1], RB[2], RB[3]); \
    }
    const int ar = wm * 32 + (lane & 31), br = wn * 32 + (lane & 31), hi = lane >> 5;
#define CW_COMPUTE(BUF)                                                                          \
    {                                                                                            \
        const float4* pa_ = reinterpret_cast<const float4*>(&As[BUF][ar][8 * hi]);               \
        const float4* pb_ = reinterpret_cast<const float4*>(&Bs[BUF][br][8 * hi]);               \
        const float4 al = pa_[0], au = pa_[1], bl = pb_[0], bu = pb_[1];                         \
        const float av[8] = {al.x, al.y, al.z, al.w, au.x, au.y, au.z, au.w};                    \
        const float bv[8] = {bl.x, bl.y, bl.z, bl.w, bu.x, bu.y, bu.z, bu.w};                    \
        _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_)                                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s_], bv[s_], acc, 0, 0, 0);            \
    }
    if (mbeg < mend) {
        CW_GLOAD(a0, b0, 0);
        CW_GLOAD(a1, b1, 1);
        CW_LSTORE(a0, b0, 0);
        __syncthreads();
        CW_GLOAD(a0, b0, 2);
        int tile = 0;
        for (; tile + 1 < ntile; tile += 2) {
            CW_COMPUTE(0);
            CW_LSTORE(a1, b1, 1);
            __syncthreads();
            CW_GLOAD(a1, b1, tile + 3);
            CW_COMPUTE(1);
            CW_LSTORE(a0, b0, 0);
            __syncthreads();
            CW_GLOAD(a0, b0, tile + 4);
        }
        if (ntile & 1) CW_COMPUTE(0);
    }
#undef CW_GLOAD
#undef CW_LSTORE
#undef CW_COMPUTE
    // bias gradient partial of this slab: row sums of dz (4 lanes per row)
    if (ct == 0) {
        asum += __shfl_xor(asum, 1, 64);
        asum += __shfl_xor(asum, 2, 64);
        if ((t & 3) == 0 && arow_ok) g.dbws[(size_t)z * g.K + k0 + a_r] = asum;
    }
    // slab z: ws[z][k][kk]
    const int col = c0 + wn * 32 + (lane & 31);
    if (col < g.Kd) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k < g.K) g.ws[((size_t)z * g.K + k) * g.Kd + col] = acc[r];
        }
    }
}

// dW[k][c][u][v] = sum_z ws[z][k][(c,u,v)]  (kk is already in W's native order: no flip)
__global__ __launch_bounds__(256) void conv_mfma_wgrad_reduce(const float* __restrict__ ws,
                                                             float* __restrict__ dW, int n, int S,
                                                             const float* __restrict__ dbws,
                                                             float* __restrict__ db, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        float s = 0.f;
#pragma unroll 8
        for (int z = 0; z < S; ++z) s += ws[(size_t)z * n + i];
        dW[i] = s;
    }
    if (i < K) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += dbws[(size_t)z * K + i];
        db[i] = s;
    }
}

// conv_tile.hip: the LDS-resident-tile kernels for 3x3 windows
int tn_conv_tile_ok(const float* x, int N, int C, int H, int Wd, int K, int f, int pad, int Ho, int Wo);
int tn_conv_tile_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int N, int C,
                     int H, int Wd, int K, int pad, int Ho, int Wo, int act, float prm);
int tn_conv_tile_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int N, int C, int H,
                       int Wd, int K, int pad, int Ho, int Wo, const float* prev_a, int act, float prm);

int tn_conv_tile_wgrad_ok(tn_ctx* ctx, const float* x, const float* dz, int N, int C, int H, int Wd, int K,
                          int f, int pad, int Ho, int Wo);
int tn_conv_tile_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N, int C,
                       int H, int Wd, int K);

static int vecA(const void* p, int kd) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (kd % 4 == 0) && kd >= 4; }

// 1 if the MFMA path applies (stride 1, deep enough reduction, enough filters)
extern "C" int tn_conv_mfma_supported(int C, int K, int f, int stride) {
    return stride == 1 && C * f * f >= 32 && K >= 16;
}

int tn_conv_mfma_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int N,
                     int C, int H, int Wd, int K, int f, int pad, int Ho, int Wo, int act, float prm) {
    if (tn_conv_tile_ok(x, N, C, H, Wd, K, f, pad, Ho, Wo))
        return tn_conv_tile_fwd(ctx, x, W, b, a, N, C, H, Wd, K, pad, Ho, Wo, act, prm);
    ConvMG g{};
    g.x = x; g.W = W; g.out = a; g.bias = b;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K; g.f = f; g.pad = pad; g.Ho = Ho; g.Wo = Wo;
    g.Kd = C * f * f; g.M = N * Ho * Wo; g.act = act; g.prm = prm; g.a_vec = vecA(W, g.Kd);
    g.KT = cdiv(K, 64); g.MT = cdiv(g.M, 64);
    TN_REQUIRE((long long)N * C * H * Wd < (1ll << 31) && (long long)N * K * Ho * Wo < (1ll << 31),
               "tn_conv_mfma_fwd: tensor too large for 32-bit offsets");
    const int grid = 8 * cdiv(g.MT, 8) * g.KT;
    if (pad > 0 || Ho + f - 1 > H || Wo + f - 1 > Wd)
        conv_mfma_fwd_kernel<true, false><<<grid, 256, 0, ctx->stream>>>(g);
    else
        conv_mfma_fwd_kernel<false, false><<<grid, 256, 0, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_conv_mfma_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int N, int C, int H,
                       int Wd, int K, int f, int pad, int Ho, int Wo, const float* prev_a, int act,
                       float prm) {
    // dx = conv_fwd(dz, Wt) with padding f-1-pad: gathered tensor = dz (N,K,Ho,Wo), rows = C maps
    if (tn_conv_tile_ok(dz, N, K, Ho, Wo, C, f, f - 1 - pad, H, Wd))
        return tn_conv_tile_dgrad(ctx, dz, W, dx, N, C, H, Wd, K, pad, Ho, Wo, prev_a, act, prm);
    const size_t wt_bytes = (size_t)K * C * f * f * sizeof(float);
    float* Wt;
    int rc = tn_scratch_get(ctx, wt_bytes, &Wt);
    if (rc) return rc;
    conv_wt_kernel<<<cdiv((long long)K * C * f * f, 256), 256, 0, ctx->stream>>>(W, Wt, K, C, f);
    TN_LAUNCH_CHECK();
    ConvMG g{};
    g.x = dz; g.W = Wt; g.out = dx; g.prev_a = prev_a;
    g.N = N; g.C = K; g.H = Ho; g.Wd = Wo; g.K = C; g.f = f; g.pad = f - 1 - pad; g.Ho = H; g.Wo = Wd;
    g.Kd = K * f * f; g.M = N * H * Wd; g.act = act; g.prm = prm; g.a_vec = vecA(Wt, g.Kd);
    g.KT = cdiv(C, 64); g.MT = cdiv(g.M, 64);
    TN_REQUIRE((long long)N * C * H * Wd < (1ll << 31) && (long long)N * K * Ho * Wo < (1ll << 31),
               "tn_conv_mfma_dgrad: tensor too large for 32-bit offsets");
    const int grid = 8 * cdiv(g.MT, 8) * g.KT;
    conv_mfma_fwd_kernel<true, true><<<grid, 256, 0, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_conv_mfma_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N,
                       int C, int H, int Wd, int K, int f, int pad, int Ho, int Wo) {
    if (tn_conv_tile_wgrad_ok(ctx, x, dz, N, C, H, Wd, K, f, pad, Ho, Wo))
        return tn_conv_tile_wgrad(ctx, x, dz, dW, db, N, C, H, Wd, K);
    ConvMG g{};
    g.x = x; g.W = dz;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K; g.f = f; g.pad = pad; g.Ho = Ho; g.Wo = Wo;
    g.Kd = C * f * f; g.M = N * Ho * Wo;
    g.KT = cdiv(K, 64); g.MT = cdiv(g.Kd, 64);
    TN_REQUIRE((long long)N * C * H * Wd < (1ll << 31) && (long long)N * K * Ho * Wo < (1ll << 31),
               "tn_conv_mfma_wgrad: tensor too large for 32-bit offsets");
    // pixel slabs: a multiple of 8 (one or more per XCD), enough blocks to fill the chip
    int S = 8;
    while (S < 64 && (long long)S * g.KT * g.MT < 2 * ctx->num_cus && g.M / (2 * S) >= 8 * CM_BK) S *= 2;
    g.S = S;
    g.mchunk = cdiv(cdiv(g.M, S), CM_BK) * CM_BK;
    const size_t n = (size_t)K * g.Kd;
    int rc = tn_scratch_get(ctx, ((size_t)S * n + (size_t)S * K) * sizeof(float), &g.ws);
    if (rc) return rc;
    g.dbws = g.ws + (size_t)S * n;
    const int grid = S * g.KT * g.MT;
    if (pad > 0 || Ho + f - 1 > H || Wo + f - 1 > Wd)
        conv_mfma_wgrad_kernel<true><<<grid, 256, 0, ctx->stream>>>(g);
    else
        conv_mfma_wgrad_kernel<false><<<grid, 256, 0, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    conv_mfma_wgrad_reduce<<<cdiv(n, 256), 256, 0, ctx->stream>>>(g.ws, dW, (int)n, S, g.dbws, db, K);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
