// Fully-connected layers with a handful of outputs (n_out <= 16, e.g. mnist.prms' 500 -> 10
// softmax layer: theanet/layer/outlayers.py:87-95, hidden.py:40-43).  Too skinny for the tiled
// GEMM: every op is one pass over the (B, n_in) activation matrix, so the kernels are built
// around 16-byte row-contiguous accesses of that matrix, all issued before the first use:
//   fwd    out = act(x W + b) * mask     16x16x4 f32 MFMA, M = 16 rows, N = outputs, 4-way split of K
//   wgrad  dW = x^T dz, db = 1^T dz      16x16x4 f32 MFMA, M = input features (+ a ones column),
//                                        N = outputs, reduction over rows; slabs summed by reduce.hip
//   dgrad  dx = (dz W^T) act'(a) mask    VALU: thread = 4 input features, dz rows as scalar loads
// Requires n_in % 4 == 0 and 16-byte aligned rows (tn_fc_skinny_ok); anything else keeps the
// scalar kernels in gemm.hip.
#include <cstdlib>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SK_MAX 16

__device__ __forceinline__ f32x4 sk_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// optional softmax / NLL tail of the forward kernel (outlayers.py:50-51, :87-95)
struct SkSoftmax {
    const int32_t* y;          // labels (may be NULL: log-probabilities and argmax only)
    int64_t y_row0;
    const int64_t* d_row0;
    float* logprob;            // NULL: plain fully-connected forward
    float* rowloss;
    int32_t* pred;
    float* rowp;
    float* dz;
    float inv_batch;
};

template <int CTRL>
__device__ __forceinline__ float sk_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// all-reduce over the 16 lanes of a DPP row: quad xor 1, quad xor 2, half mirror, mirror
__device__ __forceinline__ float sk_row_max(float v) {
    v = fmaxf(v, sk_dpp<0xB1>(v));
    v = fmaxf(v, sk_dpp<0x4E>(v));
    v = fmaxf(v, sk_dpp<0x141>(v));
    return fmaxf(v, sk_dpp<0x140>(v));
}
__device__ __forceinline__ float sk_row_min(float v) {
    v = fminf(v, sk_dpp<0xB1>(v));
    v = fminf(v, sk_dpp<0x4E>(v));
    v = fminf(v, sk_dpp<0x141>(v));
    return fminf(v, sk_dpp<0x140>(v));
}
__device__ __forceinline__ float sk_row_sum(float v) {
    v += sk_dpp<0xB1>(v);
    v += sk_dpp<0x4E>(v);
    v += sk_dpp<0x141>(v);
    return v + sk_dpp<0x140>(v);
}

// ---- forward --------------------------------------------------------------------------------
// block = 16 rows, 4 waves; wave w owns the 16-wide k chunks w, w+4, ...  Lane (lo, qd) loads
// x[row lo][16c + 4qd .. +3] as one float4: component e is the A operand of reduction step
// (c, e) with k = 16c + 4qd + e, and the B operand W[k][n = lo] follows the same enumeration.
#define SKF_CH 8
__global__ __launch_bounds__(256) void fc_skinny_fwd_mfma(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    float* __restrict__ a, int B, int n_in, int n_out, int act, float prm,
    const uint8_t* __restrict__ mask, SkSoftmax sm) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lo = lane & 15, qd = lane >> 4;
    const float* xr = x + (size_t)min(blockIdx.x * 16 + lo, B - 1) * n_in;
    const int nch = (n_in + 15) >> 4;
    const int nc = min(lo, n_out - 1);
    const bool nlive = lo < n_out;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = w; c0 < nch; c0 += 4 * SKF_CH) {
        float4 xv[SKF_CH];
        float wv[SKF_CH][4];
#pragma unroll
        for (int i = 0; i < SKF_CH; ++i) {
            const int k = 16 * (c0 + 4 * i) + 4 * qd;
            const int kc = min(k, n_in - 4);
            xv[i] = *reinterpret_cast<const float4*>(xr + kc);
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[i][e] = W[(size_t)(kc + e) * n_out + nc];
        }
#pragma unroll
        for (int i = 0; i < SKF_CH; ++i) {
            const bool live = nlive && (16 * (c0 + 4 * i) + 4 * qd < n_in);     // n_in % 4 == 0
            acc = sk_mfma(xv[i].x, live ? wv[i][0] : 0.f, acc);
            acc = sk_mfma(xv[i].y, live ? wv[i][1] : 0.f, acc);
            acc = sk_mfma(xv[i].z, live ? wv[i][2] : 0.f, acc);
            acc = sk_mfma(xv[i].w, live ? wv[i][3] : 0.f, acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][r * 64 + lane] = acc[r];
    __syncthreads();
    if (w != 0) return;
    const float bias = (b && nlive) ? b[nc] : 0.f;
    if (sm.logprob == nullptr) {
        if (!nlive) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = blockIdx.x * 16 + 4 * qd + r;        // accumulator row of register r
            const float s = ((red[0][r * 64 + lane] + red[1][r * 64 + lane]) + red[2][r * 64 + lane]) +
                            red[3][r * 64 + lane];
            if (row < B) {
                const size_t o = (size_t)row * n_out + lo;
                float v = tn_act_fwd(s + bias, act, prm);
                if (mask) v *= (float)mask[o];
                a[o] = v;
            }
        }
        return;
    }
    // softmax / NLL tail: a row's logits sit in the 16 lanes of one DPP row (lane lo = class)
    const int64_t yoff = sm.y_row0 + (sm.d_row0 ? *sm.d_row0 : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = blockIdx.x * 16 + 4 * qd + r;
        const int rowc = min(row, B - 1);
        const float z = ((red[0][r * 64 + lane] + red[1][r * 64 + lane]) + red[2][r * 64 + lane]) +
                        red[3][r * 64 + lane] + bias;
        const float zv = nlive ? z : -INFINITY;
        const float m = sk_row_max(zv);
        // first maximal class (numpy argmax)
        const float am = sk_row_min((nlive && zv == m) ? (float)lo : 1e9f);
        const float se = sk_row_sum(nlive ? expf(zv - m) : 0.f);
        const float lp = zv - m - logf(se);
        const int label = sm.y ? sm.y[yoff + rowc] : -1;
        if (nlive && row < B) {
            const size_t o = (size_t)row * n_out + lo;
            if (a) a[o] = z;
            sm.logprob[o] = lp;
            if (sm.dz) sm.dz[o] = (expf(lp) - (lo == label ? 1.f : 0.f)) * sm.inv_batch;
            if (lo == label) {
                if (sm.rowloss) sm.rowloss[row] = -lp;
                if (sm.rowp) sm.rowp[row] = expf(lp);
            }
            if (lo == 0 && sm.pred) sm.pred[row] = (int)am;
        }
    }
}

// ---- wgrad ----------------------------------------------------------------------------------
// grid (T, RC): T = group of 64 input features (feature n_in is a virtual column of ones -> db),
// RC = chunk of 128 rows, 32 per wave.  Lane (lo, qd) loads x[row 4s+qd][64T + 4lo .. +3]; its
// component e feeds accumulator e, whose M rows are the features 64T + 4i + e.
#define SKW_ST 8
__device__ __forceinline__ void sk_wgrad_body(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ slab, int B,
    int n_in, int n_out, int bx, int by, float (*red)[1024]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lo = lane & 15, qd = lane >> 4;
    const int k0 = 64 * bx + 4 * lo;
    const int kc = min(k0, n_in - 4);
    const int rowbase = 128 * by + 32 * w;
    const int nc = min(lo, n_out - 1);
    float4 xv[SKW_ST];
    float dv[SKW_ST];
#pragma unroll
    for (int s = 0; s < SKW_ST; ++s) {
        const int row = rowbase + 4 * s + qd;
        const int rc = min(row, B - 1);
        xv[s] = *reinterpret_cast<const float4*>(x + (size_t)rc * n_in + kc);
        dv[s] = dz[(size_t)rc * n_out + nc];
        dv[s] = (row < B && lo < n_out) ? dv[s] : 0.f;
    }
    // n_in % 4 == 0: the lane's 4 columns are all real, or (ones, 0, 0, 0), or beyond
    const bool real = k0 < n_in, ones = k0 == n_in;
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SKW_ST; ++s) {
        acc[0] = sk_mfma(real ? xv[s].x : (ones ? 1.f : 0.f), dv[s], acc[0]);
        acc[1] = sk_mfma(real ? xv[s].y : 0.f, dv[s], acc[1]);
        acc[2] = sk_mfma(real ? xv[s].z : 0.f, dv[s], acc[2]);
        acc[3] = sk_mfma(real ? xv[s].w : 0.f, dv[s], acc[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][(e * 4 + r) * 64 + lane] = acc[e][r];
    __syncthreads();
    // accumulator e, register r, lane (n = lo, qd): feature 64T + 4(4qd + r) + e
    const int per = (n_in + 1) * n_out;
    for (int t = threadIdx.x; t < 1024; t += 256) {
        const int l = t & 63, er = t >> 6, e = er >> 2, r = er & 3;
        const int n = l & 15, k = 64 * bx + 4 * (4 * (l >> 4) + r) + e;
        if (n < n_out && k <= n_in)
            slab[(size_t)by * per + (size_t)k * n_out + n] =
                ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    }
}

__global__ __launch_bounds__(256) void fc_skinny_wgrad_mfma(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ slab, int B,
    int n_in, int n_out) {
    __shared__ float red[4][1024];
    sk_wgrad_body(x, dz, slab, B, n_in, n_out, blockIdx.x, blockIdx.y, red);
}


// ---- dgrad ----------------------------------------------------------------------------------
// thread = 4 consecutive input features for SKD_ROWS rows; the W rows stay in registers, every
// prev_a / mask access is one 16-byte / 4-byte load issued up front, the dz rows are wave-uniform.
#define SKD_ROWS 8
template <int NOUT>
__device__ __forceinline__ void sk_dgrad_body(
    const float* __restrict__ dz, const float* __restrict__ W, float* __restrict__ dx, int B, int n_in,
    const float* __restrict__ prev_a, int act, float prm, const uint8_t* __restrict__ mask, int q,
    int row0) {
    const int nq = n_in >> 2;
    const bool live = q < nq;
    const int k = 4 * min(q, nq - 1);
    // the 4 W rows of this thread are 4*NOUT consecutive floats: NOUT 16-byte loads
    float wf[4 * NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(W + (size_t)k * NOUT + 4 * i);
        wf[4 * i] = v.x; wf[4 * i + 1] = v.y; wf[4 * i + 2] = v.z; wf[4 * i + 3] = v.w;
    }
    float4 pa[SKD_ROWS];
    uint32_t pm[SKD_ROWS];
#pragma unroll
    for (int r = 0; r < SKD_ROWS; ++r) {
        const size_t o = (size_t)min(row0 + r, B - 1) * n_in + k;
        pa[r] = prev_a ? *reinterpret_cast<const float4*>(prev_a + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        pm[r] = mask ? *reinterpret_cast<const uint32_t*>(mask + o) : 0x01010101u;
    }
#pragma unroll
    for (int r = 0; r < SKD_ROWS; ++r) {
        const int row = row0 + r;
        const float* dzr = dz + (size_t)min(row, B - 1) * NOUT;     // wave-uniform -> scalar loads
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            const float d = dzr[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = fmaf(d, wf[e * NOUT + n], s[e]);
        }
        if (prev_a) {
            float4 gq = make_float4(1.f, 1.f, 1.f, 1.f);
            tn_act_grad4(gq, pa[r], act, prm);
            s[0] *= gq.x; s[1] *= gq.y; s[2] *= gq.z; s[3] *= gq.w;
        }
        s[0] *= (float)(pm[r] & 0xffu);
        s[1] *= (float)((pm[r] >> 8) & 0xffu);
        s[2] *= (float)((pm[r] >> 16) & 0xffu);
        s[3] *= (float)(pm[r] >> 24);
        if (live && row < B)
            *reinterpret_cast<float4*>(dx + (size_t)row * n_in + k) = make_float4(s[0], s[1], s[2], s[3]);
    }
}

template <int NOUT>
__global__ __launch_bounds__(128) void fc_skinny_dgrad_v4(
    const float* __restrict__ dz, const float* __restrict__ W, float* __restrict__ dx, int B, int n_in,
    const float* __restrict__ prev_a, int act, float prm, const uint8_t* __restrict__ mask) {
    sk_dgrad_body<NOUT>(dz, W, dx, B, n_in, prev_a, act, prm, mask, blockIdx.x * 128 + threadIdx.x,
                        blockIdx.y * SKD_ROWS);
}

// weight gradient and input gradient of one layer in ONE launch: they only share dz, so the blocks
// of the first nW ids run the wgrad body and the rest two 128-thread dgrad bodies each
template <int NOUT>
__global__ __launch_bounds__(256) void fc_skinny_bwd_pair(
    const float* __restrict__ x, const float* __restrict__ dz, const float* __restrict__ W,
    float* __restrict__ slab, float* __restrict__ dx, int B, int n_in, const float* __restrict__ prev_a,
    int act, float prm, const uint8_t* __restrict__ mask, int gxW, int nW, int gxD) {
    __shared__ float red[4][1024];
    const int bid = blockIdx.x;
    if (bid < nW) {
        sk_wgrad_body(x, dz, slab, B, n_in, NOUT, bid % gxW, bid / gxW, red);
        return;
    }
    const int d = 2 * (bid - nW) + (threadIdx.x >> 7);      // 128-thread dgrad block id
    const int by = d / gxD, bx = d - by * gxD;
    if (by * SKD_ROWS >= B) return;
    sk_dgrad_body<NOUT>(dz, W, dx, B, n_in, prev_a, act, prm, mask, bx * 128 + (threadIdx.x & 127),
                        by * SKD_ROWS);
}

// ---- the whole SoftmaxLayer training step in one launch -----------------------------------------
// block = RB rows.  (1) logits on the matrix core + the softmax / NLL tail, twice 16 rows, dlogits
// kept in LDS; (2) input gradient of the 32 rows (thread = 4 features, 16 rows); (3) the rows'
// contribution to the weight gradient on the matrix core: wave w owns the 64-feature groups
// w, w+4, ... and writes them straight into this block's slab.  h is read from HBM once (the two
// later passes hit L2), dh is written once, and two kernel boundaries disappear.
// RB = rows per block: 16, or 4 for short batches (a 512-image shard of the 8-GPU run, wide6's 128 images): the block's
// time is a chain of latencies that does not shrink with its rows, so short batches are spread over 4x the blocks
// (rows 4.. of the 16-row logits tile are duplicates and go nowhere).
__host__ __device__ inline int sk_train_rb(int B) { return B < 2048 ? 4 : 16; }
template <int NOUT, int RB>
__global__ __launch_bounds__(256) void fc_skinny_softmax_train(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    float* __restrict__ logits, SkSoftmax sm, float* __restrict__ slab, float* __restrict__ dx, int B,
    int n_in, int act, float prm, const uint8_t* __restrict__ mask, int fuse_act) {
    __shared__ float red[4][256];
    __shared__ __attribute__((aligned(16))) float sdz[RB][16];
    constexpr int NTILE = RB >= 16 ? RB / 16 : 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lo = lane & 15, qd = lane >> 4;
    const int base = blockIdx.x * RB;
    const int nch = (n_in + 15) >> 4;
    const int nc = min(lo, NOUT - 1);
    const bool nlive = lo < NOUT;
    const int64_t yoff = sm.y_row0 + (sm.d_row0 ? *sm.d_row0 : 0);
    const float bias = (b && nlive) ? b[nc] : 0.f;
    // ---- (1) forward ---------------------------------------------------------------------------
    for (int tile = 0; tile < NTILE; ++tile) {
        const int rb = base + 16 * tile;
        const float* xr = x + (size_t)min(rb + min(lo, RB - 1), B - 1) * n_in;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = w; c0 < nch; c0 += 4 * SKF_CH) {
            float4 xv[SKF_CH];
            float wv[SKF_CH][4];
#pragma unroll
            for (int i = 0; i < SKF_CH; ++i) {
                const int kc = min(16 * (c0 + 4 * i) + 4 * qd, n_in - 4);
                xv[i] = *reinterpret_cast<const float4*>(xr + kc);
#pragma unroll
                for (int e = 0; e < 4; ++e) wv[i][e] = W[(size_t)(kc + e) * NOUT + nc];
            }
#pragma unroll
            for (int i = 0; i < SKF_CH; ++i) {
                const bool live = nlive && (16 * (c0 + 4 * i) + 4 * qd < n_in);
                acc = sk_mfma(xv[i].x, live ? wv[i][0] : 0.f, acc);
                acc = sk_mfma(xv[i].y, live ? wv[i][1] : 0.f, acc);
                acc = sk_mfma(xv[i].z, live ? wv[i][2] : 0.f, acc);
                acc = sk_mfma(xv[i].w, live ? wv[i][3] : 0.f, acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][r * 64 + lane] = acc[r];
        __syncthreads();
        {   // every wave sees all partial sums in LDS: wave w finishes accumulator row r = w of each 4-row group, so
            // the four dependent exp / log / row-reduction chains run side by side (1.75 -> 0.6 us of the block)
            {
                const int r = w;
                const int rl = 16 * tile + 4 * qd + r;
                const int row = rl < RB ? base + rl : B;         // rows of the tile beyond the block's: nothing is stored
                const int rowc = min(row, B - 1);
                const float z = ((red[0][r * 64 + lane] + red[1][r * 64 + lane]) + red[2][r * 64 + lane]) +
                                red[3][r * 64 + lane] + bias;
                const float zv = nlive ? z : -INFINITY;
                const float m = sk_row_max(zv);
                const float am = sk_row_min((nlive && zv == m) ? (float)lo : 1e9f);
                const float se = sk_row_sum(nlive ? expf(zv - m) : 0.f);
                const float lp = zv - m - logf(se);
                const int label = sm.y[yoff + rowc];
                const float d = (nlive && row < B) ? (expf(lp) - (lo == label ? 1.f : 0.f)) * sm.inv_batch : 0.f;
                if (rl < RB) sdz[rl][lo] = d;
                if (nlive && row < B) {
                    const size_t o = (size_t)row * NOUT + lo;
                    if (logits) logits[o] = z;
                    sm.logprob[o] = lp;
                    if (sm.dz) sm.dz[o] = d;
                    if (lo == label) {
                        if (sm.rowloss) sm.rowloss[row] = -lp;
                        if (sm.rowp) sm.rowp[row] = expf(lp);
                    }
                    if (lo == 0 && sm.pred) sm.pred[row] = (int)am;
                }
            }
        }
        __syncthreads();
    }
    // ---- (2) input gradient: dx = (dz W^T) * act'(x) * mask  (x IS the output of the layer below) ----
    {
        const int nq = n_in >> 2, half = threadIdx.x >> 7;
        constexpr int GR = RB / 2 < 8 ? RB / 2 : 8, NG = (RB / 2) / GR;     // rows per half in groups of GR
        for (int q0 = 0; q0 < nq; q0 += 128) {
            const int q = q0 + (threadIdx.x & 127);
            const bool live = q < nq;
            const int k = 4 * min(q, nq - 1);
            float wf[4 * NOUT];
#pragma unroll
            for (int i = 0; i < NOUT; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(W + (size_t)k * NOUT + 4 * i);
                wf[4 * i] = v.x; wf[4 * i + 1] = v.y; wf[4 * i + 2] = v.z; wf[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float4 pa[GR];
                uint32_t pm[GR];
#pragma unroll
                for (int r = 0; r < GR; ++r) {
                    const size_t o = (size_t)min(base + (RB / 2) * half + GR * g + r, B - 1) * n_in + k;
                    pa[r] = fuse_act ? *reinterpret_cast<const float4*>(x + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                    pm[r] = mask ? *reinterpret_cast<const uint32_t*>(mask + o) : 0x01010101u;
                }
                // act'(x) of the 8 rows behind ONE test of the activation kind (per element the whole switch of
                // tn_act_grad_from_out was paid 32 times: 401 scalar branches, 5.5 of the block's 12.6 us)
                float4 gp[GR];
#pragma unroll
                for (int r = 0; r < GR; ++r) gp[r] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (fuse_act) {
#pragma unroll
                    for (int r = 0; r < GR; ++r) tn_act_grad4(gp[r], pa[r], act, prm);
                }
#pragma unroll
                for (int r = 0; r < GR; ++r) {
                    const int rl = (RB / 2) * half + GR * g + r, row = base + rl;
                    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int n = 0; n < NOUT; ++n) {
                        const float d = sdz[rl][n];            // wave-uniform LDS broadcast
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[e] = fmaf(d, wf[e * NOUT + n], s[e]);
                    }
                    s[0] *= gp[r].x; s[1] *= gp[r].y; s[2] *= gp[r].z; s[3] *= gp[r].w;
                    s[0] *= (float)(pm[r] & 0xffu);
                    s[1] *= (float)((pm[r] >> 8) & 0xffu);
                    s[2] *= (float)((pm[r] >> 16) & 0xffu);
                    s[3] *= (float)(pm[r] >> 24);
                    if (live && row < B)
                        *reinterpret_cast<float4*>(dx + (size_t)row * n_in + k) = make_float4(s[0], s[1], s[2], s[3]);
                }
            }
        }
    }
    // ---- (3) weight gradient of the block's 32 rows -> slab[blockIdx.x] ------------------------------
    {
        const int per = (n_in + 1) * NOUT, ngrp = (n_in + 1 + 63) >> 6;
        for (int T = w; T < ngrp; T += 4) {
            const int k0 = 64 * T + 4 * lo;
            const int kc = min(k0, n_in - 4);
            const bool real = k0 < n_in, ones = k0 == n_in;
            f32x4 acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            float4 xv[RB / 4];
#pragma unroll
            for (int s2 = 0; s2 < RB / 4; ++s2)
                xv[s2] = *reinterpret_cast<const float4*>(x + (size_t)min(base + 4 * s2 + qd, B - 1) * n_in + kc);
#pragma unroll
            for (int s2 = 0; s2 < RB / 4; ++s2) {
                const float dv = sdz[4 * s2 + qd][lo];         // 0 for rows >= B and classes >= NOUT
                acc[0] = sk_mfma(real ? xv[s2].x : (ones ? 1.f : 0.f), dv, acc[0]);
                acc[1] = sk_mfma(real ? xv[s2].y : 0.f, dv, acc[1]);
                acc[2] = sk_mfma(real ? xv[s2].z : 0.f, dv, acc[2]);
                acc[3] = sk_mfma(real ? xv[s2].w : 0.f, dv, acc[3]);
            }
            if (nlive) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 64 * T + 4 * (4 * qd + r) + e;
                        if (k <= n_in) slab[(size_t)blockIdx.x * per + (size_t)k * NOUT + lo] = acc[e][r];
                    }
            }
        }
    }
}

// ---- host side (called from the tn_fc_* entry points in gemm.hip) -------------------------------
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

bool tn_fc_skinny_ok(int n_in, int n_out, const void* p0, const void* p1, const void* p2) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("TN_FC_SKINNY");
        enabled = e ? atoi(e) : 1;
    }
    return enabled && n_out <= SK_MAX && n_in % 4 == 0 && n_in >= 4 && al16(p0) && (!p1 || al16(p1)) &&
           (!p2 || ((uintptr_t)p2 & 3) == 0);
}

int tn_fc_skinny_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B,
                     int n_in, int n_out, int act, float prm, const uint8_t* mask) {
    SkSoftmax sm{};
    fc_skinny_fwd_mfma<<<cdiv(B, 16), 256, 0, ctx->stream>>>(x, W, b, a, B, n_in, n_out, act, prm, mask, sm);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// logits = x W + b and the whole softmax / NLL row computation in one launch
int tn_fc_skinny_softmax(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits,
                         int B, int n_in, int n_out, const int32_t* y, int64_t y_row0,
                         const int64_t* d_row0, float* logprob, float* rowloss, int32_t* pred,
                         float* rowp, float* dz, float inv_batch) {
    SkSoftmax sm{};
    sm.y = y; sm.y_row0 = y_row0; sm.d_row0 = d_row0; sm.logprob = logprob; sm.rowloss = rowloss;
    sm.pred = pred; sm.rowp = rowp; sm.dz = dz; sm.inv_batch = inv_batch;
    fc_skinny_fwd_mfma<<<cdiv(B, 16), 256, 0, ctx->stream>>>(x, W, b, logits, B, n_in, n_out,
                                                            TN_ACT_LINEAR, 0.f, nullptr, sm);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_fc_skinny_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int B,
                       int n_in, int n_out, float* ws) {
    const int S = cdiv(B, 128);
    fc_skinny_wgrad_mfma<<<dim3(cdiv(n_in + 1, 64), S), 256, 0, ctx->stream>>>(x, dz, ws, B, n_in, n_out);
    TN_LAUNCH_CHECK();
    const int per = (n_in + 1) * n_out, MN = n_in * n_out;
    int rc = tn_red_push(ctx, ws, dW, (uint32_t)MN, (uint32_t)S, (uint32_t)per, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, ws + MN, db, (uint32_t)n_out, (uint32_t)S, (uint32_t)per, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

int tn_fc_skinny_bwd(tn_ctx* ctx, const float* x, const float* dz, const float* W, float* dW, float* db,
                     float* dx, int B, int n_in, int n_out, float* ws, const float* prev_a, int act,
                     float prm, const uint8_t* mask) {
    const int S = cdiv(B, 128), gxW = cdiv(n_in + 1, 64), nW = gxW * S;
    const int gxD = cdiv(n_in / 4, 128), nD = gxD * cdiv(B, SKD_ROWS);
    const int grid = nW + cdiv(nD, 2);
#define SKP_GO(N_)                                                                              \
    case N_:                                                                                    \
        fc_skinny_bwd_pair<N_><<<grid, 256, 0, ctx->stream>>>(x, dz, W, ws, dx, B, n_in, prev_a, act, prm, \
                                                             mask, gxW, nW, gxD);               \
        break
    switch (n_out) {
        SKP_GO(1); SKP_GO(2); SKP_GO(3); SKP_GO(4); SKP_GO(5); SKP_GO(6); SKP_GO(7); SKP_GO(8);
        SKP_GO(9); SKP_GO(10); SKP_GO(11); SKP_GO(12); SKP_GO(13); SKP_GO(14); SKP_GO(15);
        default: SKP_GO(16);
    }
#undef SKP_GO
    TN_LAUNCH_CHECK();
    const int per = (n_in + 1) * n_out, MN = n_in * n_out;
    int rc = tn_red_push(ctx, ws, dW, (uint32_t)MN, (uint32_t)S, (uint32_t)per, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, ws + MN, db, (uint32_t)n_out, (uint32_t)S, (uint32_t)per, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

// ws must hold cdiv(B, sk_train_rb(B)) slabs of (n_in + 1) * n_out floats (tn_fc_wgrad_ws_bytes provides it)
int tn_fc_skinny_softmax_train(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits,
                               int B, int n_in, int n_out, const int32_t* y, int64_t y_row0,
                               const int64_t* d_row0, float* logprob, float* rowloss, int32_t* pred,
                               float* rowp, float* dz, float inv_batch, float* dW, float* db, float* dx,
                               float* ws, int fuse_act, int act, float prm, const uint8_t* mask) {
    SkSoftmax sm{};
    sm.y = y; sm.y_row0 = y_row0; sm.d_row0 = d_row0; sm.logprob = logprob; sm.rowloss = rowloss;
    sm.pred = pred; sm.rowp = rowp; sm.dz = dz; sm.inv_batch = inv_batch;
    const int rb = sk_train_rb(B), S = cdiv(B, rb);
#define SKT_GO(N_)                                                                              \
    case N_:                                                                                    \
        if (rb == 4)                                                                            \
            fc_skinny_softmax_train<N_, 4><<<S, 256, 0, ctx->stream>>>(x, W, b, logits, sm, ws, dx, B, n_in, act, \
                                                                       prm, mask, fuse_act);    \
        else                                                                                    \
            fc_skinny_softmax_train<N_, 16><<<S, 256, 0, ctx->stream>>>(x, W, b, logits, sm, ws, dx, B, n_in, act, \
                                                                        prm, mask, fuse_act);   \
        break
    switch (n_out) {
        SKT_GO(1); SKT_GO(2); SKT_GO(3); SKT_GO(4); SKT_GO(5); SKT_GO(6); SKT_GO(7); SKT_GO(8);
        SKT_GO(9); SKT_GO(10); SKT_GO(11); SKT_GO(12); SKT_GO(13); SKT_GO(14); SKT_GO(15);
        default: SKT_GO(16);
    }
#undef SKT_GO
    TN_LAUNCH_CHECK();
    const int per = (n_in + 1) * n_out, MN = n_in * n_out;
    int rc = tn_red_push(ctx, ws, dW, (uint32_t)MN, (uint32_t)S, (uint32_t)per, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, ws + MN, db, (uint32_t)n_out, (uint32_t)S, (uint32_t)per, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

int tn_fc_skinny_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int B, int n_in,
                       int n_out, const float* prev_a, int act, float prm, const uint8_t* mask) {
    const dim3 grid(cdiv(n_in / 4, 128), cdiv(B, SKD_ROWS));
#define SKD_GO(N_)                                                                              \
    case N_:                                                                                    \
        fc_skinny_dgrad_v4<N_><<<grid, 128, 0, ctx->stream>>>(dz, W, dx, B, n_in, prev_a, act, prm, mask); \
        break
    switch (n_out) {
        SKD_GO(1); SKD_GO(2); SKD_GO(3); SKD_GO(4); SKD_GO(5); SKD_GO(6); SKD_GO(7); SKD_GO(8);
        SKD_GO(9); SKD_GO(10); SKD_GO(11); SKD_GO(12); SKD_GO(13); SKD_GO(14); SKD_GO(15);
        default: SKD_GO(16);
    }
#undef SKD_GO
    TN_LAUNCH_CHECK();
    return TN_OK;
}
