// Fused conv + bias + activation + max-pool kernels for small feature maps (the MNIST /
// CIFAR-first-layer regime: C*f*f far below the MFMA break-even, whole receptive fields fit
// in registers).  MI355X-first design: the conv activation tensor -- the largest tensor of the
// net -- never exists in HBM.
//
//   forward : thread = one POOLED output pixel of one image.  It loads its (p+f-1)^2 x C input
//             patch into registers once, then for every output map k computes the p x p conv
//             outputs of its pooling window with wave-uniform (scalar-loaded) weights, applies
//             the activation and writes only the max.       HBM: read x once, write y once.
//   backward: same mapping; recomputes the window's conv outputs (cheap: C*f*f FMAs each),
//             finds the max itself (every tie receives the gradient, like Theano's MaxPoolGrad),
//             forms dz = g * act'(a) in registers and accumulates dW / db in registers over many
//             pixels; dz is written out only when a dgrad pass needs it.
//             HBM: read x and g once (+ write dz when needed) instead of the unfused
//             pool-bwd + wgrad traffic of ~5x that.
//
// Semantics: theanet/layer/convpool.py:54-72 (true convolution, flipped W), :106-112 (pool).
#include <cstdlib>

#include "common.h"

// conv_tile.hip: the matrix-core kernels for wide 3x3 'same' blocks
extern "C" int tn_convpool_tile_supported(int N, int C, int H, int Wd, int K, int f, int stride, int pad, int Ho,
                                          int Wo, int p, int Hp, int Wp);
extern "C" int tn_convpool_smallc_supported(int N, int C, int H, int Wd, int K, int f, int pad, int Ho,
                                            int Wo, int p, int Hp, int Wp);
int tn_conv_tile_smallc_bwd(tn_ctx* ctx, const float* x, const float* g_, const float* y, const uint8_t* mask,
                            float* dW, float* db, int N, int C, int H, int Wd, int K, int act, float prm);
int tn_conv_tile_pool_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* y,
                          uint8_t* mask, int N, int C, int H, int Wd, int K, int act, float prm);
int tn_conv_tile_pool_bwd(tn_ctx* ctx, const float* x, const float* W, const float* g_, const float* y,
                          const uint8_t* mask, float* dx, float* dW, float* db, int N, int C, int H, int Wd,
                          int K, int act, float prm, const float* prev_a, int prev_act, float prev_prm);

template <int ACT>
__device__ __forceinline__ float act_fwd_t(float z, int act, float prm) {
    if (ACT == TN_ACT_LEAKY) return fmaxf(0.f, z) + fminf(0.f, z) * prm;
    return tn_act_fwd(z, act, prm);
}
template <int ACT>
__device__ __forceinline__ float act_grad_t(float a, int act, float prm) {
    if (ACT == TN_ACT_LEAKY) return a > 0.f ? 1.f : (a < 0.f ? prm : (prm > 0.f ? 1.f + prm : 0.f));
    return tn_act_grad_from_out(a, act, prm);
}

// Input window of S x S x C values held in registers.  Row / column offsets are clamped ONCE
// (S + S integer ops), then all loads are issued back to back from base + roff + coff (hipcc
// serialises "load; wait; select" chains, so the zero-padding selects -- needed only for
// mode 'same' -- come in a separate pass after every load is in flight).  Without padding,
// clamped reads only ever feed conv outputs that are excluded by the valid flags.
template <int S, int C, bool PADDED>
struct Window {
    float v[C][S][S];
    // x: the tensor (block-uniform), base: first element of the thread's image -- unsigned 32-bit BYTE offsets from a uniform
    // base (the launchers check the input's element count < 2^30) let every load take the scalar-base + 32-bit-offset form; with
    // a per-thread base pointer each of the S * S * C loads carried its own 64-bit address (v_ashrrev + v_lshl_add_u64)
    __device__ __forceinline__ void load(const float* __restrict__ x, unsigned base, int H, int Wd, int y0, int x0) {
        unsigned roff[S], coff[S];
        bool rok[S], cok[S];
#pragma unroll
        for (int r = 0; r < S; ++r) {
            const int yy = y0 + r, xx = x0 + r;
            rok[r] = (yy >= 0) && (yy < H);
            cok[r] = (xx >= 0) && (xx < Wd);
            roff[r] = (unsigned)(min(max(yy, 0), H - 1) * Wd);
            coff[r] = (unsigned)min(max(xx, 0), Wd - 1);
        }
        const unsigned HW = (unsigned)(H * Wd);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < S; ++r)
#pragma unroll
                for (int q = 0; q < S; ++q)
                    v[c][r][q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) +
                                                                 (base + c * HW + roff[r] + coff[q]) * 4u);
        if (PADDED) {
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int r = 0; r < S; ++r)
#pragma unroll
                    for (int q = 0; q < S; ++q) v[c][r][q] = (rok[r] && cok[q]) ? v[c][r][q] : 0.f;
        }
    }
};

// conv outputs of the P x P window for map k (fixed FMA order: c, u, v -- shared by fwd and bwd)
template <int F, int P, int C, bool PADDED>
__device__ __forceinline__ void window_conv(const Window<P + F - 1, C, PADDED>& pt,
                                            const float* __restrict__ Wk, float bias,
                                            float (&z)[P][P]) {
#pragma unroll
    for (int di = 0; di < P; ++di)
#pragma unroll
        for (int dj = 0; dj < P; ++dj) z[di][dj] = bias;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int u = 0; u < F; ++u)
#pragma unroll
            for (int v = 0; v < F; ++v) {
                const float w = Wk[(c * F + (F - 1 - u)) * F + (F - 1 - v)];   // wave-uniform
#pragma unroll
                for (int di = 0; di < P; ++di)
#pragma unroll
                    for (int dj = 0; dj < P; ++dj)
                        z[di][dj] = fmaf(pt.v[c][di + u][dj + v], w, z[di][dj]);
            }
}

template <int F, int P, int C, int ACT, bool PADDED>
__global__ __launch_bounds__(256) void convpool_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    float* __restrict__ y, uint8_t* __restrict__ mask, int N, int H, int Wd, int K, int pad, int Ho,
    int Wo, int Hp, int Wp, int act, float prm) {
    const int HpWp = Hp * Wp;
    const unsigned total = (unsigned)N * HpWp;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    const int n = (int)(t / (unsigned)HpWp);
    const int q = (int)(t - (unsigned)n * HpWp);
    const int pi = q / Wp, pj = q - pi * Wp;
    Window<P + F - 1, C, PADDED> pt;
    pt.load(x, (unsigned)n * C * H * Wd, H, Wd, pi * P - pad, pj * P - pad);
    bool valid[P][P];
#pragma unroll
    for (int di = 0; di < P; ++di)
#pragma unroll
        for (int dj = 0; dj < P; ++dj) valid[di][dj] = (pi * P + di < Ho) && (pj * P + dj < Wo);
    float* yn = y + (size_t)n * K * HpWp + q;
    // short batches: gridDim.y slices of the filters, so that the launch has enough waves and a thread's serial
    // chain (K * C * F * F * P * P dependent-free FMAs, but one wave at a time per SIMD) is gridDim.y times shorter
    const int kper = (K + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kbeg = (int)blockIdx.y * kper, kend = min(K, kbeg + kper);
    for (int k = kbeg; k < kend; ++k) {
        float z[P][P];
        window_conv<F, P, C, PADDED>(pt, W + (size_t)k * C * F * F, b[k], z);
        float m = -INFINITY;
#pragma unroll
        for (int di = 0; di < P; ++di)
#pragma unroll
            for (int dj = 0; dj < P; ++dj) {
                z[di][dj] = act_fwd_t<ACT>(z[di][dj], act, prm);
                m = valid[di][dj] ? fmaxf(m, z[di][dj]) : m;
            }
        yn[(size_t)k * HpWp] = m;
        if (mask) {       // bit di*P+dj: that window element attains the maximum (ties: all of them)
                          // bits 4 / 5: y > 0 / y < 0 (all a leaky-relu backward needs of y)
            unsigned bits = 0;
#pragma unroll
            for (int di = 0; di < P; ++di)
#pragma unroll
                for (int dj = 0; dj < P; ++dj)
                    bits |= (valid[di][dj] && z[di][dj] == m) ? (1u << (di * P + dj)) : 0u;
            bits |= (m > 0.f ? 16u : 0u) | (m < 0.f ? 32u : 0u);      // sign of the pooled value
            mask[((size_t)n * K + k) * HpWp + q] = (uint8_t)bits;
        }
    }
}

// ---- cross-lane helpers (DPP: no LDS round trip) -------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
#define DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140

// sum over the 64 lanes; the total is returned in every lane of... lane 0 (uniform scalar adds)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_f<DPP_QUAD_XOR1>(v);
    v += dpp_f<DPP_QUAD_XOR2>(v);
    v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);          // every lane of a 16-lane row now holds the row sum
    const int iv = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(iv, 0)) +
           __int_as_float(__builtin_amdgcn_readlane(iv, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(iv, 32)) +
           __int_as_float(__builtin_amdgcn_readlane(iv, 48));
}

// Backward of the fused block, p = 2.  Thread = one CONV output position; the four positions of
// a pooling window sit in four adjacent lanes (a DPP quad), so the window max is two DPP ops
// and every lane does exactly its own C*f*f wgrad FMAs (no multiply-by-zero work).
// Slot s of an image: window = s >> 2, (di, dj) = ((s >> 1) & 1, s & 1).
template <int F, int C, int KT, int ACT, bool PADDED>
__global__ __launch_bounds__(256) void convpool_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    const float* __restrict__ g, float* __restrict__ dz_out, float* __restrict__ partial,
    float* __restrict__ dbpartial, int N, int H, int Wd, int K, int pad, int Ho, int Wo, int Hp,
    int Wp, int act, float prm) {
    constexpr int FF = F * F;
    constexpr int NACC = KT * C * FF + KT;
    __shared__ float red[4][NACC];
    const int HpWp = Hp * Wp, HoWo = Ho * Wo;
    const unsigned slots = HpWp * 4;
    const unsigned total = (unsigned)N * slots;
    const int k0 = blockIdx.y * KT;

    float acc[KT][C][FF];
    float accb[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        accb[kk] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int s = 0; s < FF; ++s) acc[kk][c][s] = 0.f;
    }

    // grid-stride over slots; the trip count is uniform per block so the DPP quads stay whole
    const unsigned stride = gridDim.x * 256u;
    for (unsigned base = blockIdx.x * 256u; base < total; base += stride) {
        const unsigned t = base + threadIdx.x;
        const bool live = t < total;
        const unsigned tt = live ? t : total - 1;
        const int n = (int)(tt / slots);
        const int s = (int)(tt - (unsigned)n * slots);
        const int q = s >> 2;
        const int pi = q / Wp, pj = q - pi * Wp;
        const int i = pi * 2 + ((s >> 1) & 1), j = pj * 2 + (s & 1);
        const bool valid = live && (i < Ho) && (j < Wo);
        // F x F x C input window of this conv output (all loads in flight together)
        Window<F, C, PADDED> pt;
        pt.load(x, (unsigned)n * C * H * Wd, H, Wd, min(i, Ho - 1) - pad, min(j, Wo - 1) - pad);
        float gk[KT];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk)
            gk[kk] = g[((size_t)n * K + min(k0 + kk, K - 1)) * HpWp + q];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const int k = k0 + kk;
            if (k < K) {   // wave-uniform
                const float* Wk = W + (size_t)k * C * FF;      // wave-uniform -> scalar loads
                float z = b[k];
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int u = 0; u < F; ++u)
#pragma unroll
                        for (int v = 0; v < F; ++v)
                            z = fmaf(pt.v[c][u][v], Wk[(c * F + (F - 1 - u)) * F + (F - 1 - v)], z);
                const float a = act_fwd_t<ACT>(z, act, prm);
                float m = valid ? a : -INFINITY;
                m = fmaxf(m, dpp_f<DPP_QUAD_XOR1>(m));
                m = fmaxf(m, dpp_f<DPP_QUAD_XOR2>(m));
                const float d = (valid && a == m) ? gk[kk] * act_grad_t<ACT>(a, act, prm) : 0.f;
                if (dz_out && valid) dz_out[((size_t)n * K + k) * HoWo + i * Wo + j] = d;
                accb[kk] += d;
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int u = 0; u < F; ++u)
#pragma unroll
                        for (int v = 0; v < F; ++v)
                            acc[kk][c][u * F + v] = fmaf(d, pt.v[c][u][v], acc[kk][c][u * F + v]);
            }
        }
    }

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int s = 0; s < FF; ++s) {
                const float r = wave_sum_dpp(acc[kk][c][s]);
                if (lane == 0) red[wave][(kk * C + c) * FF + s] = r;
            }
        const float rb = wave_sum_dpp(accb[kk]);
        if (lane == 0) red[wave][KT * C * FF + kk] = rb;
    }
    __syncthreads();
    const int KCFF = K * C * FF;
    for (int s = threadIdx.x; s < NACC; s += 256) {
        const float r = red[0][s] + red[1][s] + red[2][s] + red[3][s];
        if (s < KT * C * FF) {
            const int kk = s / (C * FF), rem = s - kk * C * FF;
            if (k0 + kk < K) partial[(size_t)blockIdx.x * KCFF + (size_t)(k0 + kk) * C * FF + rem] = r;
        } else {
            const int kk = s - KT * C * FF;
            if (k0 + kk < K) dbpartial[(size_t)blockIdx.x * K + k0 + kk] = r;
        }
    }
}

// Backward of the fused block from the forward's pooling mask, f = 3, p = 2.  Thread = one POOLING
// window: its 4 x 4 x C input patch is loaded once (16 C loads for 4 conv outputs), dz of the 4
// window elements is mask bit ? g * act'(y) : 0 -- no conv recompute, no cross-lane max -- and
// the wgrad FMAs run on the patch registers.  Same partial-slab contract as convpool_bwd_kernel.
template <int C, int KT, int ACT, bool PADDED>
__global__ __launch_bounds__(256) void convpool_bwd_mask_kernel(
    const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ y,
    const uint8_t* __restrict__ mask, float* __restrict__ dz_out, float* __restrict__ partial,
    float* __restrict__ dbpartial, int N, int H, int Wd, int K, int pad, int Ho, int Wo, int Hp,
    int Wp, int act, float prm) {
    constexpr int F = 3, FF = 9;
    constexpr int NACC = KT * C * FF + KT;
    __shared__ float red[4][NACC];
    __shared__ __attribute__((aligned(16))) float redx[NACC <= 48 ? NACC : 1][NACC <= 48 ? 256 : 1];
    const int HpWp = Hp * Wp, HoWo = Ho * Wo;
    const unsigned total = (unsigned)N * HpWp;
    const int k0 = blockIdx.y * KT;
    const float tie = prm > 0.f ? 1.f + prm : 0.f;

    float acc[KT][C][FF];
    float accb[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        accb[kk] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int s = 0; s < FF; ++s) acc[kk][c][s] = 0.f;
    }
    const unsigned stride = gridDim.x * 256u;
    for (unsigned base = blockIdx.x * 256u; base < total; base += stride) {
        const unsigned t = base + threadIdx.x;
        const bool live = t < total;
        const unsigned tt = live ? t : total - 1;
        const int n = (int)(tt / (unsigned)HpWp);
        const int q = (int)(tt - (unsigned)n * HpWp);
        const int pi = q / Wp, pj = q - pi * Wp;
        Window<4, C, PADDED> pt;
        pt.load(x, (unsigned)n * C * H * Wd, H, Wd, 2 * pi - pad, 2 * pj - pad);
        float gy[KT];
        unsigned mk[KT];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const size_t e = ((size_t)n * K + min(k0 + kk, K - 1)) * HpWp + q;
            gy[kk] = g[e];
            mk[kk] = mask[e];
            if (ACT != TN_ACT_LEAKY) gy[kk] *= tn_act_grad_from_out(y[e], act, prm);
        }
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const int k = k0 + kk;
            if (k < K) {   // wave-uniform
                if (ACT == TN_ACT_LEAKY) {
                    float gp = (mk[kk] & 32u) ? prm : tie;
                    gp = (mk[kk] & 16u) ? 1.f : gp;
                    gy[kk] *= gp;
                }
                const float gl = live ? gy[kk] : 0.f;
                float d[2][2];
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r >> 1][r & 1] = (mk[kk] >> r & 1u) ? gl : 0.f;
                if (dz_out && live) {
                    float* o = dz_out + ((size_t)n * K + k) * HoWo + (2 * pi) * Wo + 2 * pj;
                    const bool vj = 2 * pj + 1 < Wo, vi = 2 * pi + 1 < Ho;
                    o[0] = d[0][0];
                    if (vj) o[1] = d[0][1];
                    if (vi) o[Wo] = d[1][0];
                    if (vi && vj) o[Wo + 1] = d[1][1];
                }
                accb[kk] += (d[0][0] + d[0][1]) + (d[1][0] + d[1][1]);
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int u = 0; u < F; ++u)
#pragma unroll
                        for (int v = 0; v < F; ++v) {
                            float a = acc[kk][c][u * F + v];
                            a = fmaf(d[0][0], pt.v[c][u][v], a);
                            a = fmaf(d[0][1], pt.v[c][u][v + 1], a);
                            a = fmaf(d[1][0], pt.v[c][u + 1][v], a);
                            a = fmaf(d[1][1], pt.v[c][u + 1][v + 1], a);
                            acc[kk][c][u * F + v] = a;
                        }
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int KCFF = K * C * FF;
    if constexpr (NACC <= 48) {
        // (round 4) A wave-wide sum per accumulator -- 4 DPP adds + 4 v_readlane + their wait states, 40 times -- was 600
        // of the ~1500 instructions a thread executes.  Through LDS: every thread parks its NACC sums ([accumulator]
        // [thread], conflict-free), then wave w adds up rows w, w + 4, ...: a lane takes 4 neighbours (one 16-byte read),
        // then one wave-wide sum -- a quarter of the wave-wide sums, in a fixed order.
        float* const park = &redx[0][0];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int s2 = 0; s2 < FF; ++s2) park[((kk * C + c) * FF + s2) * 256 + threadIdx.x] = acc[kk][c][s2];
            park[(KT * C * FF + kk) * 256 + threadIdx.x] = accb[kk];
        }
        __syncthreads();
        for (int s2 = wave; s2 < NACC; s2 += 4) {
            const float4 q = *reinterpret_cast<const float4*>(park + s2 * 256 + 4 * lane);
            const float r = wave_sum_dpp((q.x + q.y) + (q.z + q.w));
            if (lane == 0) {
                if (s2 < KT * C * FF) {
                    const int kk = s2 / (C * FF), rem = s2 - kk * C * FF;
                    if (k0 + kk < K) partial[(size_t)blockIdx.x * KCFF + (size_t)(k0 + kk) * C * FF + rem] = r;
                } else {
                    const int kk = s2 - KT * C * FF;
                    if (k0 + kk < K) dbpartial[(size_t)blockIdx.x * K + k0 + kk] = r;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int s = 0; s < FF; ++s) {
                const float r = wave_sum_dpp(acc[kk][c][s]);
                if (lane == 0) red[wave][(kk * C + c) * FF + s] = r;
            }
        const float rb = wave_sum_dpp(accb[kk]);
        if (lane == 0) red[wave][KT * C * FF + kk] = rb;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < NACC; s += 256) {
        const float r = red[0][s] + red[1][s] + red[2][s] + red[3][s];
        if (s < KT * C * FF) {
            const int kk = s / (C * FF), rem = s - kk * C * FF;
            if (k0 + kk < K) partial[(size_t)blockIdx.x * KCFF + (size_t)(k0 + kk) * C * FF + rem] = r;
        } else {
            const int kk = s - KT * C * FF;
            if (k0 + kk < K) dbpartial[(size_t)blockIdx.x * K + k0 + kk] = r;
        }
    }
}

// shared with conv.hip
int tn_conv_wgrad_finish(tn_ctx* ctx, const float* partial, const float* dbpartial, float* dW,
                         float* db, int nblk, int K, int C, int f);

template <int F, int P, int C>
static int launch_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* y,
                      uint8_t* mask, int N, int H, int Wd, int K, int pad, int Ho, int Wo, int Hp, int Wp, int act,
                      float prm) {
    const long long total = (long long)N * Hp * Wp;
    TN_REQUIRE(total < (1ll << 31), "tn_convpool_fwd: too many outputs for 32-bit indexing");
    TN_REQUIRE((long long)N * C * H * Wd < (1ll << 30), "tn_convpool_fwd: input too large for 32-bit byte offsets");
    // filter slices (blockIdx.y) until the launch has about two blocks per CU
    int ks = 1;
    while (ks < 8 && ks * 2 <= K && (long long)cdiv(total, 256) * ks < 2 * ctx->num_cus) ks *= 2;
    {
        static int force = -1;
        if (force < 0) {
            const char* e = getenv("TN_CONVPOOL_KS");
            force = e ? atoi(e) : 0;
        }
        if (force > 0) ks = force;
    }
    const dim3 grid(cdiv(total, 256), ks);
#define CP_L(ACT_, PAD_)                                                                          \
    convpool_fwd_kernel<F, P, C, ACT_, PAD_><<<grid, 256, 0, ctx->stream>>>(                       \
        x, W, b, y, mask, N, H, Wd, K, pad, Ho, Wo, Hp, Wp, act, prm)
    if (act == TN_ACT_LEAKY) {
        if (pad) CP_L(TN_ACT_LEAKY, true); else CP_L(TN_ACT_LEAKY, false);
    } else {
        if (pad) CP_L(-1, true); else CP_L(-1, false);
    }
#undef CP_L
    TN_LAUNCH_CHECK();
    return TN_OK;
}

static int tn_tune_kt() { return 4; }          // filters per thread of the recomputing backward

static int tn_tune_ppt() { return 8; }         // pixels per thread of the recomputing backward

template <int F, int C, int KT>
static int launch_bwd(tn_ctx* ctx, const float* x, const float* W, const float* b, const float* g,
                      float* dz, float* dW, float* db, int N, int H, int Wd, int K, int pad, int Ho,
                      int Wo, int Hp, int Wp, int act, float prm) {
    const long long total = (long long)N * Hp * Wp * 4;
    int nblk = cdiv(total, 256 * tn_tune_ppt());
    if (nblk > 2048) nblk = 2048;
    if (nblk < 1) nblk = 1;
    const size_t KCFF = (size_t)K * C * F * F;
    float* partial;
    int rc = tn_scratch_get(ctx, (size_t)nblk * (KCFF + K) * sizeof(float), &partial);
    if (rc) return rc;
    float* dbpartial = partial + (size_t)nblk * KCFF;
    TN_REQUIRE(total < (1ll << 31), "tn_convpool_bwd: too many outputs for 32-bit indexing");
    TN_REQUIRE((long long)N * C * H * Wd < (1ll << 30), "tn_convpool_bwd: input too large for 32-bit byte offsets");
    const dim3 grid(nblk, cdiv(K, KT));
#define CP_L(ACT_, PAD_)                                                                          \
    convpool_bwd_kernel<F, C, KT, ACT_, PAD_><<<grid, 256, 0, ctx->stream>>>(                       \
        x, W, b, g, dz, partial, dbpartial, N, H, Wd, K, pad, Ho, Wo, Hp, Wp, act, prm)
    if (act == TN_ACT_LEAKY) {
        if (pad) CP_L(TN_ACT_LEAKY, true); else CP_L(TN_ACT_LEAKY, false);
    } else {
        if (pad) CP_L(-1, true); else CP_L(-1, false);
    }
#undef CP_L
    TN_LAUNCH_CHECK();
    return tn_conv_wgrad_finish(ctx, partial, dbpartial, dW, db, nblk, K, C, F);
}

static int tn_tune_mwin() { return 4; }        // windows per thread of the mask-driven backward (full batches)

template <int C, int KT>
static int launch_bwd_mask(tn_ctx* ctx, const float* x, const float* g, const float* y,
                           const uint8_t* mask, float* dz, float* dW, float* db, int N, int H, int Wd,
                           int K, int pad, int Ho, int Wo, int Hp, int Wp, int act, float prm) {
    const long long total = (long long)N * Hp * Wp;
    TN_REQUIRE(total < (1ll << 31), "tn_convpool_bwd_mask: too many outputs for 32-bit indexing");
    TN_REQUIRE((long long)N * C * H * Wd < (1ll << 30), "tn_convpool_bwd_mask: input too large for 32-bit byte offsets");
    int mwin = tn_tune_mwin();                          // windows per thread ...
    while (mwin > 1 && cdiv(total, 256 * mwin) < 2 * ctx->num_cus) mwin >>= 1;   // ... fewer for short batches (a 512-image shard: 85 blocks otherwise)
    int nblk = cdiv(total, 256 * mwin);
    if (nblk > 2048) nblk = 2048;
    if (nblk < 1) nblk = 1;
    const size_t KCFF = (size_t)K * C * 9;
    float* partial;
    int rc = tn_scratch_get(ctx, (size_t)nblk * (KCFF + K) * sizeof(float), &partial);
    if (rc) return rc;
    float* dbpartial = partial + (size_t)nblk * KCFF;
    const dim3 grid(nblk, cdiv(K, KT));
#define CP_L(ACT_, PAD_)                                                                          \
    convpool_bwd_mask_kernel<C, KT, ACT_, PAD_><<<grid, 256, 0, ctx->stream>>>(                     \
        x, g, y, mask, dz, partial, dbpartial, N, H, Wd, K, pad, Ho, Wo, Hp, Wp, act, prm)
    if (act == TN_ACT_LEAKY) {
        if (pad) CP_L(TN_ACT_LEAKY, true); else CP_L(TN_ACT_LEAKY, false);
    } else {
        if (pad) CP_L(-1, true); else CP_L(-1, false);
    }
#undef CP_L
    TN_LAUNCH_CHECK();
    return tn_conv_wgrad_finish(ctx, partial, dbpartial, dW, db, nblk, K, C, 3);
}

extern "C" {

int tn_convpool_bwd_mask(tn_ctx* ctx, const float* x, const float* g, const float* y,
                         const uint8_t* mask, float* dz, float* dW, float* db, int N, int C, int H,
                         int Wd, int K, int f, int pad_lo, int Ho, int Wo, int p, int Hp, int Wp,
                         int act, float act_param) {
    TN_REQUIRE(f == 3 && p == 2 && C >= 1 && C <= 4, "tn_convpool_bwd_mask: unsupported C=%d f=%d p=%d", C, f, p);
    TN_REQUIRE(x && g && y && mask && dW && db, "tn_convpool_bwd_mask: null argument");
    // weight gradients only (first layer), 'same' block with even maps: matrix-core kernel of conv_tile.hip
    if (!dz && tn_convpool_smallc_supported(N, C, H, Wd, K, f, pad_lo, Ho, Wo, p, Hp, Wp))
        return tn_conv_tile_smallc_bwd(ctx, x, g, y, mask, dW, db, N, C, H, Wd, K, act, act_param);
    if (dz && (Hp * p < Ho || Wp * p < Wo))   // rows/cols outside every window (ignore_border)
        TN_HIP(hipMemsetAsync(dz, 0, (size_t)N * K * Ho * Wo * sizeof(float), ctx->stream));
#define CP_BWDM(C_, KT_)                                                                          \
    return launch_bwd_mask<C_, KT_>(ctx, x, g, y, mask, dz, dW, db, N, H, Wd, K, pad_lo, Ho, Wo, Hp, Wp, \
                                    act, act_param)
    switch (C) {
        case 1: CP_BWDM(1, 4);
        case 2: CP_BWDM(2, 4);
        case 3: CP_BWDM(3, 4);
        default: CP_BWDM(4, 2);
    }
#undef CP_BWDM
}

int tn_convpool_supported(int C, int f, int stride, int p) {
    if (stride != 1 || p != 2) return 0;
    if (f == 3) return C >= 1 && C <= 4;
    if (f == 5) return C >= 1 && C <= 2;
    return 0;
}

int tn_convpool_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* y, int N,
                    int C, int H, int Wd, int K, int f, int pad_lo, int Ho, int Wo, int p, int Hp,
                    int Wp, int act, float act_param) {
    return tn_convpool_fwd_mask(ctx, x, W, b, y, nullptr, N, C, H, Wd, K, f, pad_lo, Ho, Wo, p, Hp, Wp,
                                act, act_param);
}

int tn_convpool_fwd_mask(tn_ctx* ctx, const float* x, const float* W, const float* b, float* y,
                         uint8_t* mask, int N, int C, int H, int Wd, int K, int f, int pad_lo, int Ho,
                         int Wo, int p, int Hp, int Wp, int act, float act_param) {
    TN_REQUIRE(!ctx->mm_f16, "tn_convpool_fwd_mask: fp32 tensors in DTYPE float16 mode (the mode's entry points are tn_c8_*)");
    if (!tn_convpool_supported(C, f, 1, p) &&
        tn_convpool_tile_supported(N, C, H, Wd, K, f, 1, pad_lo, Ho, Wo, p, Hp, Wp))
        return tn_conv_tile_pool_fwd(ctx, x, W, b, y, mask, N, C, H, Wd, K, act, act_param);   // wide layers
    TN_REQUIRE(tn_convpool_supported(C, f, 1, p), "tn_convpool_fwd: unsupported C=%d f=%d p=%d", C, f, p);
#define CP_FWD(F_, C_)                                                                            \
    return launch_fwd<F_, 2, C_>(ctx, x, W, b, y, mask, N, H, Wd, K, pad_lo, Ho, Wo, Hp, Wp, act,  \
                                 act_param)
    if (f == 3) {
        switch (C) {
            case 1: CP_FWD(3, 1);
            case 2: CP_FWD(3, 2);
            case 3: CP_FWD(3, 3);
            default: CP_FWD(3, 4);
        }
    } else {
        if (C == 1) CP_FWD(5, 1);
        CP_FWD(5, 2);
    }
#undef CP_FWD
}

int tn_convpool_bwd_mask_dx(tn_ctx* ctx, const float* x, const float* W, const float* g, const float* y,
                            const uint8_t* mask, float* dx, float* dW, float* db, int N, int C, int H,
                            int Wd, int K, int f, int pad_lo, int Ho, int Wo, int p, int Hp, int Wp, int act,
                            float act_param, const float* prev_a, int prev_act, float prev_act_param) {
    TN_REQUIRE(x && W && g && y && mask, "tn_convpool_bwd_mask_dx: null argument");
    TN_REQUIRE(!ctx->mm_f16, "tn_convpool_bwd_mask_dx: fp32 tensors in DTYPE float16 mode (the mode's entry points are tn_c8_*)");
    TN_REQUIRE(tn_convpool_tile_supported(N, C, H, Wd, K, f, 1, pad_lo, Ho, Wo, p, Hp, Wp),
               "tn_convpool_bwd_mask_dx: unsupported block (C=%d K=%d %dx%d f=%d p=%d)", C, K, H, Wd, f, p);
    return tn_conv_tile_pool_bwd(ctx, x, W, g, y, mask, dx, dW, db, N, C, H, Wd, K, act, act_param, prev_a,
                                 prev_act, prev_act_param);
}

int tn_convpool_bwd(tn_ctx* ctx, const float* x, const float* W, const float* b, const float* g,
                    float* dz, float* dW, float* db, int N, int C, int H, int Wd, int K, int f,
                    int pad_lo, int Ho, int Wo, int p, int Hp, int Wp, int act, float act_param) {
    TN_REQUIRE(tn_convpool_supported(C, f, 1, p), "tn_convpool_bwd: unsupported C=%d f=%d p=%d", C, f, p);
    if (dz && (Hp * p < Ho || Wp * p < Wo))   // rows/cols outside every window (ignore_border)
        TN_HIP(hipMemsetAsync(dz, 0, (size_t)N * K * Ho * Wo * sizeof(float), ctx->stream));
#define CP_BWD(F_, C_, KT_)                                                                       \
    return launch_bwd<F_, C_, KT_>(ctx, x, W, b, g, dz, dW, db, N, H, Wd, K, pad_lo, Ho, Wo, Hp, Wp, \
                                   act, act_param)
    if (f == 3) {
        switch (C) {
            case 1: CP_BWD(3, 1, 4);
            case 2: CP_BWD(3, 2, 4);
            case 3: CP_BWD(3, 3, 4);
            default:
                if (tn_tune_kt() == 2) CP_BWD(3, 4, 2);
                CP_BWD(3, 4, 4);
        }
    } else {
        if (C == 1) CP_BWD(5, 1, 4);
        CP_BWD(5, 2, 2);
    }
#undef CP_BWD
}

}  // extern "C"
