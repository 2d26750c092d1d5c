// Shared pieces of the LDS-tile 3x3 convolution kernels (conv_tile.hip: fp32 MFMA; conv_c8.hip takes the
// pooled-gradient source from here): geometry struct, pooled-gradient expansion, epilogues.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CT_CH 8           // input channels per chunk
#define CT_SX 4           // 16-byte staging slots per thread for the input tile

// dz of a conv + act + 2x2 max-pool block, formed on the fly from what the fused forward left behind:
// the pooled gradient g, the pooling mask (bit 2*di+dj: that window element attained the maximum,
// bits 4 / 5: pooled value > 0 / < 0) and, for activations other than leaky-ReLU, the pooled output y.
struct PoolSrc {
    const float* g;
    const float* y;
    const uint8_t* mask;
    int Hp, Wp, act;
    float prm;
};
// dz[plane][row][col4 .. col4+3] (col4 % 4 == 0, pooled width even: the two window columns are one
// 8-byte g load and one 2-byte mask load)
__device__ __forceinline__ float4 pool_expand4(const PoolSrc& s, int plane, int row, int col4) {
    const int idx = (plane * s.Hp + (row >> 1)) * s.Wp + (col4 >> 1);
    const float2 g2 = *reinterpret_cast<const float2*>(s.g + idx);
    const unsigned m2 = *reinterpret_cast<const unsigned short*>(s.mask + idx);
    const unsigned m0 = m2 & 0xffu, m1 = m2 >> 8;
    float ga0, ga1;
    if (s.act == TN_ACT_LEAKY) {
        const float tie = 1.f + s.prm;
        float p0 = (m0 & 32u) ? s.prm : tie, p1 = (m1 & 32u) ? s.prm : tie;
        p0 = (m0 & 16u) ? 1.f : p0;
        p1 = (m1 & 16u) ? 1.f : p1;
        ga0 = g2.x * p0; ga1 = g2.y * p1;
    } else {
        const float2 y2 = *reinterpret_cast<const float2*>(s.y + idx);
        ga0 = g2.x * tn_act_grad_from_out(y2.x, s.act, s.prm);
        ga1 = g2.y * tn_act_grad_from_out(y2.y, s.act, s.prm);
    }
    const int sh = (row & 1) * 2;
    return make_float4((m0 >> sh) & 1u ? ga0 : 0.f, (m0 >> (sh + 1)) & 1u ? ga0 : 0.f,
                       (m1 >> sh) & 1u ? ga1 : 0.f, (m1 >> (sh + 1)) & 1u ? ga1 : 0.f);
}

struct ConvTG {
    const float* x;       // gathered tensor (N, C, H, Wd)
    const float* wt;      // arranged weights [KT][nchunk][4][9][2][32*FT]
    float* out;           // (N, K, Ho, Wo)
    const float* bias;
    const float* prev_a;
    int N, C, H, Wd, K, pad, Ho, Wo, act;
    float prm;
    int KT, MT, RT, NI, TH, THi, RS, LP, plane, nchunk, TP, q4, nx4, vec_out;
    PoolSrc ps;                // POOL dgrad: the gathered tensor is formed from (g, mask, y)
    uint8_t* mask_out;         // POOL forward: pooling mask (may be NULL)
    unsigned long long* dbg;   // TN_CT_DBG=1: per block {start, prologue done, loop done, end} (s_memtime) + wall clock
};

// Epilogue of conv_tile_kernel: the block's accumulators (FT filter tiles x 2 pixel tiles per
// wave, MFMA C/D layout) -> bias + act (+ 2x2 max-pool + mask) or act' of the layer below -> HBM.
// ACT >= 0: the activation kind is a compile-time constant (leaky-ReLU, the nets' usual one): with a run-time
// kind every element of the epilogue paid for the whole switch of tn_act_fwd / tn_act_grad_from_out -- 12 k
// of a 17 k-cycle forward epilogue (cycle stamps, tools/dbg_c16.py).
template <int FT, bool DGRAD, bool POOL, int ACT>
__device__ __forceinline__ void ct_epilogue_impl(const ConvTG& g_, f32x16 (&acc)[FT][2], float* ct_smem, int kt, int n0,
                                                 int r0, int lane, int wave, int l31, int hi) {
    // a view of g whose .act folds to the constant
    struct G {
        const float* x; const float* wt; float* out; const float* bias; const float* prev_a;
        int N, K, Ho, Wo, act; float prm; int TH, TP, vec_out; uint8_t* mask_out;
    } g{g_.x, g_.wt, g_.out, g_.bias, g_.prev_a, g_.N, g_.K, g_.Ho, g_.Wo, ACT >= 0 ? ACT : g_.act, g_.prm, g_.TH, g_.TP, g_.vec_out,
        g_.mask_out};
    constexpr int KBF = 32 * FT;
    const int HoWo = g.Ho * g.Wo;
    if (!DGRAD) {
        // bias joins the accumulators here: all of a lane's bias loads are in flight together, instead of one
        // dependent global load per filter inside the store loops below
        float bv[FT][16];
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bv[f][r] = g.bias[min(kt * KBF + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, g.K - 1)];
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[f][0][r] += bv[f][r]; acc[f][1][r] += bv[f][r]; }
    }
    if (!DGRAD && POOL) {
        // ---- pooled epilogue: tile -> LDS [filter][pixel]; lane = one pooled pixel, wave = one filter
        float* Os = ct_smem;
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Os[(f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 256 + wave * 64 + pt * 32 + l31] = acc[f][pt][r];
        __syncthreads();
        const int Wp = g.Wo >> 1, Hp = g.Ho >> 1, THp = g.TH >> 1, perp = THp * Wp;
        const int pp = lane < (g.TP >> 2) ? lane : 0;
        const int ni = pp / perp, rem = pp - ni * perp;
        const int pr = rem / Wp, pc = rem - pr * Wp;
        const int p00 = ni * g.TH * g.Wo + 2 * pr * g.Wo + 2 * pc;
        const bool ok = lane < (g.TP >> 2) && n0 + ni < g.N && r0 + 2 * pr < g.Ho;
        const size_t obase = ((size_t)(n0 + ni) * g.K * Hp + (r0 >> 1) + pr) * Wp + pc;
        if (ok) {
#pragma unroll 2
            for (int i = 0; i < KBF / 4; ++i) {
                const int kl = wave + 4 * i, k = kt * KBF + kl;
                if (k >= g.K) break;
                const float2 t0 = *reinterpret_cast<const float2*>(Os + kl * 256 + p00);
                const float2 t1 = *reinterpret_cast<const float2*>(Os + kl * 256 + p00 + g.Wo);
                const float bk = 0.f;
                const float a00 = tn_act_fwd(t0.x + bk, g.act, g.prm), a01 = tn_act_fwd(t0.y + bk, g.act, g.prm);
                const float a10 = tn_act_fwd(t1.x + bk, g.act, g.prm), a11 = tn_act_fwd(t1.y + bk, g.act, g.prm);
                const float m = fmaxf(fmaxf(a00, a01), fmaxf(a10, a11));
                const size_t o = obase + (size_t)k * Hp * Wp;
                g.out[o] = m;
                if (g.mask_out) {
                    unsigned bits = (a00 == m ? 1u : 0u) | (a01 == m ? 2u : 0u) | (a10 == m ? 4u : 0u) | (a11 == m ? 8u : 0u);
                    bits |= (m > 0.f ? 16u : 0u) | (m < 0.f ? 32u : 0u);
                    g.mask_out[o] = (uint8_t)bits;
                }
            }
        }
    } else if (g.vec_out) {
        // ---- epilogue through LDS: the block's (32*FT filters) x (256 pixels) tile is laid out
        // [filter][pixel]; a wave then owns whole filter rows: one 16-byte access per lane, 1 KB bursts
        float* Os = ct_smem;
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Os[(f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 256 + wave * 64 + pt * 32 + l31] = acc[f][pt][r];
        __syncthreads();
        const int p = 4 * lane;                         // this thread's 4 pixels (same for all its filters)
        const int per = g.TH * g.Wo;
        const int pp = p < g.TP ? p : 0;
        const int ni = pp / per, rem = pp - ni * per;
        const int r = rem / g.Wo, col = rem - r * g.Wo;
        const bool ok = p < g.TP && n0 + ni < g.N && r0 + r < g.Ho;
        const size_t pbase = (size_t)(n0 + ni) * g.K * HoWo + (r0 + r) * g.Wo + col;
        if (ok) {
#pragma unroll 4
            for (int i = 0; i < KBF / 4; ++i) {
                const int kl = wave + 4 * i, k = kt * KBF + kl;
                if (k >= g.K) break;
                float4 v = *reinterpret_cast<const float4*>(Os + kl * 256 + p);
                if (DGRAD) {
                    if (g.prev_a) {
                        const float4 pa = *reinterpret_cast<const float4*>(g.prev_a + pbase + (size_t)k * HoWo);
                        v.x *= tn_act_grad_from_out(pa.x, g.act, g.prm);
                        v.y *= tn_act_grad_from_out(pa.y, g.act, g.prm);
                        v.z *= tn_act_grad_from_out(pa.z, g.act, g.prm);
                        v.w *= tn_act_grad_from_out(pa.w, g.act, g.prm);
                    }
                } else {
                    const float bk = 0.f;
                    v.x = tn_act_fwd(v.x + bk, g.act, g.prm);
                    v.y = tn_act_fwd(v.y + bk, g.act, g.prm);
                    v.z = tn_act_fwd(v.z + bk, g.act, g.prm);
                    v.w = tn_act_fwd(v.w + bk, g.act, g.prm);
                }
                *reinterpret_cast<float4*>(g.out + pbase + (size_t)k * HoWo) = v;
            }
        }
    } else {
        // ---- scalar epilogue: lane <-> pixel (32 consecutive pixels of a map per store)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int p = wave * 64 + pt * 32 + l31;
            const int per = g.TH * g.Wo;
            const int pp = p < g.TP ? p : 0;
            const int ni = pp / per, rem = pp - ni * per;
            const int r = rem / g.Wo, col = rem - r * g.Wo;
            if (!(p < g.TP && n0 + ni < g.N && r0 + r < g.Ho)) continue;
            const size_t pbase = (size_t)(n0 + ni) * g.K * HoWo + (r0 + r) * g.Wo + col;
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                float pa[16];
                if (DGRAD && g.prev_a) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int k = min(kt * KBF + f * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi, g.K - 1);
                        pa[q] = g.prev_a[pbase + (size_t)k * HoWo];
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int k = kt * KBF + f * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi;
                    if (k < g.K) {
                        float v = acc[f][pt][q];
                        if (DGRAD) {
                            if (g.prev_a) v *= tn_act_grad_from_out(pa[q], g.act, g.prm);
                        } else {
                            v = tn_act_fwd(v, g.act, g.prm);
                        }
                        g.out[pbase + (size_t)k * HoWo] = v;
                    }
                }
            }
        }
    }
}

template <int FT, bool DGRAD, bool POOL>
__device__ __forceinline__ void ct_epilogue(const ConvTG& g, f32x16 (&acc)[FT][2], float* ct_smem, int kt, int n0,
                                            int r0, int lane, int wave, int l31, int hi) {
    if (g.act == TN_ACT_LEAKY) ct_epilogue_impl<FT, DGRAD, POOL, TN_ACT_LEAKY>(g, acc, ct_smem, kt, n0, r0, lane, wave, l31, hi);
    else ct_epilogue_impl<FT, DGRAD, POOL, -1>(g, acc, ct_smem, kt, n0, r0, lane, wave, l31, hi);
}
