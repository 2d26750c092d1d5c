// Elastic-distortion / noising input stage on the device.
//   tn_elastic_draws / _field / _apply : theanet/layer/inlayers.py:63-144 (one field per batch)
//   tn_deformer_transform              : extras/deformer.py:7-18 (one field per image)
// Coordinates are computed in float64 exactly as the Theano CPU path does (int64 indices +
// float32 random inputs upcast to float64); the field is tiny (2 x h x w) so fp64 is free.
// The gather kernel is the HBM-bound part: x read once, out written once.
#include <cstdlib>

#include "common.h"
#include "update_body.h"
#include "elastic_field.h"


__global__ __launch_bounds__(256) void elastic_draws_kernel(float* __restrict__ draws, int total,
                                                           uint32_t k0, uint32_t k1, uint32_t step,
                                                           const uint32_t* d_step) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= total) return;
    const uint32_t st = step + (d_step ? *d_step : 0u);
    float v[4];
    elastic_draw4(q, st, k0, k1, v);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * q + e < total) draws[4 * q + e] = v[e];
}

// one 64-lane WAVE per output pixel (4 pixels per block): the lanes split the (2s+1)^2 taps of
// the gaussian, so the 31x31 smoothing of mnist.prms is ~15 taps per lane instead of a
// 961-tap serial loop; lane 0 then applies translation / zoom / rotation / clipping.
// GEN: the draws are not read but generated -- every block fills its own LDS copy (a few Philox
// calls per thread), block 0 also stores them to draws_out: saves the separate draws launch.
template <bool GEN>
__global__ __launch_bounds__(256) void elastic_field_kernel(ElField f) {
    extern __shared__ float filt[];
    elastic_field_block<GEN>(f, filt, (int)blockIdx.x);
}

// The last launch of a training step: the momentum-SGD update of every tensor (+ the cost rider),
// and -- in the extra row blockIdx.y == n_upd -- the elastic field of the NEXT minibatch, which only
// depends on the RNG step counter (already advanced by the step's reduction launch).  The two are
// independent, so they share one kernel boundary.
__global__ __launch_bounds__(256) void step_tail_kernel(const tn_sgd_seg* __restrict__ segs, int nseg,
                                                       const float* __restrict__ d_lr, float gscale,
                                                       const float* __restrict__ rowloss, int nrow,
                                                       float cost_scale, float* __restrict__ d_cost,
                                                       int n_upd, int nbx_upd, ElField f) {
    extern __shared__ float filt[];
    if ((int)blockIdx.y == n_upd) {
        if ((int)blockIdx.x * 4 < f.h * f.w) elastic_field_block<true>(f, filt, (int)blockIdx.x);
        return;
    }
    if ((int)blockIdx.x >= nbx_upd) return;
    __shared__ float red[4];
    sgd_update_multi_block(segs, nseg, d_lr, gscale, nullptr, rowloss, nrow, cost_scale, d_cost,
                           blockIdx.x, blockIdx.y, nbx_upd, red);
}

__global__ __launch_bounds__(256) void elastic_apply_kernel(
    const float* __restrict__ x, int64_t x_row0, const int64_t* __restrict__ d_row0,
    float* __restrict__ out, long long total, int C, int hw, int w, int invert, int nearest,
    const int32_t* __restrict__ map_idx, const float* __restrict__ map_fy,
    const float* __restrict__ map_fx, float pflip, const uint8_t* __restrict__ flipmask, uint32_t k0,
    uint32_t k1, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long img = t / hw;              // n*C + c
    const int p = (int)(t - img * hw);
    const int64_t row_off = x_row0 + (d_row0 ? *d_row0 : 0);
    const float* xi = x + ((size_t)row_off * C + img) * hw;
    float v;
    if (!map_idx) {
        v = xi[p];
        if (invert) v = 1.f - v;
    } else if (nearest) {
        v = xi[map_idx[p]];
        if (invert) v = 1.f - v;
    } else {
        const int i00 = map_idx[p];
        const float fy = map_fy[p], fx = map_fx[p];
        float a = xi[i00], b = xi[i00 + 1], c = xi[i00 + w], d = xi[i00 + w + 1];
        if (invert) {
            a = 1.f - a;
            b = 1.f - b;
            c = 1.f - c;
            d = 1.f - d;
        }
        // same association as inlayers.py:134-137
        v = a * (1.f - fy) * (1.f - fx) + b * (1.f - fy) * fx + c * fy * (1.f - fx) + d * fy * fx;
    }
    if (flipmask) {
        if (flipmask[t]) v = 1.f - v;
    } else if (pflip > 0.f) {
        const uint32_t st = step + (d_step ? *d_step : 0u);
        const uint64_t e = (uint64_t)row_global0 * C * hw + (uint64_t)t;
        const uint64_t cq = e >> 2;
        const u32x4 r = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), st, TN_STREAM_FLIP, k0, k1);
        const uint32_t wd = ((e & 3) == 0) ? r.x : ((e & 3) == 1) ? r.y : ((e & 3) == 2) ? r.z : r.w;
        if (tn_u01(wd) < pflip) v = 1.f - v;
    }
    out[t] = v;
}

typedef float el_f2u __attribute__((ext_vector_type(2), aligned(4)));     // two neighbouring taps, dword-aligned
// The same gather, 4 consecutive output pixels per thread (h*w % 4 == 0): one 16-byte map load,
// one Philox call (the 4 flip draws of an aligned quad share a counter) and one 16-byte store.
// elastic_quad: the four values of pixels p .. p+3 of one (image, channel) plane xi; t = index of the quad's first
// element in the local (N, C, h, w) batch (flip mask / flip draws)
__device__ __forceinline__ void elastic_quad(const float* __restrict__ xi, int p, int w, int invert, int nearest,
                                             const int32_t* __restrict__ map_idx, const float* __restrict__ map_fy,
                                             const float* __restrict__ map_fx, float pflip,
                                             const uint8_t* __restrict__ flipmask, uint32_t k0, uint32_t k1, uint32_t step,
                                             const uint32_t* d_step, int64_t row_global0, int C, int hw, long long t,
                                             float (&v)[4]) {
    if (!map_idx) {
        const float4 q = *reinterpret_cast<const float4*>(xi + p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        const int4 mi = *reinterpret_cast<const int4*>(map_idx + p);
        const int m[4] = {mi.x, mi.y, mi.z, mi.w};
        if (nearest) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = xi[m[e]];
        } else {
            const float4 fy4 = *reinterpret_cast<const float4*>(map_fy + p);
            const float4 fx4 = *reinterpret_cast<const float4*>(map_fx + p);
            const float fy[4] = {fy4.x, fy4.y, fy4.z, fy4.w}, fx[4] = {fx4.x, fx4.y, fx4.z, fx4.w};
            float a[4], b[4], c[4], d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const el_f2u r0 = *reinterpret_cast<const el_f2u*>(xi + m[e]);              // two taps per 8-byte gather
                const el_f2u r1 = *reinterpret_cast<const el_f2u*>(xi + m[e] + w);
                a[e] = r0.x; b[e] = r0.y; c[e] = r1.x; d[e] = r1.y;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (invert) {
                    a[e] = 1.f - a[e]; b[e] = 1.f - b[e]; c[e] = 1.f - c[e]; d[e] = 1.f - d[e];
                }
                // same association as inlayers.py:134-137
                v[e] = a[e] * (1.f - fy[e]) * (1.f - fx[e]) + b[e] * (1.f - fy[e]) * fx[e] +
                       c[e] * fy[e] * (1.f - fx[e]) + d[e] * fy[e] * fx[e];
            }
        }
    }
    if (invert && (!map_idx || nearest)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f - v[e];
    }
    if (flipmask) {
        const uint32_t fm = *reinterpret_cast<const uint32_t*>(flipmask + t);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((fm >> (8 * e)) & 0xffu) v[e] = 1.f - v[e];
    } else if (pflip > 0.f) {
        const uint32_t st = step + (d_step ? *d_step : 0u);
        const uint64_t cq = ((uint64_t)row_global0 * C * hw + (uint64_t)t) >> 2;
        const u32x4 r = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), st, TN_STREAM_FLIP, k0, k1);
        if (tn_u01(r.x) < pflip) v[0] = 1.f - v[0];
        if (tn_u01(r.y) < pflip) v[1] = 1.f - v[1];
        if (tn_u01(r.z) < pflip) v[2] = 1.f - v[2];
        if (tn_u01(r.w) < pflip) v[3] = 1.f - v[3];
    }
}

__global__ __launch_bounds__(256) void elastic_apply4_kernel(
    const float* __restrict__ x, int64_t x_row0, const int64_t* __restrict__ d_row0,
    float* __restrict__ out, long long total4, int C, int hw, int w, int invert, int nearest,
    const int32_t* __restrict__ map_idx, const float* __restrict__ map_fy,
    const float* __restrict__ map_fx, float pflip, const uint8_t* __restrict__ flipmask, uint32_t k0,
    uint32_t k1, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    const long long t4 = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t4 >= total4) return;
    const long long t = t4 * 4;
    const long long img = t / hw;              // n*C + c
    const int p = (int)(t - img * hw);
    const int64_t row_off = x_row0 + (d_row0 ? *d_row0 : 0);
    const float* xi = x + ((size_t)row_off * C + img) * hw;
    float v[4];
    elastic_quad(xi, p, w, invert, nearest, map_idx, map_fy, map_fx, pflip, flipmask, k0, k1, step, d_step, row_global0, C,
                 hw, t, v);
    *reinterpret_cast<float4*>(out + t) = make_float4(v[0], v[1], v[2], v[3]);
}

// DTYPE float16, the stage feeds the first conv layer: the same values (same expressions as elastic_quad) rounded to
// halfs and stored as the c8 tensor that layer consumes ([N][C8][pixel][8 channels], channels beyond C zero) -- the fp32
// image is neither written nor read back by a packing pass.  thread = ONE pixel x the channels of an octet (a quad per
// thread ran three channels x 16 gathers in sequence on a quarter of the threads: 26 us against 18 + 12 for the two
// passes it replaces), one 16-byte store.
typedef _Float16 el_half8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void elastic_apply_c8_kernel(
    const float* __restrict__ x, int64_t x_row0, const int64_t* __restrict__ d_row0,
    _Float16* __restrict__ out, long long total, int C, int C8, int hw, int w, int invert, int nearest,
    const int32_t* __restrict__ map_idx, const float* __restrict__ map_fy,
    const float* __restrict__ map_fx, float pflip, const uint8_t* __restrict__ flipmask, uint32_t k0,
    uint32_t k1, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % hw);
    const long long pl = i / hw;
    const int o = (int)(pl % C8);
    const long long n = pl / C8;
    const int64_t row_off = x_row0 + (d_row0 ? *d_row0 : 0);
    const int m = map_idx ? map_idx[p] : p;
    const bool bil = map_idx && !nearest;
    const float fy = bil ? map_fy[p] : 0.f, fx = bil ? map_fx[p] : 0.f;
    const uint32_t st = step + (d_step ? *d_step : 0u);
    el_half8 cell;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 8 * o + e;
        float v = 0.f;
        if (c < C) {
            const long long img = n * C + c;
            const float* xi = x + ((size_t)row_off * C + img) * hw;
            if (bil) {
                // the two taps of a row are neighbours: one 8-byte load each (dword-aligned: the hardware takes it) --
                // the kernel is bound by its gather instructions (64 addresses each), 12 of them per pixel before
                const el_f2u r0 = *reinterpret_cast<const el_f2u*>(xi + m), r1 = *reinterpret_cast<const el_f2u*>(xi + m + w);
                float a = r0.x, b = r0.y, cc = r1.x, d = r1.y;
                if (invert) {
                    a = 1.f - a; b = 1.f - b; cc = 1.f - cc; d = 1.f - d;
                }
                // same association as inlayers.py:134-137
                v = a * (1.f - fy) * (1.f - fx) + b * (1.f - fy) * fx + cc * fy * (1.f - fx) + d * fy * fx;
            } else {
                v = xi[m];
                if (invert) v = 1.f - v;
            }
            const long long t = img * hw + p;              // element of the local (N, C, h, w) batch
            if (flipmask) {
                if (flipmask[t]) v = 1.f - v;
            } else if (pflip > 0.f) {                      // the four draws of an aligned quad share one Philox call
                const uint64_t cq = ((uint64_t)row_global0 * C * hw + (uint64_t)t) >> 2;
                const u32x4 r = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), st, TN_STREAM_FLIP, k0, k1);
                const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
                if (tn_u01(rw[t & 3]) < pflip) v = 1.f - v;
            }
        }
        cell[e] = (_Float16)v;
    }
    reinterpret_cast<el_half8*>(out)[(n * C8 + o) * hw + p] = cell;
}

// ---- ElasticLayer resampling fused into the first conv block's forward --------------------------
// One block per single-channel image: the 4-wide gather of elastic_apply4_kernel fills a zero-padded
// LDS tile (and writes the resampled image out once -- the conv backward reads it), then one thread
// per pooling window runs conv(3x3) + bias + act + 2x2 max-pool + pooling mask of convpool.hip on
// the tile.  The resampled image is not read back from HBM and a kernel boundary disappears.
// KT > 0: exactly KT filters, their taps are fetched (scalar loads) before the resampling phase and
// the filter loop is unrolled; KT == 0: any K <= 16, taps fetched per filter.
template <int ACT, int KT>
__global__ __launch_bounds__(256) void elastic_convpool_fwd_kernel(
    const float* __restrict__ x, int64_t x_row0, const int64_t* __restrict__ d_row0,
    float* __restrict__ xd, int N, int h, int w, int invert, int nearest,
    const int32_t* __restrict__ map_idx, const float* __restrict__ map_fy,
    const float* __restrict__ map_fx, float pflip, const uint8_t* __restrict__ flipmask, uint32_t k0,
    uint32_t k1, uint32_t step, const uint32_t* d_step, int64_t row_global0,
    const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ y,
    uint8_t* __restrict__ mask, int K, int pad, int Ho, int Wo, int Hp, int Wp, int act, float prm) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [2Hp+2][2Wp+2], zero padded
    const int Hx = 2 * Hp + 2, Wx = 2 * Wp + 2, hw = h * w;
    const int n = blockIdx.x;
    float wk[KT > 0 ? KT * 9 : 1], bk[KT > 0 ? KT : 1];
    if (KT > 0) {
#pragma unroll
        for (int i = 0; i < KT * 9; ++i) wk[i] = W[i];
#pragma unroll
        for (int i = 0; i < KT; ++i) bk[i] = b[i];
    }
    if (Hx != h || Wx != w) {          // block-uniform: a tile larger than the image needs its zero frame
        for (int i = threadIdx.x * 4; i < Hx * Wx; i += 1024)
            *reinterpret_cast<float4*>(tile + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    const int64_t row_off = x_row0 + (d_row0 ? *d_row0 : 0);
    const float* xi = x + (size_t)(row_off + n) * hw;
    const uint32_t st = step + (d_step ? *d_step : 0u);
    for (int q4 = threadIdx.x; 4 * q4 < hw; q4 += 256) {
        const int p = 4 * q4;
        float v[4];
        if (!map_idx) {
            const float4 q = *reinterpret_cast<const float4*>(xi + p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            const int4 mi = *reinterpret_cast<const int4*>(map_idx + p);
            const int m[4] = {mi.x, mi.y, mi.z, mi.w};
            if (nearest) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xi[m[e]];
            } else {
                const float4 fy4 = *reinterpret_cast<const float4*>(map_fy + p);
                const float4 fx4 = *reinterpret_cast<const float4*>(map_fx + p);
                const float fy[4] = {fy4.x, fy4.y, fy4.z, fy4.w}, fx[4] = {fx4.x, fx4.y, fx4.z, fx4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const el_f2u r0 = *reinterpret_cast<const el_f2u*>(xi + m[e]);          // two taps per 8-byte gather
                    const el_f2u r1 = *reinterpret_cast<const el_f2u*>(xi + m[e] + w);
                    float a = r0.x, bb = r0.y, c = r1.x, d = r1.y;
                    if (invert) {
                        a = 1.f - a; bb = 1.f - bb; c = 1.f - c; d = 1.f - d;
                    }
                    // same association as inlayers.py:134-137
                    v[e] = a * (1.f - fy[e]) * (1.f - fx[e]) + bb * (1.f - fy[e]) * fx[e] +
                           c * fy[e] * (1.f - fx[e]) + d * fy[e] * fx[e];
                }
            }
        }
        if (invert && (!map_idx || nearest)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 1.f - v[e];
        }
        const size_t t = (size_t)n * hw + p;                 // element index within the local batch
        if (flipmask) {
            const uint32_t fm = *reinterpret_cast<const uint32_t*>(flipmask + t);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((fm >> (8 * e)) & 0xffu) v[e] = 1.f - v[e];
        } else if (pflip > 0.f) {
            const uint64_t cq = ((uint64_t)row_global0 * hw + (uint64_t)t) >> 2;
            const u32x4 r = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), st, TN_STREAM_FLIP, k0, k1);
            if (tn_u01(r.x) < pflip) v[0] = 1.f - v[0];
            if (tn_u01(r.y) < pflip) v[1] = 1.f - v[1];
            if (tn_u01(r.z) < pflip) v[2] = 1.f - v[2];
            if (tn_u01(r.w) < pflip) v[3] = 1.f - v[3];
        }
        *reinterpret_cast<float4*>(xd + t) = make_float4(v[0], v[1], v[2], v[3]);
        const int yy = p / w, xx = p - yy * w;               // w % 4 == 0: the 4 pixels share a row
        float* dst = tile + (yy + pad) * Wx + xx + pad;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
    __syncthreads();
    const int HpWp = Hp * Wp;
    for (int q = threadIdx.x; q < HpWp; q += 256) {
        const int pi = q / Wp, pj = q - pi * Wp;
        float pt[4][4];
        const float* base = tile + (2 * pi) * Wx + 2 * pj;     // 8-byte aligned: Wx even
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 lo = *reinterpret_cast<const float2*>(base + r * Wx);
            const float2 hi = *reinterpret_cast<const float2*>(base + r * Wx + 2);
            pt[r][0] = lo.x; pt[r][1] = lo.y; pt[r][2] = hi.x; pt[r][3] = hi.y;
        }
        const bool v01 = 2 * pj + 1 < Wo, v10 = 2 * pi + 1 < Ho;
        const bool valid[2][2] = {{true, v01}, {v10, v10 && v01}};
#pragma unroll
        for (int k = 0; k < (KT > 0 ? KT : K); ++k) {
            const float* Wk = W + (size_t)k * 9;              // block-uniform: scalar loads
            const float bias = KT > 0 ? bk[k] : b[k];
            float z[2][2] = {{bias, bias}, {bias, bias}};
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int vv = 0; vv < 3; ++vv) {
                    const float wt = KT > 0 ? wk[k * 9 + (2 - u) * 3 + (2 - vv)] : Wk[(2 - u) * 3 + (2 - vv)];
#pragma unroll
                    for (int di = 0; di < 2; ++di)
#pragma unroll
                        for (int dj = 0; dj < 2; ++dj) z[di][dj] = fmaf(pt[di + u][dj + vv], wt, z[di][dj]);
                }
            float m = -INFINITY;
#pragma unroll
            for (int di = 0; di < 2; ++di)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj) {
                    z[di][dj] = (ACT == TN_ACT_LEAKY) ? fmaxf(0.f, z[di][dj]) + fminf(0.f, z[di][dj]) * prm
                                                      : tn_act_fwd(z[di][dj], act, prm);
                    m = valid[di][dj] ? fmaxf(m, z[di][dj]) : m;
                }
            const size_t o = ((size_t)n * K + k) * HpWp + q;
            y[o] = m;
            if (mask) {
                unsigned bits = (m > 0.f ? 16u : 0u) | (m < 0.f ? 32u : 0u);
#pragma unroll
                for (int di = 0; di < 2; ++di)
#pragma unroll
                    for (int dj = 0; dj < 2; ++dj)
                        bits |= (valid[di][dj] && z[di][dj] == m) ? (1u << (di * 2 + dj)) : 0u;
                mask[o] = (uint8_t)bits;
            }
        }
    }
}

// ---- extras/deformer.py:7-18, one block per image, float64 like scipy -------------------
__global__ __launch_bounds__(256) void deformer_kernel(const float* __restrict__ imgs,
                                                      float* __restrict__ out, int h, int w,
                                                      double scale, double sigma, double cval,
                                                      const float* __restrict__ noise, uint32_t k0,
                                                      uint32_t k1, int64_t img_global0) {
    extern __shared__ double sm[];
    const int hw = h * w;
    double* tr = sm;              // [2][hw]
    double* tmp = sm + 2 * hw;    // [2][hw]
    double* kern = sm + 4 * hw;   // [2r+1]
    const int r = (int)(2.0 * sigma + 0.5);
    const int img = blockIdx.x;
    for (int t = threadIdx.x; t <= 2 * r; t += 256) {
        const double d = t - r;
        kern[t] = exp(-0.5 / (sigma * sigma) * d * d);
    }
    __syncthreads();
    double ksum = 0.0;
    for (int t = 0; t <= 2 * r; ++t) ksum += kern[t];
    __syncthreads();
    for (int t = threadIdx.x; t <= 2 * r; t += 256) kern[t] /= ksum;
    for (int t = threadIdx.x; t < 2 * hw; t += 256) {
        const int a = t / hw, p = t - a * hw;
        double u;
        if (noise) {
            u = (double)noise[(size_t)img * 2 * hw + t];
        } else {
            const uint64_t e = (uint64_t)(img_global0 + img) * 2 * hw + t;
            const u32x4 q = philox4x32((uint32_t)e, (uint32_t)(e >> 32), 0u, TN_STREAM_DEFORMER, k0, k1);
            u = -1.0 + 2.0 * (double)tn_u01(q.x);
        }
        tr[t] = (double)(a == 0 ? p / w : p % w) + scale * u;
    }
    __syncthreads();
    // axis 0 (rows), edge replicated
    for (int t = threadIdx.x; t < 2 * hw; t += 256) {
        const int a = t / hw, p = t - a * hw, y = p / w, x = p - y * w;
        double s = 0.0;
        for (int k = -r; k <= r; ++k) {
            const int yy = min(max(y + k, 0), h - 1);
            s += kern[k + r] * tr[a * hw + yy * w + x];
        }
        tmp[t] = s;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * hw; t += 256) {
        const int a = t / hw, p = t - a * hw, y = p / w, x = p - y * w;
        double s = 0.0;
        for (int k = -r; k <= r; ++k) {
            const int xx = min(max(x + k, 0), w - 1);
            s += kern[k + r] * tmp[a * hw + y * w + xx];
        }
        tr[t] = s;
    }
    __syncthreads();
    const float* im = imgs + (size_t)img * hw;
    for (int p = threadIdx.x; p < hw; p += 256) {
        const double cy = tr[p], cx = tr[hw + p];
        double v;
        if (cy < 0.0 || cy > h - 1 || cx < 0.0 || cx > w - 1) {
            v = cval;
        } else {
            const int y0 = (int)floor(cy), x0 = (int)floor(cx);
            const double fy = cy - y0, fx = cx - x0;
            auto tap = [&](int yy, int xx) -> double {
                return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (double)im[yy * w + xx] : cval;
            };
            v = tap(y0, x0) * (1 - fy) * (1 - fx) + tap(y0, x0 + 1) * (1 - fy) * fx +
                tap(y0 + 1, x0) * fy * (1 - fx) + tap(y0 + 1, x0 + 1) * fy * fx;
        }
        out[(size_t)img * hw + p] = (float)v;
    }
}

extern "C" {

size_t tn_elastic_draws_count(int h, int w) { return (size_t)EL_HDR + 2 * (size_t)h * w; }

int tn_elastic_draws(tn_ctx* ctx, float* draws, int h, int w, uint64_t seed, uint32_t step,
                     const uint32_t* d_step) {
    const int total = (int)tn_elastic_draws_count(h, w);
    elastic_draws_kernel<<<cdiv(cdiv(total, 4), 256), 256, 0, ctx->stream>>>(
        draws, total, (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_elastic_field(tn_ctx* ctx, const float* draws, int h, int w, double translation, double zoom,
                     double magnitude, int sigma, double angle, int nearest, int32_t* map_idx,
                     float* map_fy, float* map_fx, double* target) {
    TN_REQUIRE(h > 0 && w > 0 && zoom > 0 && sigma >= 0 && map_idx != nullptr,
               "tn_elastic_field: bad arguments");
    TN_REQUIRE(nearest || (map_fy && map_fx), "tn_elastic_field: bilinear needs map_fy/map_fx");
    const int ks = 2 * sigma + 1;
    const size_t lds = ((size_t)ks * ks + 16) * sizeof(float);      // filter table + 4 doubles
    TN_REQUIRE(lds <= 64 * 1024, "tn_elastic_field: sigma %d too large", sigma);
    ElField f{draws, nullptr, 0u, 0u, 0u, nullptr, h, w, translation, zoom, magnitude, sigma, angle, nearest,
              map_idx, map_fy, map_fx, target};
    elastic_field_kernel<false><<<cdiv(h * w, 4), 256, lds, ctx->stream>>>(f);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_elastic_field_gen(tn_ctx* ctx, float* draws_out, uint64_t seed, uint32_t step,
                         const uint32_t* d_step, int h, int w, double translation, double zoom,
                         double magnitude, int sigma, double angle, int nearest, int32_t* map_idx,
                         float* map_fy, float* map_fx, double* target) {
    TN_REQUIRE(h > 0 && w > 0 && zoom > 0 && sigma >= 0 && map_idx != nullptr,
               "tn_elastic_field_gen: bad arguments");
    TN_REQUIRE(nearest || (map_fy && map_fx), "tn_elastic_field_gen: bilinear needs map_fy/map_fx");
    const int ks = 2 * sigma + 1;
    const size_t lds = ((size_t)ks * ks + tn_elastic_draws_count(h, w) + 16) * sizeof(float);
    if (lds > 64 * 1024) {      // big images: two launches
        TN_REQUIRE(draws_out != nullptr, "tn_elastic_field_gen: %dx%d needs a draws buffer", h, w);
        int rc = tn_elastic_draws(ctx, draws_out, h, w, seed, step, d_step);
        if (rc) return rc;
        return tn_elastic_field(ctx, draws_out, h, w, translation, zoom, magnitude, sigma, angle,
                                nearest, map_idx, map_fy, map_fx, target);
    }
    ElField f{nullptr, draws_out, (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, h, w, translation,
              zoom, magnitude, sigma, angle, nearest, map_idx, map_fy, map_fx, target};
    elastic_field_kernel<true><<<cdiv(h * w, 4), 256, lds, ctx->stream>>>(f);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// ---- rider: the field of the NEXT minibatch waits in the context for a heavy launch to ride in ----
int tn_rider_elastic_field(tn_ctx* ctx, float* draws_out, uint64_t seed, uint32_t step,
                           const uint32_t* d_step, int h, int w, double translation, double zoom,
                           double magnitude, int sigma, double angle, int nearest, int32_t* map_idx,
                           float* map_fy, float* map_fx, double* target) {
    TN_REQUIRE(h > 0 && w > 0 && zoom > 0 && sigma >= 0 && map_idx != nullptr,
               "tn_rider_elastic_field: bad arguments");
    TN_REQUIRE(nearest || (map_fy && map_fx), "tn_rider_elastic_field: bilinear needs map_fy/map_fx");
    const int ks = 2 * sigma + 1;
    const size_t lds = ((size_t)ks * ks + tn_elastic_draws_count(h, w) + 16) * sizeof(float);
    if (lds > 24 * 1024)        // too big to sit beside a GEMM tile: run it now
        return tn_elastic_field_gen(ctx, draws_out, seed, step, d_step, h, w, translation, zoom, magnitude,
                                    sigma, angle, nearest, map_idx, map_fy, map_fx, target);
    ctx->rider = ElField{nullptr, draws_out, (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, h, w,
                         translation, zoom, magnitude, sigma, angle, nearest, map_idx, map_fy, map_fx, target};
    ctx->rider_lds = lds;
    ctx->rider_valid = true;
    return TN_OK;
}

int tn_rider_pending(tn_ctx* ctx) { return ctx->rider_valid ? 1 : 0; }

int tn_rider_cancel(tn_ctx* ctx) {
    ctx->rider_valid = false;
    return TN_OK;
}

int tn_step_tail(tn_ctx* ctx, const tn_sgd_seg* d_segs, int nseg, size_t max_n, const float* d_lr,
                 float gscale, const float* rowloss, int nrow, float cost_scale, float* d_cost,
                 float* draws_out, uint64_t seed, const uint32_t* d_step, int h, int w,
                 double translation, double zoom, double magnitude, int sigma, double angle, int nearest,
                 int32_t* map_idx, float* map_fy, float* map_fx, double* target) {
    TN_REQUIRE(h > 0 && w > 0 && zoom > 0 && sigma >= 0 && map_idx != nullptr && d_step != nullptr,
               "tn_step_tail: bad field arguments");
    TN_REQUIRE(nearest || (map_fy && map_fx), "tn_step_tail: bilinear needs map_fy/map_fx");
    TN_REQUIRE(nseg <= 0 || (d_segs != nullptr && d_lr != nullptr), "tn_step_tail: NULL update argument");
    const bool rider = rowloss != nullptr;
    TN_REQUIRE(!rider || (d_cost != nullptr && nrow > 0), "tn_step_tail: bad cost arguments");
    if (nseg < 0) nseg = 0;
    const int ks = 2 * sigma + 1;
    const size_t lds = ((size_t)ks * ks + tn_elastic_draws_count(h, w) + 16) * sizeof(float);
    if (lds > 48 * 1024) {      // the field does not fit beside the update: two launches
        int rc = tn_sgd_update_net(ctx, TN_UPD_PLAIN, d_segs, nullptr, nseg, max_n, d_lr, gscale, nullptr, 0, 0, rowloss, nrow,
                                          cost_scale, d_cost);
        if (rc) return rc;
        return tn_elastic_field_gen(ctx, draws_out, seed, 0, d_step, h, w, translation, zoom, magnitude,
                                    sigma, angle, nearest, map_idx, map_fy, map_fx, target);
    }
    int bx = cdiv(max_n, 1024);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    const int n_upd = nseg + (rider ? 1 : 0);
    const int gx = bx > cdiv(h * w, 4) ? bx : cdiv(h * w, 4);
    ElField f{nullptr, draws_out, (uint32_t)seed, (uint32_t)(seed >> 32), 0u, d_step, h, w, translation,
              zoom, magnitude, sigma, angle, nearest, map_idx, map_fy, map_fx, target};
    step_tail_kernel<<<dim3(gx, n_upd + 1), 256, lds, ctx->stream>>>(d_segs, nseg, d_lr, gscale, rowloss,
                                                                    nrow, cost_scale, d_cost, n_upd, bx, f);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_elastic_apply(tn_ctx* ctx, const float* x, int64_t x_row0, const int64_t* d_row0, float* out,
                     int N, int C, int h, int w, int invert, int nearest, const int32_t* map_idx,
                     const float* map_fy, const float* map_fx, float pflip, const uint8_t* flipmask,
                     uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    TN_REQUIRE(N > 0 && C > 0 && h > 0 && w > 0, "tn_elastic_apply: bad shape");
    const long long total = (long long)N * C * h * w;
    const bool al = (((uintptr_t)x | (uintptr_t)out | (uintptr_t)map_idx | (uintptr_t)map_fy |
                      (uintptr_t)map_fx) & 15) == 0 && ((uintptr_t)flipmask & 3) == 0;
    if ((h * w) % 4 == 0 && al) {
        elastic_apply4_kernel<<<cdiv(total / 4, 256), 256, 0, ctx->stream>>>(
            x, x_row0, d_row0, out, total / 4, C, h * w, w, invert, nearest, map_idx, map_fy, map_fx,
            pflip, flipmask, (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, row_global0);
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    elastic_apply_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(
        x, x_row0, d_row0, out, total, C, h * w, w, invert, nearest, map_idx, map_fy, map_fx, pflip,
        flipmask, (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, row_global0);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// tn_elastic_apply with the c8 fp16 tensor of the first conv layer as its output (DTYPE float16; include/theanet_hip.h)
int tn_c8_elastic_apply(tn_ctx* ctx, const float* x, int64_t x_row0, const int64_t* d_row0, void* out16,
                        int N, int C, int h, int w, int invert, int nearest, const int32_t* map_idx,
                        const float* map_fy, const float* map_fx, float pflip, const uint8_t* flipmask,
                        uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    TN_REQUIRE(N > 0 && C > 0 && h > 0 && w > 0 && x && out16, "tn_c8_elastic_apply: bad arguments");
    const bool al = (((uintptr_t)x | (uintptr_t)out16 | (uintptr_t)map_idx | (uintptr_t)map_fy | (uintptr_t)map_fx) & 15) == 0 &&
                    ((uintptr_t)flipmask & 3) == 0;
    TN_REQUIRE((h * w) % 4 == 0 && al, "tn_c8_elastic_apply: maps of %d x %d pixels / unaligned operands", h, w);
    const int C8 = (C + 7) / 8;
    const long long total = (long long)N * C8 * (h * w);
    elastic_apply_c8_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(
        x, x_row0, d_row0, static_cast<_Float16*>(out16), total, C, C8, h * w, w, invert, nearest, map_idx, map_fy, map_fx,
        pflip, flipmask, (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, row_global0);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_elastic_convpool_supported(int h, int w, int K, int f, int pad_lo, int Ho, int Wo, int p, int Hp,
                                  int Wp) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("TN_ELASTIC_CONV");
        on = e ? atoi(e) : 1;
    }
    if (!on || f != 3 || p != 2 || K < 1 || K > 16 || w % 4 != 0) return 0;
    if (h + 2 * pad_lo > 2 * Hp + 2 || w + 2 * pad_lo > 2 * Wp + 2) return 0;      // image fits the tile
    if (Hp != (Ho + 1) / 2 || Wp != (Wo + 1) / 2) return 0;
    return (size_t)(2 * Hp + 2) * (2 * Wp + 2) * sizeof(float) <= 48 * 1024;
}

int tn_elastic_convpool_fwd_mask(tn_ctx* ctx, const float* x, int64_t x_row0, const int64_t* d_row0,
                                 float* xd, int N, int h, int w, int invert, int nearest,
                                 const int32_t* map_idx, const float* map_fy, const float* map_fx,
                                 float pflip, const uint8_t* flipmask, uint64_t seed, uint32_t step,
                                 const uint32_t* d_step, int64_t row_global0, const float* W,
                                 const float* b, float* y, uint8_t* mask, int K, int f, int pad_lo, int Ho,
                                 int Wo, int p, int Hp, int Wp, int act, float act_param) {
    TN_REQUIRE(tn_elastic_convpool_supported(h, w, K, f, pad_lo, Ho, Wo, p, Hp, Wp),
               "tn_elastic_convpool_fwd_mask: unsupported shape %dx%d K=%d f=%d p=%d", h, w, K, f, p);
    TN_REQUIRE(N > 0 && x && xd && W && b && y, "tn_elastic_convpool_fwd_mask: bad arguments");
    const size_t lds = (((size_t)(2 * Hp + 2) * (2 * Wp + 2) + 3) & ~(size_t)3) * sizeof(float);
#define EC_GO(ACT_)                                                                              \
    if (K == 4)                                                                                  \
        elastic_convpool_fwd_kernel<ACT_, 4><<<N, 256, lds, ctx->stream>>>(                      \
            x, x_row0, d_row0, xd, N, h, w, invert, nearest, map_idx, map_fy, map_fx, pflip, flipmask, \
            (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, row_global0, W, b, y, mask, K, pad_lo, \
            Ho, Wo, Hp, Wp, act, act_param);                                                     \
    else                                                                                         \
    elastic_convpool_fwd_kernel<ACT_, 0><<<N, 256, lds, ctx->stream>>>(                           \
        x, x_row0, d_row0, xd, N, h, w, invert, nearest, map_idx, map_fy, map_fx, pflip, flipmask, \
        (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, row_global0, W, b, y, mask, K, pad_lo, \
        Ho, Wo, Hp, Wp, act, act_param)
    if (act == TN_ACT_LEAKY) EC_GO(TN_ACT_LEAKY); else EC_GO(-1);
#undef EC_GO
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_deformer_transform(tn_ctx* ctx, const float* imgs, float* out, int N, int h, int w,
                          double scale, double sigma, double cval, const float* noise, uint64_t seed,
                          int64_t img_global0) {
    TN_REQUIRE(N > 0 && h > 0 && w > 0 && sigma > 0, "tn_deformer_transform: bad arguments");
    const int r = (int)(2.0 * sigma + 0.5);
    const size_t lds = ((size_t)4 * h * w + 2 * r + 1) * sizeof(double);
    TN_REQUIRE(lds <= 160 * 1024, "tn_deformer_transform: image %dx%d (sigma %g) exceeds LDS", h, w,
               sigma);
    if (lds > 64 * 1024) {
        TN_HIP(hipFuncSetAttribute((const void*)deformer_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    deformer_kernel<<<N, 256, lds, ctx->stream>>>(imgs, out, h, w, scale, sigma, cval, noise,
                                                  (uint32_t)seed, (uint32_t)(seed >> 32), img_global0);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
