// The elastic displacement-field computation as a device function (inlayers.py:72-125), shared by
// its launchers: the stand-alone field kernels and the step tail of elastic.hip, and the paired GEMM
// launch of gemm.hip, which can carry it as a rider (tn_rider_elastic_field).
#pragma once
#include "common.h"

#define EL_HDR 8   // draws: [0:2] transln, [2:4] origin, [4:6] zoom, [6] theta, [7] pad, [8:] noise

// draws quad q = elements 4q .. 4q+3 from ONE Philox call: header elements are uniforms of one word
// each; the noise planes take both Box-Muller outputs of the word pairs (x,y) and (z,w)
__device__ __forceinline__ void elastic_draw4(int q, uint32_t st, uint32_t k0, uint32_t k1, float (&v)[4]) {
    const u32x4 r = philox4x32((uint32_t)q, 0u, st, TN_STREAM_ELASTIC, k0, k1);
    const uint32_t wd[4] = {r.x, r.y, r.z, r.w};
    if (4 * q < EL_HDR) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            const float u = tn_u01(wd[e]);
            v[e] = (i == 2 || i == 3) ? .25f + .5f * u : -1.f + 2.f * u;     // origin U(.25,.75), else U(-1,1)
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((wd[2 * h] >> 8) + 1) * (1.0f / 16777216.0f);
        const float a = 6.28318530717958647692f * tn_u01(wd[2 * h + 1]);
        const float rad = sqrtf(-2.f * logf(u1));
        v[2 * h] = rad * cosf(a);
        v[2 * h + 1] = rad * sinf(a);
    }
}

template <bool GEN>
__device__ __forceinline__ void elastic_field_block(const ElField& f, float* filt, int bx) {
    const float* __restrict__ draws_in = f.draws_in;
    float* __restrict__ draws_out = f.draws_out;
    const uint32_t k0 = f.k0, k1 = f.k1, step = f.step;
    const uint32_t* __restrict__ d_step = f.d_step;
    const int h = f.h, w = f.w, sigma = f.sigma, nearest = f.nearest;
    const double translation = f.translation, zoom = f.zoom, magnitude = f.magnitude, angle = f.angle;
    int32_t* __restrict__ map_idx = f.map_idx;
    float* __restrict__ map_fy = f.map_fy;
    float* __restrict__ map_fx = f.map_fx;
    double* __restrict__ target = f.target;
    // filt: (2s+1)^2 floats, float32 like the reference's filter [+ the draws]
    const int ks = 2 * sigma + 1;
    const float* draws = draws_in;
    if (GEN) {
        float* sd = filt + ks * ks;
        const uint32_t st = step + (d_step ? *d_step : 0u);
        const int total = EL_HDR + 2 * h * w;
        for (int q4 = threadIdx.x; 4 * q4 < total; q4 += 256) {
            float v[4];
            elastic_draw4(q4, st, k0, k1, v);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q4 + e < total) {
                    sd[4 * q4 + e] = v[e];
                    if (draws_out && bx == 0) draws_out[4 * q4 + e] = v[e];
                }
        }
        draws = sd;
    }
    // The zoom factors and the rotation (double-precision exp / cos / sin of three header draws) are
    // the same for every pixel: four threads work them out while the others fill the filter table,
    // instead of every wave's lane 0 doing all four after its smoothing sum.
    double* aux = reinterpret_cast<double*>(filt + ((ks * ks + (GEN ? EL_HDR + 2 * h * w : 0) + 1) & ~1));
    if (threadIdx.x >= 192 && threadIdx.x < 196 && (zoom != 1.0 || angle != 0.0)) {
        float hd[4];                               // draws[4..7]: zoom u (2), theta u, pad
        if (GEN) {
            elastic_draw4(1, step + (d_step ? *d_step : 0u), k0, k1, hd);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) hd[e] = draws_in[4 + e];
        }
        const int j = threadIdx.x - 192;
        double r;
        if (j < 2) {
            r = zoom != 1.0 ? exp(log(zoom) * (double)hd[j]) : 1.0;
        } else {
            const double theta = (angle * 3.14159265358979323846 / 180.0) * (double)hd[2];
            r = j == 2 ? cos(theta) : sin(theta);
        }
        aux[j] = r;
    }
    if (magnitude != 0.0) {
        const double var = (double)sigma * sigma;
        const float norm = (float)(2.0 * 3.14159265358979323846 * var);
        for (int t = threadIdx.x; t < ks * ks; t += 256) {
            const int i = t % ks - sigma, j = t / ks - sigma;
            filt[t] = (float)exp(-.5 * (i * i + j * j) / var) / norm;
        }
    }
    __syncthreads();
    const int p = bx * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    double ty = y, tx = x;
    if (translation != 0.0) {
        ty += (double)((float)translation * draws[0]);
        tx += (double)((float)translation * draws[1]);
    }
    if (magnitude != 0.0) {
        const float* n0 = draws + EL_HDR;
        const float* n1 = n0 + h * w;
        const float mag = (float)magnitude;
        // float32 products, float64 accumulation, rounded to float32 at the end: independent
        // of the summation order, so it reproduces the oracle bit for bit
        double s0 = 0.0, s1 = 0.0;
        for (int t = lane; t < ks * ks; t += 64) {
            const int u = t / ks, v = t - u * ks;
            const int yy = y + u - sigma, xx = x + v - sigma;
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            const float fw = filt[t];   // symmetric: convolution == correlation
            s0 += (double)fw * (double)(mag * n0[yy * w + xx]);
            s1 += (double)fw * (double)(mag * n1[yy * w + xx]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64);
            s1 += __shfl_xor(s1, o, 64);
        }
        ty += (double)(float)s0;
        tx += (double)(float)s1;
    }
    if (lane != 0) return;
    if (zoom != 1.0 || angle != 0.0) {
        const double oy = (double)draws[2] * h, ox = (double)draws[3] * w;
        ty -= oy;
        tx -= ox;
        if (zoom != 1.0) {
            ty *= aux[0];          // exp(log(zoom) * draws[4])
            tx *= aux[1];          // exp(log(zoom) * draws[5])
        }
        if (angle != 0.0) {
            const double c = aux[2], s = aux[3];      // cos / sin(angle * pi/180 * draws[6])
            // tensordot(R, target, axes=(0,0)) with R=[[c,-s],[s,c]] -> R^T applied
            const double ry = c * ty + s * tx;
            const double rx = -s * ty + c * tx;
            ty = ry;
            tx = rx;
        }
        ty += oy;
        tx += ox;
    }
    if (target) {
        target[p] = ty;
        target[h * w + p] = tx;
    }
    const double cy = fmin(fmax(ty, 0.0), (double)h - 1 - .001);
    const double cx = fmin(fmax(tx, 0.0), (double)w - 1 - .001);
    if (nearest) {
        map_idx[p] = (int)rint(cy) * w + (int)rint(cx);
    } else {
        const int top = (int)cy, left = (int)cx;
        map_idx[p] = top * w + left;
        map_fy[p] = (float)(cy - top);
        map_fx[p] = (float)(cx - left);
    }
}

