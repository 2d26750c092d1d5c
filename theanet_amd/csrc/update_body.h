// Body of the multi-tensor momentum-SGD launch (update.hip) -- shared with the step-tail kernel of
// elastic.hip, which runs it side by side with the next minibatch's elastic field.
#pragma once
#include "common.h"

// one block of the update grid: by = segment index (by == nseg: the cost rider), bx / nbx = the
// block's position along the segment
__device__ __forceinline__ void sgd_update_multi_block(const tn_sgd_seg* __restrict__ segs, int nseg,
                                                       const float* __restrict__ d_lr, float gscale,
                                                       uint32_t* d_step_inc,
                                                       const float* __restrict__ rowloss, int nrow,
                                                       float cost_scale, float* __restrict__ d_cost,
                                                       int bx, int by, int nbx, float* red) {
    if (d_step_inc && bx == 0 && by == 0 && threadIdx.x == 0)
        *d_step_inc += 1;                       // the RNG step counter advances with the update
    if (by == nseg) {
        // rider: the minibatch cost = cost_scale * sum(rowloss), summed in a fixed order by ONE block
        // (saves the separate reduction launch of the step)
        if (bx != 0) return;
        float s = 0.f;
        for (int i = threadIdx.x; i < nrow; i += 256) s += rowloss[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) d_cost[0] = cost_scale * ((red[0] + red[1]) + (red[2] + red[3]));
        return;
    }
    const tn_sgd_seg sg = segs[by];
    const float step = sg.rate * d_lr[0];
    float* __restrict__ p = sg.p;
    float* __restrict__ v = sg.v;
    const float* __restrict__ g = sg.g;
    const size_t n = sg.n;
    const float m = sg.momentum, L1 = sg.L1, L2 = sg.L2;
    for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n; i += (size_t)nbx * 256) {
        const float pv = p[i], vv = v[i];
        float gg = g[i] * gscale;
        if (L1 != 0.f) gg += L1 * ((pv > 0.f) - (pv < 0.f));
        if (L2 != 0.f) gg += 2.f * L2 * pv;
        v[i] = tn_vel(m, vv, gg);
        p[i] = tn_stepped(pv, step, vv);
    }
}
