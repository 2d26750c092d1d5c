// Momentum SGD (+L1/L2 gradient terms) and max-norm projection.
// Semantics: theanet/layer/layer.py:70-107 -- simultaneous Theano updates:
//   v' = m*v + (1-m)*g ;  p' = p - rate*lr*v   (the OLD velocity moves p) ; maxnorm(p').
#include "common.h"
#include "update_body.h"

__global__ __launch_bounds__(256) void sgd_update_kernel(float* __restrict__ p, float* __restrict__ v,
                                                        const float* __restrict__ g, size_t n,
                                                        float momentum, float rate,
                                                        const float* __restrict__ d_lr, float L1,
                                                        float L2, float gscale) {
    const float step = rate * d_lr[0];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const float pv = p[i], vv = v[i];
        float gg = g[i] * gscale;
        if (L1 != 0.f) gg += L1 * ((pv > 0.f) - (pv < 0.f));
        if (L2 != 0.f) gg += 2.f * L2 * pv;
        v[i] = tn_vel(momentum, vv, gg);
        p[i] = tn_stepped(pv, step, vv);
    }
}

// all parameter tensors of the net in ONE launch: blockIdx.y selects the segment descriptor
__global__ __launch_bounds__(256) void sgd_update_multi_kernel(const tn_sgd_seg* __restrict__ segs,
                                                              int nseg, const float* __restrict__ d_lr,
                                                              float gscale, uint32_t* d_step_inc,
                                                              const float* __restrict__ rowloss,
                                                              int nrow, float cost_scale,
                                                              float* __restrict__ d_cost) {
    __shared__ float red[4];
    sgd_update_multi_block(segs, nseg, d_lr, gscale, d_step_inc, rowloss, nrow, cost_scale, d_cost,
                           blockIdx.x, blockIdx.y, gridDim.x, red);
}

// Data-parallel "delayed" schedule (see NeuralNet._train_step): the reference applies the OLD velocity
// (layer.py:82-86: v' = m v + (1-m) g ; p' = p - rate*lr*v), so the weights of step t+1 do not depend
// on the gradient of step t -- the all-reduce of g_t may take the whole of step t+1.  At the end of
// step t the stored velocity is still v_{t-1} and seg.g points at the REDUCED gradient of step t-1:
//   mode 1: v = m v + (1-m) G_{t-1} (= v_t) ; p = p - step * v      -- the steady state
//   mode 2: p = p - step * v                                        -- first delayed step (v is v_t already)
//   mode 3: v = m v + (1-m) G_{t-1}                                 -- leaving the schedule: catch v up
// Same expressions as sgd_update_multi_block, so the weight trajectory is bit-identical.
__global__ __launch_bounds__(256) void sgd_update_delayed_kernel(const tn_sgd_seg* __restrict__ segs, int nseg,
                                                                const float* __restrict__ d_lr, float gscale,
                                                                uint32_t* d_step_inc, int mode) {
    const int bx = blockIdx.x, by = blockIdx.y, nbx = gridDim.x;
    if (d_step_inc && bx == 0 && by == 0 && threadIdx.x == 0) *d_step_inc += 1;
    const tn_sgd_seg sg = segs[by];
    const float step = sg.rate * d_lr[0];
    float* __restrict__ p = sg.p;
    float* __restrict__ v = sg.v;
    const float* __restrict__ g = sg.g;
    const size_t n = sg.n;
    const float m = sg.momentum;
    for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n; i += (size_t)nbx * 256) {
        float vv = v[i];
        if (mode != 2) {
            const float gg = g[i] * gscale;
            vv = tn_vel(m, vv, gg);
            v[i] = vv;
        }
        if (mode != 3) {
            const float pv = p[i];
            p[i] = tn_stepped(pv, step, vv);
        }
    }
}

// Update of the PIPELINED single-GPU schedule (two steps in flight on two streams, each with its own
// weights, activations and gradients; NeuralNet / _PipeTrainFn).  Because the reference applies the old
// velocity, p_t = p_{t-1} - s*v_{t-1} with v_{t-1} = m v_{t-2} + (1-m) g_{t-2}: the weights of step t
// need the gradient of step t-2 -- which the same stream produced two steps ago -- and the weights
// p_{t-1} the OTHER stream is using right now (read-only there).  One launch at the start of step t:
//   v = m v + (1-m) g   (update_v; v is shared by both streams)   ;   p = psrc - rate*lr*v
// Same expressions as sgd_update_multi_block: the weight trajectory is bit-identical.
__global__ __launch_bounds__(256) void sgd_update_pipe_kernel(const tn_pipe_seg* __restrict__ segs, int nseg,
                                                             const float* __restrict__ d_lr,
                                                             uint32_t* d_step, uint32_t step_inc, int update_v) {
    const int bx = blockIdx.x, by = blockIdx.y, nbx = gridDim.x;
    if (d_step && step_inc && bx == 0 && by == 0 && threadIdx.x == 0) *d_step += step_inc;
    const tn_pipe_seg sg = segs[by];
    const float step = sg.rate * d_lr[0];
    float* __restrict__ p = sg.p;
    const float* __restrict__ ps = sg.psrc;
    float* __restrict__ v = sg.v;
    const float* __restrict__ g = sg.g;
    const size_t n = sg.n;
    const float m = sg.momentum;
    if ((n & 3) == 0 && (((uintptr_t)p | (uintptr_t)ps | (uintptr_t)v | (uintptr_t)g) & 15) == 0) {
        // 16-byte accesses: the big tensors (wide6's 16.8 M-element FC weight) are pure HBM streaming
        for (size_t i = ((size_t)bx * 256 + threadIdx.x) * 4; i < n; i += (size_t)nbx * 1024) {
            float4 vv = *reinterpret_cast<const float4*>(v + i);
            if (update_v) {
                const float4 gg = *reinterpret_cast<const float4*>(g + i);
                vv.x = tn_vel(m, vv.x, gg.x); vv.y = tn_vel(m, vv.y, gg.y);
                vv.z = tn_vel(m, vv.z, gg.z); vv.w = tn_vel(m, vv.w, gg.w);
                *reinterpret_cast<float4*>(v + i) = vv;
            }
            const float4 pv = *reinterpret_cast<const float4*>(ps + i);
            *reinterpret_cast<float4*>(p + i) = make_float4(tn_stepped(pv.x, step, vv.x), tn_stepped(pv.y, step, vv.y),
                                                            tn_stepped(pv.z, step, vv.z), tn_stepped(pv.w, step, vv.w));
        }
        return;
    }
    for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n; i += (size_t)nbx * 256) {
        float vv = v[i];
        if (update_v) {
            const float gg = g[i];
            vv = tn_vel(m, vv, gg);
            v[i] = vv;
        }
        const float pv = ps[i];
        p[i] = tn_stepped(pv, step, vv);
    }
}

// The same launch with LAZY gradients: segments whose weight gradient is still a stack of partial
// slabs (the deferred finishing sums of reduce.hip) add the slabs up on the fly -- in exactly the
// order slab_sum_multi_kernel uses, so the result is bit-identical -- write the gradient out and
// apply the update, which saves the reduction launch and one round trip of the gradient through HBM.
#define TN_LAZY_SEGS 32                 // segments whose slab sums an update launch can fold in (others: final gradients)
struct LazyBatch {
    int8_t rec_of_seg[TN_LAZY_SEGS];   // record index of segment s, -1: the gradient is already final
    tn_red_rec r[TN_RED_MAX];
};

// Column sums of squares riding in the update (tn_sgd_update_net_maxnorm): a 2-D tensor with a max-norm bound is
// walked by the update in the tiles of maxnorm_cols_partial4 below -- thread = 4 adjacent columns, 4 row lanes, row lane
// r0 takes rows rb + r0, + 4, ... of its row chunk -- and the squares of the NEW weights are added in that kernel's order:
// partial[chunk][col] holds the same bits, and the matrix is not read a second time (wide6: 67 MB per step).
struct ColNormRec {
    int32_t seg, rows, cols, rchunk, R;    // seg < 0: unused
    float* partial;
};
#define TN_COLNORM_MAX 2
struct ColNormBatch {
    ColNormRec c[TN_COLNORM_MAX];
};
// op.load(i4): the operands of elements i4 .. i4+3 (pure loads, the pending slab stack summed); op.apply(i4, regs): the
// update with all its stores, returns the new weights.  CN_U rows' loads are issued before the first store -- the walk
// has 4 waves per CU (the tile geometry is the norm's), so the bytes in flight have to come from each thread.
#define CN_U 8
struct CnRegs {
    float4 s, vv, pv;
};
template <class OP>
__device__ __forceinline__ void colnorm_walk(const ColNormRec& cn, int bx, int nbx, float4 (*red4)[64], const OP& op) {
    const int cl = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    const int ctx4 = (cn.cols + 255) >> 8;
    for (int t = bx; t < ctx4 * cn.R; t += nbx) {
        const int ry = t / ctx4, c = ((t - ry * ctx4) * 64 + cl) * 4;
        const int rb = ry * cn.rchunk, re = min(cn.rows, rb + cn.rchunk);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < cn.cols) {
            int r = rb + r0;
            for (; r + 4 * (CN_U - 1) < re; r += 4 * CN_U) {        // whole batches: no predicate between the loads
                CnRegs rg[CN_U];
#pragma unroll
                for (int u = 0; u < CN_U; ++u) rg[u] = op.load((size_t)(r + 4 * u) * cn.cols + c);
#pragma unroll
                for (int u = 0; u < CN_U; ++u) {
                    const float4 w = op.apply((size_t)(r + 4 * u) * cn.cols + c, rg[u]);
                    s.x = fmaf(w.x, w.x, s.x); s.y = fmaf(w.y, w.y, s.y); s.z = fmaf(w.z, w.z, s.z); s.w = fmaf(w.w, w.w, s.w);
                }
            }
#pragma unroll 1
            for (; r < re; r += 4) {
                const size_t i4 = (size_t)r * cn.cols + c;
                const float4 w = op.apply(i4, op.load(i4));
                s.x = fmaf(w.x, w.x, s.x); s.y = fmaf(w.y, w.y, s.y); s.z = fmaf(w.z, w.z, s.z); s.w = fmaf(w.w, w.w, s.w);
            }
        }
        red4[r0][cl] = s;
        __syncthreads();
        if (r0 == 0 && c < cn.cols) {
            const float4 a = red4[0][cl], b = red4[1][cl], d = red4[2][cl], e = red4[3][cl];
            *reinterpret_cast<float4*>(cn.partial + (size_t)ry * cn.cols + c) =
                make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z),
                            (a.w + b.w) + (d.w + e.w));
        }
        __syncthreads();
    }
}
__device__ __forceinline__ float4 cn_slab_sum(const float* __restrict__ src, uint32_t S, uint32_t stride, size_t i4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (uint32_t z = 0; z < S; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(src + (size_t)z * stride + i4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    return s;
}
// TN_UPD_LAZY form (sgd_update_lazy_kernel) and TN_UPD_PIPE form (sgd_update_pipe_lazy_kernel) of one element quad
template <bool NOSLAB>       // NOSLAB: no tensor of the launch has a pending slab stack (src == NULL everywhere)
struct CnLazyOp {
    float* p; float* v; float* g; const float* src;
    uint32_t S, stride;
    float gscale, m, step, L1, L2;
    __device__ __forceinline__ CnRegs load(size_t i4) const {
        CnRegs r;
        r.s = (!NOSLAB && src) ? cn_slab_sum(src, S, stride, i4) : *reinterpret_cast<const float4*>(g + i4);
        r.pv = *reinterpret_cast<const float4*>(p + i4);
        r.vv = *reinterpret_cast<const float4*>(v + i4);
        return r;
    }
    __device__ __forceinline__ float4 apply(size_t i4, const CnRegs& r) const;
};
template <bool FAST>         // FAST: update_v != 0 and no pending slab stack in the launch
struct CnPipeOp {
    float* p; const float* ps; float* v; float* g; const float* src;
    uint32_t S, stride;
    int update_v;
    float m, step;
    __device__ __forceinline__ CnRegs load(size_t i4) const {
        CnRegs r;
        r.s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) r.s = *reinterpret_cast<const float4*>(g + i4);
        else if (update_v) r.s = src ? cn_slab_sum(src, S, stride, i4) : *reinterpret_cast<const float4*>(g + i4);
        r.pv = *reinterpret_cast<const float4*>(ps + i4);
        r.vv = *reinterpret_cast<const float4*>(v + i4);
        return r;
    }
    __device__ __forceinline__ float4 apply(size_t i4, const CnRegs& r) const {
        float4 vv = r.vv;
        if (FAST || update_v) {
            if (!FAST && src) *reinterpret_cast<float4*>(g + i4) = r.s;
            vv.x = tn_vel(m, vv.x, r.s.x); vv.y = tn_vel(m, vv.y, r.s.y);
            vv.z = tn_vel(m, vv.z, r.s.z); vv.w = tn_vel(m, vv.w, r.s.w);
            *reinterpret_cast<float4*>(v + i4) = vv;
        }
        const float4 pn = make_float4(tn_stepped(r.pv.x, step, vv.x), tn_stepped(r.pv.y, step, vv.y),
                                      tn_stepped(r.pv.z, step, vv.z), tn_stepped(r.pv.w, step, vv.w));
        *reinterpret_cast<float4*>(p + i4) = pn;
        return pn;
    }
};

__device__ __forceinline__ void sgd_apply(float& pv, float& vv, float gg, float gscale, float m, float step,
                                          float L1, float L2) {
    gg *= gscale;
    if (L1 != 0.f) gg += L1 * ((pv > 0.f) - (pv < 0.f));
    if (L2 != 0.f) gg += 2.f * L2 * pv;
    const float vo = vv;
    vv = tn_vel(m, vo, gg);
    pv = tn_stepped(pv, step, vo);
}

template <bool NOSLAB>
__device__ __forceinline__ float4 CnLazyOp<NOSLAB>::apply(size_t i4, const CnRegs& r) const {
    float4 pv = r.pv, vv = r.vv;
    sgd_apply(pv.x, vv.x, r.s.x, gscale, m, step, L1, L2);
    sgd_apply(pv.y, vv.y, r.s.y, gscale, m, step, L1, L2);
    sgd_apply(pv.z, vv.z, r.s.z, gscale, m, step, L1, L2);
    sgd_apply(pv.w, vv.w, r.s.w, gscale, m, step, L1, L2);
    if (!NOSLAB && src) *reinterpret_cast<float4*>(g + i4) = r.s;
    *reinterpret_cast<float4*>(v + i4) = vv;
    *reinterpret_cast<float4*>(p + i4) = pv;
    return pv;
}

__global__ __launch_bounds__(256) void sgd_update_lazy_kernel(const tn_sgd_seg* __restrict__ segs, int nseg,
                                                             const float* __restrict__ d_lr, float gscale,
                                                             uint32_t* d_step_inc,
                                                             const float* __restrict__ rowloss, int nrow,
                                                             float cost_scale, float* __restrict__ d_cost,
                                                             LazyBatch lb, ColNormBatch cb) {
    __shared__ float red[16][17];
    const int bx = blockIdx.x, by = blockIdx.y, nbx = gridDim.x;
    const int ri = (by < nseg && by < TN_LAZY_SEGS) ? lb.rec_of_seg[by] : -1;
#pragma unroll
    for (int k = 0; k < TN_COLNORM_MAX; ++k)
        if (cb.c[k].seg == by) {        // this tensor is walked by sgd_colnorm_lazy_kernel
            if (d_step_inc && bx == 0 && by == 0 && threadIdx.x == 0) *d_step_inc += 1;
            return;
        }
    if (ri < 0) {
        sgd_update_multi_block(segs, nseg, d_lr, gscale, d_step_inc, rowloss, nrow, cost_scale, d_cost, bx, by,
                               nbx, &red[0][0]);
        return;
    }
    if (d_step_inc && bx == 0 && by == 0 && threadIdx.x == 0) *d_step_inc += 1;
    const tn_red_rec rec = lb.r[ri];
    const tn_sgd_seg sg = segs[by];
    const float step = sg.rate * d_lr[0], m = sg.momentum, L1 = sg.L1, L2 = sg.L2;
    float* __restrict__ p = sg.p;
    float* __restrict__ v = sg.v;
    float* __restrict__ g = const_cast<float*>(sg.g);
    const float* __restrict__ src = rec.src;
    const uint32_t n = rec.n, S = rec.S, stride = rec.stride;
    if (S <= 32) {
        const bool vec = rec.flip == 0 && (n & 3) == 0 && (stride & 3) == 0 &&
                         (((uintptr_t)src | (uintptr_t)g | (uintptr_t)p | (uintptr_t)v) & 15) == 0;
        if (vec) {
            for (uint32_t i4 = (bx * 256u + threadIdx.x) * 4u; i4 < n; i4 += nbx * 1024u) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
                for (uint32_t z = 0; z < S; ++z) {
                    const float4 t = *reinterpret_cast<const float4*>(src + (size_t)z * stride + i4);
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                float4 pv = *reinterpret_cast<const float4*>(p + i4), vv = *reinterpret_cast<const float4*>(v + i4);
                sgd_apply(pv.x, vv.x, s.x, gscale, m, step, L1, L2);
                sgd_apply(pv.y, vv.y, s.y, gscale, m, step, L1, L2);
                sgd_apply(pv.z, vv.z, s.z, gscale, m, step, L1, L2);
                sgd_apply(pv.w, vv.w, s.w, gscale, m, step, L1, L2);
                *reinterpret_cast<float4*>(g + i4) = s;
                *reinterpret_cast<float4*>(v + i4) = vv;
                *reinterpret_cast<float4*>(p + i4) = pv;
            }
        } else {
            for (uint32_t i = bx * 256u + threadIdx.x; i < n; i += nbx * 256u) {
                uint32_t j = i;
                if (rec.flip) {
                    const uint32_t kc = i / rec.flip, uv = i - kc * rec.flip;
                    j = kc * rec.flip + (rec.flip - 1 - uv);
                }
                float s = 0.f;
#pragma unroll 8
                for (uint32_t z = 0; z < S; ++z) s += src[(size_t)z * stride + j];
                float pv = p[i], vv = v[i];
                sgd_apply(pv, vv, s, gscale, m, step, L1, L2);
                g[i] = s; v[i] = vv; p[i] = pv;
            }
        }
        return;
    }
    // tall: 16 outputs x 16 slab lanes per trip, lane sums added in lane order (as in reduce.hip)
    const uint32_t ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (uint32_t o0 = bx * 16u; o0 < n; o0 += nbx * 16u) {
        const uint32_t i = o0 + ol;
        float s = 0.f;
        if (i < n) {
            uint32_t j = i;
            if (rec.flip) {
                const uint32_t kc = i / rec.flip, uv = i - kc * rec.flip;
                j = kc * rec.flip + (rec.flip - 1 - uv);
            }
#pragma unroll 4
            for (uint32_t z = sl; z < S; z += 16) s += src[(size_t)z * stride + j];
        }
        red[sl][ol] = s;
        __syncthreads();
        if (sl == 0 && i < n) {
            float t = red[0][ol];
#pragma unroll
            for (int l = 1; l < 16; ++l) t += red[l][ol];
            float pv = p[i], vv = v[i];
            sgd_apply(pv, vv, t, gscale, m, step, L1, L2);
            g[i] = t; v[i] = vv; p[i] = pv;
        }
        __syncthreads();
    }
}

// The pipelined update with LAZY gradients and the cost rider: the stream's previous step left its
// weight-gradient slabs pending (and its per-row losses unsummed); the launch that opens the stream's
// next step adds the slabs up on the fly (order of slab_sum_multi_kernel: bit-identical), stores the
// gradient, applies v = m v + (1-m) g ; p = psrc - rate*lr*v, and one extra block row (by == nseg) sums
// the previous step's cost in the fixed order of sgd_update_multi_block's rider.
__global__ __launch_bounds__(256) void sgd_update_pipe_lazy_kernel(const tn_pipe_seg* __restrict__ segs, int nseg,
                                                                  const float* __restrict__ d_lr, uint32_t* d_step,
                                                                  uint32_t step_inc, int update_v,
                                                                  const float* __restrict__ rowloss, int nrow,
                                                                  float cost_scale, float* __restrict__ d_cost,
                                                                  LazyBatch lb, ColNormBatch cb) {
    __shared__ float red[16][17];
    const int bx = blockIdx.x, by = blockIdx.y, nbx = gridDim.x;
    if (d_step && step_inc && bx == 0 && by == 0 && threadIdx.x == 0) *d_step += step_inc;
    if (by == nseg) {
        if (bx != 0) return;
        float s = 0.f;
        for (int i = threadIdx.x; i < nrow; i += 256) s += rowloss[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        float* r4 = &red[0][0];
        if ((threadIdx.x & 63) == 0) r4[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) d_cost[0] = cost_scale * ((r4[0] + r4[1]) + (r4[2] + r4[3]));
        return;
    }
    const tn_pipe_seg sg = segs[by];
    const float step = sg.rate * d_lr[0], m = sg.momentum;
    float* __restrict__ p = sg.p;
    const float* __restrict__ ps = sg.psrc;
    float* __restrict__ v = sg.v;
    float* __restrict__ g = const_cast<float*>(sg.g);
    const int ri = by < TN_LAZY_SEGS ? lb.rec_of_seg[by] : -1;
#pragma unroll
    for (int k = 0; k < TN_COLNORM_MAX; ++k)
        if (cb.c[k].seg == by) return;  // this tensor is walked by sgd_colnorm_pipe_kernel
    if (ri < 0 || !update_v) {
        const size_t n = sg.n;
        if ((n & 3) == 0 && (((uintptr_t)p | (uintptr_t)ps | (uintptr_t)v | (uintptr_t)g) & 15) == 0) {
            for (size_t i = ((size_t)bx * 256 + threadIdx.x) * 4; i < n; i += (size_t)nbx * 1024) {
                float4 vv = *reinterpret_cast<const float4*>(v + i);
                if (update_v) {
                    const float4 gg = *reinterpret_cast<const float4*>(g + i);
                    vv.x = tn_vel(m, vv.x, gg.x); vv.y = tn_vel(m, vv.y, gg.y);
                    vv.z = tn_vel(m, vv.z, gg.z); vv.w = tn_vel(m, vv.w, gg.w);
                    *reinterpret_cast<float4*>(v + i) = vv;
                }
                const float4 pv = *reinterpret_cast<const float4*>(ps + i);
                *reinterpret_cast<float4*>(p + i) = make_float4(tn_stepped(pv.x, step, vv.x), tn_stepped(pv.y, step, vv.y),
                                                                tn_stepped(pv.z, step, vv.z), tn_stepped(pv.w, step, vv.w));
            }
            return;
        }
        for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n; i += (size_t)nbx * 256) {
            float vv = v[i];
            if (update_v) {
                vv = tn_vel(m, vv, g[i]);
                v[i] = vv;
            }
            p[i] = tn_stepped(ps[i], step, vv);
        }
        return;
    }
    const tn_red_rec rec = lb.r[ri];
    const float* __restrict__ src = rec.src;
    const uint32_t n = rec.n, S = rec.S, stride = rec.stride;
    if (S <= 32) {
        const bool vec = rec.flip == 0 && (n & 3) == 0 && (stride & 3) == 0 &&
                         (((uintptr_t)src | (uintptr_t)g | (uintptr_t)p | (uintptr_t)ps | (uintptr_t)v) & 15) == 0;
        if (vec) {
            for (uint32_t i4 = (bx * 256u + threadIdx.x) * 4u; i4 < n; i4 += nbx * 1024u) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
                for (uint32_t z = 0; z < S; ++z) {
                    const float4 t = *reinterpret_cast<const float4*>(src + (size_t)z * stride + i4);
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                float4 vv = *reinterpret_cast<const float4*>(v + i4);
                const float4 pv = *reinterpret_cast<const float4*>(ps + i4);
                vv.x = tn_vel(m, vv.x, s.x); vv.y = tn_vel(m, vv.y, s.y);
                vv.z = tn_vel(m, vv.z, s.z); vv.w = tn_vel(m, vv.w, s.w);
                *reinterpret_cast<float4*>(g + i4) = s;
                *reinterpret_cast<float4*>(v + i4) = vv;
                *reinterpret_cast<float4*>(p + i4) = make_float4(tn_stepped(pv.x, step, vv.x), tn_stepped(pv.y, step, vv.y),
                                                                 tn_stepped(pv.z, step, vv.z), tn_stepped(pv.w, step, vv.w));
            }
        } else {
            for (uint32_t i = bx * 256u + threadIdx.x; i < n; i += nbx * 256u) {
                uint32_t j = i;
                if (rec.flip) {
                    const uint32_t kc = i / rec.flip, uv = i - kc * rec.flip;
                    j = kc * rec.flip + (rec.flip - 1 - uv);
                }
                float s = 0.f;
#pragma unroll 8
                for (uint32_t z = 0; z < S; ++z) s += src[(size_t)z * stride + j];
                const float vv = tn_vel(m, v[i], s);
                g[i] = s; v[i] = vv; p[i] = tn_stepped(ps[i], step, vv);
            }
        }
        return;
    }
    const uint32_t ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (uint32_t o0 = bx * 16u; o0 < n; o0 += nbx * 16u) {
        const uint32_t i = o0 + ol;
        float s = 0.f;
        if (i < n) {
            uint32_t j = i;
            if (rec.flip) {
                const uint32_t kc = i / rec.flip, uv = i - kc * rec.flip;
                j = kc * rec.flip + (rec.flip - 1 - uv);
            }
#pragma unroll 4
            for (uint32_t z = sl; z < S; z += 16) s += src[(size_t)z * stride + j];
        }
        red[sl][ol] = s;
        __syncthreads();
        if (sl == 0 && i < n) {
            float t = red[0][ol];
#pragma unroll
            for (int l = 1; l < 16; ++l) t += red[l][ol];
            const float vv = tn_vel(m, v[i], t);
            g[i] = t; v[i] = vv; p[i] = tn_stepped(ps[i], step, vv);
        }
        __syncthreads();
    }
}

// The tensors whose column norms ride in the update, in launches of their own (blockIdx.y = slot): the walk keeps 24
// loads per thread in flight (168 registers) -- inside the update kernels it took their flat paths from 8 waves per SIMD to 3.
template <bool NOSLAB>
__global__ __launch_bounds__(256) void sgd_colnorm_lazy_kernel(const tn_sgd_seg* __restrict__ segs, const float* __restrict__ d_lr,
                                                              float gscale, LazyBatch lb, ColNormBatch cb) {
    __shared__ float4 red4[4][64];
    const ColNormRec cn = cb.c[blockIdx.y];
    const int ri = lb.rec_of_seg[cn.seg];
    const tn_sgd_seg sg = segs[cn.seg];
    const CnLazyOp<NOSLAB> op{sg.p, sg.v, const_cast<float*>(sg.g), ri >= 0 ? lb.r[ri].src : nullptr,
                      ri >= 0 ? lb.r[ri].S : 0u, ri >= 0 ? lb.r[ri].stride : 0u,
                      gscale, sg.momentum, sg.rate * d_lr[0], sg.L1, sg.L2};
    colnorm_walk(cn, blockIdx.x, gridDim.x, red4, op);
}
template <bool FAST>
__global__ __launch_bounds__(256) void sgd_colnorm_pipe_kernel(const tn_pipe_seg* __restrict__ segs, const float* __restrict__ d_lr,
                                                              int update_v, LazyBatch lb, ColNormBatch cb) {
    __shared__ float4 red4[4][64];
    const ColNormRec cn = cb.c[blockIdx.y];
    const int ri = lb.rec_of_seg[cn.seg];
    const tn_pipe_seg sg = segs[cn.seg];
    const bool lazy = ri >= 0 && update_v;
    const CnPipeOp<FAST> op{sg.p, sg.psrc, sg.v, const_cast<float*>(sg.g), lazy ? lb.r[ri].src : nullptr, lazy ? lb.r[ri].S : 0u,
                      lazy ? lb.r[ri].stride : 0u, update_v, sg.momentum, sg.rate * d_lr[0]};
    colnorm_walk(cn, blockIdx.x, gridDim.x, red4, op);
}

__global__ __launch_bounds__(256) void clip_kernel(float* __restrict__ p, size_t n, float mx) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = fminf(fmaxf(p[i], -mx), mx);
}

// ndim 4: one block per leading index d0, contiguous 'rest' elements
__global__ __launch_bounds__(256) void maxnorm_rows_kernel(float* __restrict__ p, int rest, float mx) {
    __shared__ float red[4];
    __shared__ float scale_s;
    float* row = p + (size_t)blockIdx.x * rest;
    float s = 0.f;
    for (int i = threadIdx.x; i < rest; i += 256) s += row[i] * row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
        scale_s = (1e-7f + fminf(fmaxf(nrm, 0.f), mx)) / (1e-7f + nrm);
    }
    __syncthreads();
    const float sc = scale_s;
    if (sc == 1.f) return;       // norm within the bound: (1e-7 + n) / (1e-7 + n) is exactly 1, nothing to rewrite
    for (int i = threadIdx.x; i < rest; i += 256) row[i] *= sc;
}

// ndim 2 (rows x cols, row-major): per-COLUMN norm, two passes over row slabs so that a tall matrix
// (wide6's 16384 x 1024 FC weight) fills the chip: partial[slab][col] = sum of squares, then every
// block adds the slabs of its 64 columns in slab order (deterministic) and rescales its rows.
__global__ __launch_bounds__(256) void maxnorm_cols_partial(const float* __restrict__ p, int rows, int cols,
                                                           int rchunk, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, r0 = threadIdx.x >> 6;
    const int rb = blockIdx.y * rchunk, re = min(rows, rb + rchunk);
    float s = 0.f;
    if (c < cols)
        for (int r = rb + r0; r < re; r += 4) {
            const float v = p[(size_t)r * cols + c];
            s = fmaf(v, v, s);
        }
    red[r0][cl] = s;
    __syncthreads();
    if (r0 == 0 && c < cols)
        partial[(size_t)blockIdx.y * cols + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// the same partial sums with 16-byte loads: thread = 4 adjacent columns, 4 row lanes per block (cols % 4 == 0, p
// 16-byte aligned); a column's rows are added in the same order as above (row lane r0 takes rows rb + r0, + 4, ...)
__global__ __launch_bounds__(256) void maxnorm_cols_partial4(const float* __restrict__ p, int rows, int cols,
                                                            int rchunk, float* __restrict__ partial) {
    __shared__ float4 red[4][64];
    const int cl = threadIdx.x & 63, c = (blockIdx.x * 64 + cl) * 4, r0 = threadIdx.x >> 6;
    const int rb = blockIdx.y * rchunk, re = min(rows, rb + rchunk);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
#pragma unroll 8
        for (int r = rb + r0; r < re; r += 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)r * cols + c);
            s.x = fmaf(v.x, v.x, s.x); s.y = fmaf(v.y, v.y, s.y); s.z = fmaf(v.z, v.z, s.z); s.w = fmaf(v.w, v.w, s.w);
        }
    }
    red[r0][cl] = s;
    __syncthreads();
    if (r0 == 0 && c < cols) {
        const float4 a = red[0][cl], b = red[1][cl], d = red[2][cl], e = red[3][cl];
        *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * cols + c) =
            make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z),
                        (a.w + b.w) + (d.w + e.w));
    }
}

__global__ __launch_bounds__(256) void maxnorm_cols_scale(float* __restrict__ p, int rows, int cols, int rchunk,
                                                         const float* __restrict__ partial, int R, float mx) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r0 = threadIdx.x >> 6;
    if (c >= cols) return;
    float s = 0.f;
    // the slab partials are independent loads added in slab order: 16 in flight at a time (a plain loop ran the R
    // L2 round trips back to back -- 16 us for 64 slabs of a 16384 x 1024 matrix, with 0.3 MB moved)
    int z = 0;
    for (; z + 16 <= R; z += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = partial[(size_t)(z + j) * cols + c];
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
    }
    for (; z < R; ++z) s += partial[(size_t)z * cols + c];
    const float nrm = sqrtf(s);
    const float sc = (1e-7f + fminf(fmaxf(nrm, 0.f), mx)) / (1e-7f + nrm);
    // columns within the bound have sc == 1 exactly: their 4 * rows bytes are neither read nor written (a wave
    // whose 64 columns are all within the bound -- the usual case -- leaves without touching the matrix)
    if (sc == 1.f) return;
    const int rb = blockIdx.y * rchunk, re = min(rows, rb + rchunk);
    for (int r = rb + r0; r < re; r += 4) p[(size_t)r * cols + c] *= sc;
}

// every 1-D / 4-D tensor of a net in one launch: blockIdx.y = tensor, blockIdx.x = leading index (4-D: one block
// per kernel, the arithmetic of maxnorm_rows_kernel) or 256-element chunk (1-D: clip_kernel)
struct MnBatch {
    float* p[32];
    int32_t kind[32], d0[32], rest[32];     // kind: 1 or 4
    float mx[32];
};
__global__ __launch_bounds__(256) void maxnorm_multi_kernel(MnBatch b) {
    __shared__ float red[4];
    __shared__ float scale_s;
    const int s = blockIdx.y;
    float* __restrict__ p = b.p[s];
    const float mx = b.mx[s];
    if (b.kind[s] == 1) {
        const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        if (i < (size_t)b.d0[s]) p[i] = fminf(fmaxf(p[i], -mx), mx);
        return;
    }
    if ((int)blockIdx.x >= b.d0[s]) return;
    const int rest = b.rest[s];
    float* row = p + (size_t)blockIdx.x * rest;
    float a = 0.f;
    for (int i = threadIdx.x; i < rest; i += 256) a += row[i] * row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
        scale_s = (1e-7f + fminf(fmaxf(nrm, 0.f), mx)) / (1e-7f + nrm);
    }
    __syncthreads();
    const float sc = scale_s;
    if (sc == 1.f) return;
    for (int i = threadIdx.x; i < rest; i += 256) row[i] *= sc;
}

// row chunks of a (d0 x rest) matrix for the two-pass column norm: enough blocks to fill the chip
static void colnorm_geom(const tn_ctx* ctx, int d0, int rest, int* R_, int* rchunk_) {
    const int ct = cdiv(rest, 64);
    int R = cdiv(4 * ctx->num_cus, ct);
    if (R > cdiv(d0, 16)) R = cdiv(d0, 16);
    if (R < 1) R = 1;
    const int rchunk = cdiv(d0, R);
    *R_ = cdiv(d0, rchunk);
    *rchunk_ = rchunk;
}

// Which of the max-norm tensors get their column sums from the update launch: 2-D, 16-byte rows, the tensor of exactly
// one update segment whose operands (and pending slab stack, if any) take the 16-byte walk.  fused_of_mn[i] = slot or -1.
template <class SEG>
static int colnorm_pick(tn_ctx* ctx, const SEG* h_segs, int nseg, const LazyBatch& lb, const tn_mn_seg* h_mn, int nmn,
                        ColNormBatch* cb, int* fused_of_mn, const float* const* extra_ptr) {
    for (int k = 0; k < TN_COLNORM_MAX; ++k) cb->c[k] = ColNormRec{-1, 0, 0, 0, 0, nullptr};
    for (int i = 0; i < nmn; ++i) fused_of_mn[i] = -1;
    if (!h_segs || !h_mn || nseg > TN_LAZY_SEGS) return TN_OK;
    int nf = 0;
    size_t off = 0, offs[TN_COLNORM_MAX];
    for (int i = 0; i < nmn && nf < TN_COLNORM_MAX; ++i) {
        const tn_mn_seg& mn = h_mn[i];
        if (mn.ndim != 2 || mn.maxnorm == 0.f || !mn.p || mn.rest % 4 != 0 || mn.d0 < 1) continue;
        int seg = -1;
        for (int s = 0; s < nseg; ++s)
            if (h_segs[s].p == mn.p && h_segs[s].n == (uint64_t)mn.d0 * mn.rest) seg = s;
        if (seg < 0) continue;
        uintptr_t al = (uintptr_t)h_segs[seg].p | (uintptr_t)h_segs[seg].v | (uintptr_t)h_segs[seg].g |
                       (uintptr_t)(extra_ptr ? extra_ptr[seg] : nullptr);
        const int ri = lb.rec_of_seg[seg];
        if (ri >= 0) {
            const tn_red_rec& r = lb.r[ri];
            if (r.flip != 0 || r.S > 32 || (r.stride & 3)) continue;
            al |= (uintptr_t)r.src;
        }
        if (al & 15) continue;
        ColNormRec& c = cb->c[nf];
        c.seg = seg; c.rows = mn.d0; c.cols = mn.rest;
        colnorm_geom(ctx, mn.d0, mn.rest, &c.R, &c.rchunk);
        offs[nf] = off;
        off += (((size_t)c.R * c.cols * sizeof(float)) + 255) & ~(size_t)255;
        fused_of_mn[i] = nf++;
    }
    if (nf) {
        float* buf;
        int rc = tn_tmp_get(ctx, off, &buf);
        if (rc) return rc;
        for (int k = 0; k < nf; ++k) cb->c[k].partial = reinterpret_cast<float*>(reinterpret_cast<char*>(buf) + offs[k]);
    }
    return TN_OK;
}
static int maxnorm_multi_impl(tn_ctx* ctx, const tn_mn_seg* h_segs, int nseg, const ColNormBatch* cb, const int* fused_of_mn);
static bool colnorm_slabs(const ColNormBatch& cb, const LazyBatch& lb) {
    for (int k = 0; k < TN_COLNORM_MAX; ++k)
        if (cb.c[k].seg >= 0 && lb.rec_of_seg[cb.c[k].seg] >= 0) return true;
    return false;
}
static dim3 colnorm_grid(const ColNormBatch& cb) {
    int gx = 1, gy = 0;
    for (int k = 0; k < TN_COLNORM_MAX; ++k)
        if (cb.c[k].seg >= 0) {
            gx = std::max(gx, cdiv(cb.c[k].cols, 256) * cb.c[k].R);
            gy = k + 1;
        }
    return dim3(gx, gy);
}

extern "C" {

int tn_sgd_update(tn_ctx* ctx, float* p, float* v, const float* g, size_t n, float momentum,
                  float rate, const float* d_lr, float L1, float L2, float gscale) {
    if (!n) return TN_OK;
    TN_REQUIRE(d_lr != nullptr, "tn_sgd_update: d_lr is NULL");
    int blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    sgd_update_kernel<<<blocks, 256, 0, ctx->stream>>>(p, v, g, n, momentum, rate, d_lr, L1, L2,
                                                      gscale);
    TN_LAUNCH_CHECK();
    return TN_OK;
}


static int upd_cost(tn_ctx* ctx, const tn_sgd_seg* d_segs, int nseg, size_t max_n,
                             const float* d_lr, float gscale, uint32_t* d_step_inc,
                             const float* rowloss, int nrow, float cost_scale, float* d_cost) {
    const bool rider = rowloss != nullptr;
    if (nseg <= 0 && !rider) return TN_OK;
    TN_REQUIRE(nseg <= 0 || (d_segs != nullptr && d_lr != nullptr), "tn_sgd_update_net: NULL argument");
    TN_REQUIRE(!rider || (d_cost != nullptr && nrow > 0), "tn_sgd_update_net (cost): bad cost arguments");
    if (nseg < 0) nseg = 0;
    int bx = cdiv(max_n, 1024);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    sgd_update_multi_kernel<<<dim3(bx, nseg + (rider ? 1 : 0)), 256, 0, ctx->stream>>>(
        d_segs, nseg, d_lr, gscale, d_step_inc, rowloss, nrow, cost_scale, d_cost);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

static int upd_delayed(tn_ctx* ctx, const tn_sgd_seg* d_segs, int nseg, size_t max_n,
                                const float* d_lr, float gscale, uint32_t* d_step_inc, int mode) {
    TN_REQUIRE(nseg > 0 && d_segs && d_lr && mode >= 1 && mode <= 3, "tn_sgd_update_net (delayed): bad arguments");
    int bx = cdiv(max_n, 1024);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    sgd_update_delayed_kernel<<<dim3(bx, nseg), 256, 0, ctx->stream>>>(d_segs, nseg, d_lr, gscale, d_step_inc, mode);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

static int upd_pipe(tn_ctx* ctx, const tn_pipe_seg* d_segs, const tn_pipe_seg* h_segs, int nseg, size_t max_n,
                             const float* d_lr, uint32_t* d_step, uint32_t step_inc, int update_v,
                             const float* rowloss, int nrow, float cost_scale, float* d_cost,
                             const tn_mn_seg* h_mn = nullptr, int nmn = 0) {
    TN_REQUIRE(nseg > 0 && d_segs && d_lr, "tn_sgd_update_net (pipe): bad arguments");
    const bool rider = rowloss != nullptr;
    TN_REQUIRE(!rider || (d_cost != nullptr && nrow > 0), "tn_sgd_update_net (pipe): bad cost arguments");
    int bx = cdiv(max_n, 1024);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    // this stream's pending slab sums whose output is the gradient of one of the segments are folded into
    // the update; the others are finished by the ordinary reduction launch first
    LazyBatch lb{};
    for (int s = 0; s < TN_LAZY_SEGS; ++s) lb.rec_of_seg[s] = -1;
    int nlazy = 0, keep = 0;
    const bool can = h_segs != nullptr && nseg <= TN_LAZY_SEGS && update_v;
    for (int i = 0; i < ctx->npend; ++i) {
        int seg = -1;
        if (can)
            for (int s = 0; s < nseg; ++s)
                if (h_segs[s].g == ctx->pend[i].out && h_segs[s].n == ctx->pend[i].n && lb.rec_of_seg[s] < 0) seg = s;
        if (seg >= 0) {
            lb.r[nlazy] = ctx->pend[i];
            lb.rec_of_seg[seg] = (int8_t)nlazy++;
        } else {
            ctx->pend[keep++] = ctx->pend[i];
        }
    }
    ctx->npend = keep;
    ctx->defer = false;
    int rc = tn_red_flush(ctx);             // leftovers (also resets the scratch bump pointer)
    if (rc) return rc;
    ctx->scratch_off = 0;
    ColNormBatch cb;
    int fused_of_mn[32];
    {
        const float* ps[TN_LAZY_SEGS];
        if (h_segs && nseg <= TN_LAZY_SEGS)
            for (int s = 0; s < nseg; ++s) ps[s] = h_segs[s].psrc;
        rc = colnorm_pick(ctx, h_segs, nseg, lb, h_mn, nmn, &cb, fused_of_mn, ps);
        if (rc) return rc;
    }
    if (nlazy == 0 && !rider && cb.c[0].seg < 0) {
        sgd_update_pipe_kernel<<<dim3(bx, nseg), 256, 0, ctx->stream>>>(d_segs, nseg, d_lr, d_step, step_inc, update_v);
    } else {
        sgd_update_pipe_lazy_kernel<<<dim3(bx, nseg + (rider ? 1 : 0)), 256, 0, ctx->stream>>>(
            d_segs, nseg, d_lr, d_step, step_inc, update_v, rowloss, nrow, cost_scale, d_cost, lb, cb);
    }
    TN_LAUNCH_CHECK();
    if (cb.c[0].seg >= 0) {
        if (update_v && !colnorm_slabs(cb, lb))
            sgd_colnorm_pipe_kernel<true><<<colnorm_grid(cb), 256, 0, ctx->stream>>>(d_segs, d_lr, update_v, lb, cb);
        else
            sgd_colnorm_pipe_kernel<false><<<colnorm_grid(cb), 256, 0, ctx->stream>>>(d_segs, d_lr, update_v, lb, cb);
        TN_LAUNCH_CHECK();
    }
    return nmn ? maxnorm_multi_impl(ctx, h_mn, nmn, &cb, fused_of_mn) : TN_OK;
}

static int upd_lazy(tn_ctx* ctx, const tn_sgd_seg* d_segs, const tn_sgd_seg* h_segs, int nseg,
                             size_t max_n, const float* d_lr, float gscale, uint32_t* d_step_inc,
                             const float* rowloss, int nrow, float cost_scale, float* d_cost,
                             const tn_mn_seg* h_mn = nullptr, int nmn = 0) {
    TN_REQUIRE(nseg > 0 && nseg <= TN_LAZY_SEGS && d_segs && h_segs && d_lr, "tn_sgd_update_net (lazy): bad arguments");
    const bool rider = rowloss != nullptr;
    TN_REQUIRE(!rider || (d_cost != nullptr && nrow > 0), "tn_sgd_update_net (lazy): bad cost arguments");
    // pending slab sums whose output is the gradient of one of the segments are folded into the update;
    // the others (and everything when deferral is off) are finished by the ordinary reduction launch
    LazyBatch lb{};
    for (int s = 0; s < TN_LAZY_SEGS; ++s) lb.rec_of_seg[s] = -1;
    int nlazy = 0, keep = 0;
    for (int i = 0; i < ctx->npend; ++i) {
        int seg = -1;
        for (int s = 0; s < nseg; ++s)
            if (h_segs[s].g == ctx->pend[i].out && h_segs[s].n == ctx->pend[i].n && lb.rec_of_seg[s] < 0) seg = s;
        if (seg >= 0) {
            lb.r[nlazy] = ctx->pend[i];
            lb.rec_of_seg[seg] = (int8_t)nlazy++;
        } else {
            ctx->pend[keep++] = ctx->pend[i];
        }
    }
    ctx->npend = keep;
    ctx->defer = false;
    int rc = tn_red_flush(ctx);             // leftovers (also resets the scratch bump pointer)
    if (rc) return rc;
    ctx->scratch_off = 0;
    ColNormBatch cb;
    int fused_of_mn[32];
    rc = colnorm_pick(ctx, h_segs, nseg, lb, h_mn, nmn, &cb, fused_of_mn, nullptr);
    if (rc) return rc;
    if (nlazy == 0 && cb.c[0].seg < 0) {
        rc = upd_cost(ctx, d_segs, nseg, max_n, d_lr, gscale, d_step_inc, rowloss, nrow, cost_scale, d_cost);
        if (rc) return rc;
        return nmn ? maxnorm_multi_impl(ctx, h_mn, nmn, nullptr, nullptr) : TN_OK;
    }
    int bx = cdiv(max_n, 1024);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    sgd_update_lazy_kernel<<<dim3(bx, nseg + (rider ? 1 : 0)), 256, 0, ctx->stream>>>(
        d_segs, nseg, d_lr, gscale, d_step_inc, rowloss, nrow, cost_scale, d_cost, lb, cb);
    TN_LAUNCH_CHECK();
    if (cb.c[0].seg >= 0) {
        if (!colnorm_slabs(cb, lb))
            sgd_colnorm_lazy_kernel<true><<<colnorm_grid(cb), 256, 0, ctx->stream>>>(d_segs, d_lr, gscale, lb, cb);
        else
            sgd_colnorm_lazy_kernel<false><<<colnorm_grid(cb), 256, 0, ctx->stream>>>(d_segs, d_lr, gscale, lb, cb);
        TN_LAUNCH_CHECK();
    }
    return nmn ? maxnorm_multi_impl(ctx, h_mn, nmn, &cb, fused_of_mn) : TN_OK;
}

// The one update entry point of a step (include/theanet_hip.h): mode selects the schedule's form
int tn_sgd_update_net_maxnorm(tn_ctx* ctx, int mode, const void* d_segs, const void* h_segs, int nseg, size_t max_n,
                              const float* d_lr, float gscale, uint32_t* d_step, uint32_t step_inc, int flags,
                              const float* rowloss, int nrow, float cost_scale, float* d_cost, const tn_mn_seg* h_mn,
                              int nmn) {
    TN_REQUIRE(nmn >= 0 && nmn <= 32 && (nmn == 0 || h_mn), "tn_sgd_update_net_maxnorm: bad max-norm list");
    TN_REQUIRE(mode == TN_UPD_PIPE || d_step == nullptr || step_inc == 1,
               "tn_sgd_update_net: the step counter advances by one outside the pipelined schedule");
    if (mode == TN_UPD_LAZY)
        return upd_lazy(ctx, static_cast<const tn_sgd_seg*>(d_segs), static_cast<const tn_sgd_seg*>(h_segs), nseg, max_n,
                        d_lr, gscale, d_step, rowloss, nrow, cost_scale, d_cost, h_mn, nmn);
    if (mode == TN_UPD_PIPE)
        return upd_pipe(ctx, static_cast<const tn_pipe_seg*>(d_segs), static_cast<const tn_pipe_seg*>(h_segs), nseg, max_n,
                        d_lr, d_step, step_inc, flags & 1, rowloss, nrow, cost_scale, d_cost, h_mn, nmn);
    int rc = tn_sgd_update_net(ctx, mode, d_segs, h_segs, nseg, max_n, d_lr, gscale, d_step, step_inc, flags, rowloss, nrow,
                               cost_scale, d_cost);
    if (rc) return rc;
    return nmn ? maxnorm_multi_impl(ctx, h_mn, nmn, nullptr, nullptr) : TN_OK;
}

int tn_sgd_update_net(tn_ctx* ctx, int mode, const void* d_segs, const void* h_segs, int nseg, size_t max_n,
                      const float* d_lr, float gscale, uint32_t* d_step, uint32_t step_inc, int flags,
                      const float* rowloss, int nrow, float cost_scale, float* d_cost) {
    TN_REQUIRE(mode == TN_UPD_PIPE || d_step == nullptr || step_inc == 1,
               "tn_sgd_update_net: the step counter advances by one outside the pipelined schedule");
    switch (mode) {
        case TN_UPD_PLAIN:
            return upd_cost(ctx, static_cast<const tn_sgd_seg*>(d_segs), nseg, max_n, d_lr, gscale, d_step, rowloss, nrow,
                            cost_scale, d_cost);
        case TN_UPD_LAZY:
            return upd_lazy(ctx, static_cast<const tn_sgd_seg*>(d_segs), static_cast<const tn_sgd_seg*>(h_segs), nseg, max_n,
                            d_lr, gscale, d_step, rowloss, nrow, cost_scale, d_cost);
        case TN_UPD_DELAYED:
            TN_REQUIRE(rowloss == nullptr, "tn_sgd_update_net (delayed): no cost rider in this mode");
            return upd_delayed(ctx, static_cast<const tn_sgd_seg*>(d_segs), nseg, max_n, d_lr, gscale, d_step, flags);
        case TN_UPD_PIPE:
            return upd_pipe(ctx, static_cast<const tn_pipe_seg*>(d_segs), static_cast<const tn_pipe_seg*>(h_segs), nseg, max_n,
                            d_lr, d_step, step_inc, flags & 1, rowloss, nrow, cost_scale, d_cost);
        default:
            return tn_fail(ctx, TN_E_ARG, "tn_sgd_update_net: mode %d", mode);
    }
}

int tn_maxnorm(tn_ctx* ctx, float* p, int ndim, int d0, int rest, float maxnorm) {
    if (maxnorm == 0.f) return TN_OK;
    if (ndim == 1) {
        clip_kernel<<<cdiv(d0, 256), 256, 0, ctx->stream>>>(p, (size_t)d0, maxnorm);
    } else if (ndim == 2) {
        const int ct = cdiv(rest, 64);
        int R, rchunk;
        colnorm_geom(ctx, d0, rest, &R, &rchunk);
        float* partial;
        int rc = tn_scratch_get(ctx, (size_t)R * rest * sizeof(float), &partial);
        if (rc) return rc;
        if (rest % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0)
            maxnorm_cols_partial4<<<dim3(cdiv(rest, 256), R), 256, 0, ctx->stream>>>(p, d0, rest, rchunk, partial);
        else
            maxnorm_cols_partial<<<dim3(ct, R), 256, 0, ctx->stream>>>(p, d0, rest, rchunk, partial);
        TN_LAUNCH_CHECK();
        maxnorm_cols_scale<<<dim3(ct, R), 256, 0, ctx->stream>>>(p, d0, rest, rchunk, partial, R, maxnorm);
    } else if (ndim == 4) {
        maxnorm_rows_kernel<<<d0, 256, 0, ctx->stream>>>(p, rest, maxnorm);
    } else {
        return tn_fail(ctx, TN_E_ARG, "tn_maxnorm: ndim %d unsupported (1, 2 or 4)", ndim);
    }
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_maxnorm_multi(tn_ctx* ctx, const tn_mn_seg* h_segs, int nseg) {
    return maxnorm_multi_impl(ctx, h_segs, nseg, nullptr, nullptr);
}

}  // extern "C"

// fused_of_mn[i] >= 0: the column sums of tensor i are already in cb->c[slot].partial (left by the update launch)
static int maxnorm_multi_impl(tn_ctx* ctx, const tn_mn_seg* h_segs, int nseg, const ColNormBatch* cb, const int* fused_of_mn) {
    TN_REQUIRE(nseg >= 0 && nseg <= 32 && (nseg == 0 || h_segs), "tn_maxnorm_multi: bad arguments");
    MnBatch b{};
    int nb = 0, gx = 1;
    for (int i = 0; i < nseg; ++i) {
        const tn_mn_seg& sg = h_segs[i];
        if (sg.maxnorm == 0.f || !sg.p) continue;
        if (sg.ndim == 2 && cb && fused_of_mn[i] >= 0) {
            const ColNormRec& c = cb->c[fused_of_mn[i]];
            maxnorm_cols_scale<<<dim3(cdiv(sg.rest, 64), c.R), 256, 0, ctx->stream>>>(sg.p, sg.d0, sg.rest, c.rchunk, c.partial,
                                                                                     c.R, sg.maxnorm);
            TN_LAUNCH_CHECK();
            continue;
        }
        if (sg.ndim == 2) {
            int rc = tn_maxnorm(ctx, sg.p, 2, sg.d0, sg.rest, sg.maxnorm);
            if (rc) return rc;
            continue;
        }
        TN_REQUIRE(sg.ndim == 1 || sg.ndim == 4, "tn_maxnorm_multi: ndim %d unsupported (1, 2 or 4)", sg.ndim);
        b.p[nb] = sg.p; b.kind[nb] = sg.ndim; b.d0[nb] = sg.d0; b.rest[nb] = sg.rest; b.mx[nb] = sg.maxnorm;
        const int blocks = sg.ndim == 1 ? cdiv(sg.d0, 256) : sg.d0;
        if (blocks > gx) gx = blocks;
        ++nb;
    }
    if (nb) {
        maxnorm_multi_kernel<<<dim3(gx, nb), 256, 0, ctx->stream>>>(b);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}
