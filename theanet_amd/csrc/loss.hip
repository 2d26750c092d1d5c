// Fused log-softmax + NLL + argmax + dlogits.  One 64-lane wave per row, 4 rows per block.
// Semantics: theanet/layer/outlayers.py:50-51 (nll), :69-80 (error stats), :87-95 (SoftmaxLayer).
#include "common.h"

template <bool COST>
__global__ __launch_bounds__(256) void softmax_nll_kernel(
    const float* __restrict__ z, const int32_t* __restrict__ y, int64_t y_row0,
    const int64_t* __restrict__ d_row0, float* __restrict__ logprob, float* __restrict__ rowloss,
    int32_t* __restrict__ pred, float* __restrict__ rowp, float* __restrict__ dz, int B, int n_out,
    float inv_batch, float cost_scale, float* __restrict__ cost, float* __restrict__ blk_part,
    unsigned* __restrict__ counter) {
    __shared__ float wl[4];
    __shared__ unsigned ticket_s;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float myloss = 0.f;
    if (row < B) {
    const float* zr = z + (size_t)row * n_out;
    // max + first argmax (numpy argmax semantics: first maximal index)
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < n_out; c += 64) {
        const float v = zr[c];
        if (v > m) {
            m = v;
            am = c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) {
            m = om;
            am = oa;
        }
    }
    float s = 0.f;
    for (int c = lane; c < n_out; c += 64) s += expf(zr[c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float lse = logf(s);
    const int64_t yoff = y_row0 + (d_row0 ? *d_row0 : 0);
    const int label = y ? y[yoff + row] : -1;
    for (int c = lane; c < n_out; c += 64) {
        const float lp = zr[c] - m - lse;
        if (logprob) logprob[(size_t)row * n_out + c] = lp;
        if (dz) dz[(size_t)row * n_out + c] = (expf(lp) - (c == label ? 1.f : 0.f)) * inv_batch;
        if (c == label) {
            if (rowloss) rowloss[row] = -lp;
            if (rowp) rowp[row] = expf(lp);
            myloss = -lp;
        }
    }
    if (lane == 0 && pred) pred[row] = am;
    }
    if (COST) {
        // block partial -> global; the last block to arrive sums all partials in index order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) myloss += __shfl_xor(myloss, o, 64);
        if (lane == 0) wl[threadIdx.x >> 6] = myloss;
        __syncthreads();
        if (threadIdx.x == 0) {
            // write-through (sc1) store of the partial, drained before the ticket: no L2
            // write-back fence needed (cdna_hip_programming.md G16, form R1)
            __hip_atomic_store(&blk_part[blockIdx.x], wl[0] + wl[1] + wl[2] + wl[3], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ticket_s = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (ticket_s == gridDim.x - 1) {
            float s = 0.f;
            for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)
                s += __hip_atomic_load(&blk_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            __syncthreads();
            if (lane == 0) wl[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) {
                cost[0] = cost_scale * (wl[0] + wl[1] + wl[2] + wl[3]);
                *counter = 0;                    // ready for the next launch
            }
        }
    }
}

extern "C" int tn_softmax_nll(tn_ctx* ctx, const float* z, const int32_t* y, int64_t y_row0,
                              const int64_t* d_row0, float* logprob, float* rowloss, int32_t* pred,
                              float* rowp, float* dz, int B, int n_out, float inv_batch) {
    TN_REQUIRE(B > 0 && n_out > 0, "tn_softmax_nll: bad shape");
    TN_REQUIRE(y != nullptr || (rowloss == nullptr && dz == nullptr && rowp == nullptr),
               "tn_softmax_nll: labels required for loss/gradient outputs");
    softmax_nll_kernel<false><<<cdiv(B, 4), 256, 0, ctx->stream>>>(
        z, y, y_row0, d_row0, logprob, rowloss, pred, rowp, dz, B, n_out, inv_batch, 0.f, nullptr,
        nullptr, nullptr);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

extern "C" size_t tn_softmax_cost_ws_bytes(int B) { return ((size_t)cdiv(B, 4) + 4) * sizeof(float); }

extern "C" int tn_softmax_nll_cost(tn_ctx* ctx, const float* z, const int32_t* y, int64_t y_row0,
                                   const int64_t* d_row0, float* logprob, float* rowloss,
                                   int32_t* pred, float* rowp, float* dz, int B, int n_out,
                                   float inv_batch, float cost_scale, float* cost, void* ws) {
    TN_REQUIRE(B > 0 && n_out > 0 && y && cost && ws, "tn_softmax_nll_cost: bad arguments");
    unsigned* counter = (unsigned*)ws;
    float* part = (float*)ws + 4;
    softmax_nll_kernel<true><<<cdiv(B, 4), 256, 0, ctx->stream>>>(
        z, y, y_row0, d_row0, logprob, rowloss, pred, rowp, dz, B, n_out, inv_batch, cost_scale, cost,
        part, counter);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
