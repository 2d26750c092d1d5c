// Output heads and losses other than Softmax + 'nll' (theanet/layer/outlayers.py:38-64, 105-224) as ONE row
// kernel: a 64-lane wave owns a sample, produces the head's log-probabilities / predictions / error
// statistic, the per-row loss and d cost / d (pre-activation) in a single pass over its row -- the same
// plumbing as loss.hip's softmax_nll_kernel (rows in, rows out; HBM-bound, B x n floats each way).
//
//   head SOFTMAX : probs = softmax(a); losses nll, nllsq (:41-42), nll truncated at a threshold (:44-48),
//                  hinge and exp on the softmax OUTPUT (:50-64 / :38-39 applied to self.output = probs)
//   head EXPLOSS : o = a - mean(a); probs = softmax(o); cost = mean exp(-o[n, y_n])            (:105-126)
//   head HINGE   : output = a;  cost = mean_{n,c} max(0, a[n,c] + 1 - a[n, y_n])               (:129-147)
//   head LOGIT   : a = sigmoid features v; v' = v(1-2e)+e; bitprob = c v' + (1-c)(1-v');
//                  logprob[n,k] = sum_f log bitprob[n,k,f]                                      (:196-203)
//   head RBF     : a = 1.7 tanh features v; dists[n,k] = sum_f (v - c_k)^2; probs = softmax(-[dists, junk]) (:204-210)
// For the centered heads da is already d cost / d z (multiplied by act'(v)); dcenters (RBF, learn_centers)
// accumulates with float atomics into a buffer the caller zeroed.
#include "common.h"

enum { HEAD_SOFTMAX = 0, HEAD_EXPLOSS = 1, HEAD_HINGE = 2, HEAD_LOGIT = 3, HEAD_RBF = 4 };
enum { LOSS_NLL = 0, LOSS_NLLSQ = 1, LOSS_NLLTRUNC = 2, LOSS_HINGE = 3, LOSS_EXP = 4 };

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// first maximal index of vals[0:n) (numpy argmax), wave-cooperative
__device__ __forceinline__ int wargmax(const float* vals, int n, int lane) {
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < n; c += 64) {
        const float v = vals[c];
        if (v > m) { m = v; am = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    return am == 0x7fffffff ? 0 : am;
}

__global__ __launch_bounds__(256) void head_rows_kernel(
    int head, int loss, float loss_param, const float* __restrict__ a, const float* __restrict__ centers, int ncls,
    const int32_t* __restrict__ y, int64_t y_row0, const int64_t* __restrict__ d_row0, float* __restrict__ feat,
    float* __restrict__ logprob, float* __restrict__ rowloss, int32_t* __restrict__ pred, float* __restrict__ rowstat,
    float* __restrict__ da, float* __restrict__ dcenters, int B, int n, float inv_batch, float junk_dist, int act,
    float act_prm) {
    extern __shared__ float head_sm[];                 // centered heads: [4 waves][ncls + 1] class scores
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* ar = a + (size_t)row * n;
    const int label = y ? y[y_row0 + (d_row0 ? *d_row0 : 0) + row] : -1;
    if (head <= HEAD_HINGE) {
        float* lp = logprob + (size_t)row * n;
        float mean = 0.f;
        if (head == HEAD_EXPLOSS) {
            float s = 0.f;
            for (int c = lane; c < n; c += 64) s += ar[c];
            mean = wsum(s) / n;
        }
        const int am = wargmax(ar, n, lane);
        if (lane == 0 && pred) pred[row] = am;
        if (head == HEAD_HINGE) {
            const float zy = label >= 0 ? ar[label] : 0.f;
            float s = 0.f, cnt = 0.f;
            for (int c = lane; c < n; c += 64) {
                lp[c] = ar[c];
                const float t = ar[c] + 1.f - zy;
                if (label >= 0) {
                    s += fmaxf(0.f, t);
                    const float act_c = (c != label && t >= 0.f) ? 1.f : 0.f;
                    cnt += act_c;
                    if (da && c != label) da[(size_t)row * n + c] = act_c * inv_batch / n;
                }
            }
            s = wsum(s); cnt = wsum(cnt);
            if (lane == 0 && label >= 0) {
                if (rowloss) rowloss[row] = s / n;
                if (rowstat) rowstat[row] = zy;
                if (da) da[(size_t)row * n + label] = -cnt * inv_batch / n;
            }
            return;
        }
        // softmax of (a - mean)
        float m = -INFINITY;
        for (int c = lane; c < n; c += 64) m = fmaxf(m, ar[c] - mean);
        m = wmax(m);
        float s = 0.f;
        for (int c = lane; c < n; c += 64) s += expf(ar[c] - mean - m);
        const float lse = logf(wsum(s));
        for (int c = lane; c < n; c += 64) {
            lp[c] = ar[c] - mean - m - lse;
            if (feat) feat[(size_t)row * n + c] = ar[c] - mean;
        }
        if (label < 0) return;
        const float lpy = ar[label] - mean - m - lse, py = expf(lpy);
        if (lane == 0 && rowstat) rowstat[row] = py;
        float rl = 0.f;
        if (head == HEAD_EXPLOSS) {
            const float oy = ar[label] - mean, e = expf(-oy);
            rl = e;
            if (da)
                for (int c = lane; c < n; c += 64) da[(size_t)row * n + c] = -e * inv_batch * ((c == label ? 1.f : 0.f) - 1.f / n);
        } else if (loss == LOSS_NLL || loss == LOSS_NLLTRUNC || loss == LOSS_NLLSQ) {
            float gl;                                   // d cost / d logprob[y] * B
            if (loss == LOSS_NLL) { rl = -lpy; gl = -1.f; }
            else if (loss == LOSS_NLLSQ) { rl = lpy * lpy; gl = 2.f * lpy; }
            else { const float t = loss_param - lpy; rl = fmaxf(0.f, t); gl = t >= 0.f ? -1.f : 0.f; }
            if (da)
                for (int c = lane; c < n; c += 64)
                    da[(size_t)row * n + c] = gl * inv_batch * ((c == label ? 1.f : 0.f) - expf(lp[c]));
        } else {
            // losses on the softmax OUTPUT p: gp = d cost / d p, then dz_c = p_c (gp_c - sum_k gp_k p_k)
            float dot = 0.f, hs = 0.f;
            for (int c = lane; c < n; c += 64) {
                const float pc = expf(lp[c]);
                float gp;
                if (loss == LOSS_HINGE) {
                    hs += fmaxf(0.f, pc + 1.f - py);
                    gp = c == label ? -(float)(n - 1) / n : 1.f / n;    // every margin is active: p in (0,1)
                } else {
                    gp = c == label ? -expf(-py) : 0.f;
                }
                dot += gp * pc;
            }
            dot = wsum(dot); hs = wsum(hs);
            rl = loss == LOSS_HINGE ? hs / n : expf(-py);
            if (da)
                for (int c = lane; c < n; c += 64) {
                    const float pc = expf(lp[c]);
                    const float gp = loss == LOSS_HINGE ? (c == label ? -(float)(n - 1) / n : 1.f / n)
                                                        : (c == label ? -expf(-py) : 0.f);
                    da[(size_t)row * n + c] = inv_batch * pc * (gp - dot);
                }
        }
        if (lane == 0 && rowloss) rowloss[row] = rl;
        return;
    }
    // ---- centered heads: a = features v (B x n), centers (ncls x n) ----
    const int ncol = head == HEAD_RBF ? ncls + 1 : ncls;
    float* lp = logprob + (size_t)row * ncol;
    float* sc = head_sm + (threadIdx.x >> 6) * (ncls + 1);          // per-class scores of this wave's row
    const float eps = 0.001f;
    for (int k = 0; k < ncls; ++k) {
        const float* ck = centers + (size_t)k * n;
        float s = 0.f;
        for (int f = lane; f < n; f += 64) {
            if (head == HEAD_LOGIT) {
                const float v = ar[f] * (1.f - 2.f * eps) + eps;
                s += logf(ck[f] * v + (1.f - ck[f]) * (1.f - v));
            } else {
                const float d = ar[f] - ck[f];
                s += d * d;
            }
        }
        s = wsum(s);
        if (lane == 0) sc[k] = head == HEAD_LOGIT ? s : -s;
    }
    if (head == HEAD_RBF && lane == 0) sc[ncls] = -junk_dist;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the LDS pipe is in order within a wave)
    if (head == HEAD_RBF) {
        float m = -INFINITY;
        for (int c = lane; c < ncol; c += 64) m = fmaxf(m, sc[c]);
        m = wmax(m);
        float s = 0.f;
        for (int c = lane; c < ncol; c += 64) s += expf(sc[c] - m);
        const float lse = logf(wsum(s));
        for (int c = lane; c < ncol; c += 64) sc[c] = sc[c] - m - lse;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    for (int c = lane; c < ncol; c += 64) lp[c] = sc[c];
    const int am = wargmax(sc, ncol, lane);
    if (lane == 0 && pred) pred[row] = am;
    if (label < 0) return;
    const float lpy = sc[label];
    if (lane == 0 && rowloss) rowloss[row] = -lpy;
    const float* cy = centers + (size_t)label * n;
    if (head == HEAD_LOGIT) {
        float wrong = 0.f;
        for (int f = lane; f < n; f += 64) {
            const float v = ar[f] * (1.f - 2.f * eps) + eps;
            const float bp = cy[f] * v + (1.f - cy[f]) * (1.f - v);
            wrong += bp < .5f ? 1.f : 0.f;
            if (da) da[(size_t)row * n + f] = -inv_batch * (1.f - 2.f * eps) * (2.f * cy[f] - 1.f) / bp *
                                              tn_act_grad_from_out(ar[f], act, act_prm);
        }
        wrong = wsum(wrong);
        if (lane == 0 && rowstat) rowstat[row] = wrong / n;
    } else {
        if (lane == 0 && rowstat) rowstat[row] = expf(lpy);
        for (int f = lane; f < n; f += 64) {
            float gv = 0.f;
            for (int k = 0; k < ncls; ++k) {
                const float w = ((k == label ? 1.f : 0.f) - expf(sc[k])) * inv_batch;     // d cost / d dists[k]
                const float d = ar[f] - centers[(size_t)k * n + f];
                gv += w * 2.f * d;
                if (dcenters) atomicAdd(&dcenters[(size_t)k * n + f], -w * 2.f * d);
            }
            if (da) da[(size_t)row * n + f] = gv * tn_act_grad_from_out(ar[f], act, act_prm);
        }
    }
}

extern "C" int tn_head_rows(tn_ctx* ctx, int head, int loss, float loss_param, const float* a, const float* centers,
                            int ncls, const int32_t* y, int64_t y_row0, const int64_t* d_row0, float* feat,
                            float* logprob, float* rowloss, int32_t* pred, float* rowstat, float* da, float* dcenters,
                            int B, int n, float inv_batch, float junk_dist, int act, float act_param) {
    TN_REQUIRE(B > 0 && n > 0 && a && logprob && head >= 0 && head <= 4 && loss >= 0 && loss <= 4,
               "tn_head_rows: bad arguments");
    TN_REQUIRE(head < HEAD_LOGIT || (centers && ncls > 0), "tn_head_rows: centered heads need centers");
    TN_REQUIRE(y != nullptr || (rowloss == nullptr && da == nullptr && rowstat == nullptr),
               "tn_head_rows: labels required for loss / gradient outputs");
    const size_t lds = head >= HEAD_LOGIT ? (size_t)4 * (ncls + 1) * sizeof(float) : 0;
    TN_REQUIRE(lds <= 32 * 1024, "tn_head_rows: too many classes (%d)", ncls);
    head_rows_kernel<<<cdiv(B, 4), 256, lds, ctx->stream>>>(head, loss, loss_param, a, centers, ncls, y, y_row0, d_row0, feat,
                                                         logprob, rowloss, pred, rowstat, da, dcenters, B, n, inv_batch,
                                                         junk_dist, act, act_param);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
