// Shared internals of libtheanet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/theanet_hip.h"

// arguments of the elastic field computation (elastic_field.h)
struct ElField {       // arguments of the field computation (shared by its two launchers)
    const float* draws_in;
    float* draws_out;
    uint32_t k0, k1, step;
    const uint32_t* d_step;
    int h, w;
    double translation, zoom, magnitude;
    int sigma;
    double angle;
    int nearest;
    int32_t* map_idx;
    float* map_fy;
    float* map_fx;
    double* target;
};

// one finishing reduction out[i] = sum_{s<S} src[s*stride + i] (reduce.hip)
#define TN_RED_MAX 32
struct tn_red_rec {
    const float* src;
    float* out;
    uint32_t n, S, stride, flip;
};

struct tn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;          // the stream ops are currently issued on
    hipStream_t streams[2] = {nullptr, nullptr};   // [0] main, [1] side (leaf work: weight gradients)
    hipStream_t copy_stream = nullptr;     // device-to-host copies that run under later kernels (tn_d2h_early)
    hipEvent_t copy_ev = nullptr;
    hipEvent_t sync_ev[2] = {nullptr, nullptr};
    hipStream_t comm_stream = nullptr;     // collectives that travel beside the backward pass (tn_allreduce_sum_async)
    hipEvent_t comm_ev = nullptr;
    int num_cus = 256;
    // heavy launches since the second stream was last selected: small = two steps in flight share the GPU
    // (tn_fc_bwd then leaves the other stream's kernels a share of the register file), large = this stream has it alone
    int heavy_since_side = 1 << 20;
    // matmul operand precision of the 3x3 conv products (tn_set_matmul_dtype): 0 fp32, 1 fp16 operands /
    // fp32 accumulate; grad_scale: power of two applied to dz before it is rounded to fp16
    int mm_f16 = 0;
    int fc_b3 = 0;                         // tn_set_fc_matmul: 1 = FC products as bf16 triplets (gemm_b3.hip)
    float grad_scale = 1.f;
    char err[512] = {0};
    // RCCL (loaded lazily, comm.hip)
    void* rccl_lib = nullptr;
    void* comm = nullptr;
    int rank = 0, world = 1;
    // small persistent scratch (reductions)
    float* scratch = nullptr;              // scratch of the stream ops are currently issued on
    size_t scratch_bytes = 0;
    float* scratch_slot[2] = {nullptr, nullptr};   // parked scratch of the other stream (tn_stream_select):
    size_t scratch_slot_bytes[2] = {0, 0};         // two pipelined steps never share slab memory
    void* tmp[2] = {nullptr, nullptr};             // per stream: a tensor that lives from one launch to the next (tn_tmp_get)
    size_t tmp_bytes[2] = {0, 0};
    // ... and its parked deferral window: a pipelined step leaves its slab sums pending, the update that
    // opens the stream's NEXT step folds them in (tn_sgd_update_net, TN_UPD_PIPE)
    bool defer_slot[2] = {false, false};
    size_t scratch_off_slot[2] = {0, 0};
    int npend_slot[2] = {0, 0};
    tn_red_rec pend_slot[2][TN_RED_MAX];
    // deferred finishing reductions (reduce.hip)
    bool defer = false;
    size_t scratch_off = 0;
    int npend = 0;
    tn_red_rec pend[TN_RED_MAX];
    unsigned long long scratch_gen = 0;    // bumped by every tn_scratch_get: "nobody has asked for scratch since" checks
    // a light independent job waiting for a heavy launch to ride in (tn_rider_elastic_field)
    bool rider_valid = false;
    ElField rider;
    size_t rider_lds = 0;
};

int tn_scratch_get(tn_ctx* ctx, size_t bytes, float** out);
// a buffer of the current stream that only has to survive until the launches enqueued right after it have run
int tn_tmp_get(tn_ctx* ctx, size_t bytes, float** out);
int tn_tmp_get(tn_ctx* ctx, size_t bytes, float** out);
int tn_red_push(tn_ctx* ctx, const float* src, float* out, uint32_t n, uint32_t S, uint32_t stride,
                uint32_t flip);
int tn_red_commit(tn_ctx* ctx);
// MATMUL 'bf16x3' products of a fully-connected layer (gemm_b3.hip); the tn_fc_* entry points dispatch here when
// tn_set_fc_matmul(ctx, 1) is in force and tn_b3_fc_ok says the shape qualifies
int tn_b3_fc_ok(const float* x, const float* W, int B, int n_in, int n_out);
int tn_b3_fc_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B, int n_in, int n_out, int act,
                 float prm, const uint8_t* mask);
int tn_b3_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int B, int n_in, int n_out, const float* prev_a,
                   int act, float prm, const uint8_t* mask);
int tn_b3_fc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int B, int n_in, int n_out, float* ws,
                   int S);
int tn_red_flush(tn_ctx* ctx);
int tn_red_flush_inc(tn_ctx* ctx, uint32_t* inc);

extern char g_tn_err[512];

inline int tn_fail(tn_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_tn_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define TN_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return tn_fail(ctx, TN_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,    \
                           hipGetErrorString(e_));                                        \
    } while (0)

#define TN_LAUNCH_CHECK()                                                                 \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess)                                                             \
            return tn_fail(ctx, TN_E_HIP, "%s:%d launch -> %s", __FILE__, __LINE__,       \
                           hipGetErrorString(e_));                                        \
    } while (0)

#define TN_REQUIRE(cond, ...)                                                             \
    do {                                                                                  \
        if (!(cond)) return tn_fail(ctx, TN_E_ARG, __VA_ARGS__);                          \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------
// activations (theanet/layer/layer.py:27-39).  The backward pass only keeps the
// layer OUTPUT a, so act'(z) is expressed through a:
//   leaky(s): a>0 -> 1 ; a<0 -> s ; a==0 -> 1+s  (Theano's Maximum/Minimum tie rule;
//             for s==0 an exact a==0 is read as z<0 -> 0: documented in DESIGN.md)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float tn_act_fwd(float z, int act, float prm) {
    switch (act) {
        case TN_ACT_LEAKY: return fmaxf(0.f, z) + fminf(0.f, z) * prm;
        case TN_ACT_TANH: return tanhf(z);
        case TN_ACT_SIGMOID: return 1.f / (1.f + expf(-z));
        case TN_ACT_SOFTPLUS: return z > 20.f ? z : log1pf(expf(z));
        case TN_ACT_SCALED_TANH: return 1.7f * tanhf(2.f * z / 3.f);
        default: return z;
    }
}

__device__ __forceinline__ float tn_act_grad_from_out(float a, int act, float prm) {
    switch (act) {
        case TN_ACT_LEAKY:
            if (a > 0.f) return 1.f;
            if (a < 0.f) return prm;
            return prm > 0.f ? 1.f + prm : 0.f;
        case TN_ACT_TANH: return 1.f - a * a;
        case TN_ACT_SIGMOID: return a * (1.f - a);
        case TN_ACT_SOFTPLUS: return 1.f - expf(-a);
        case TN_ACT_SCALED_TANH: {
            float t = a * (1.f / 1.7f);
            return (1.7f * 2.f / 3.f) * (1.f - t * t);
        }
        default: return 1.f;
    }
}

// Four values behind ONE wave-uniform test of the activation kind.  Epilogues that called the two functions above
// per element paid the whole switch every time (fc_skinny_softmax_train's input-gradient phase: 401 scalar branches
// for 32 elements, 5.5 of the block's 12.6 us); the leaky-ReLU family -- every default of the reference
// (convpool.py:19, hidden.py:16) -- is straight-line code here, same expressions, same bits.
// LK (the caller has established act is TN_ACT_LEAKY or TN_ACT_LINEAR): the transcendental kinds are not even compiled in.
// The GEMM epilogues call these once per unrolled pass; with every kind inlined each time a GEMM kernel was 150-300 KB
// of code, its leaky-ReLU path a chain of jumps over tanhf / expf bodies -- an epilogue of ~10 k cycles, most of them
// instruction-cache misses (round 5, tools/dbg_gemm.py).
template <bool LK = false>
__device__ __forceinline__ void tn_act_fwd4(float4& v, int act, float prm) {
    if (act == TN_ACT_LEAKY) {
        v.x = fmaxf(0.f, v.x) + fminf(0.f, v.x) * prm;
        v.y = fmaxf(0.f, v.y) + fminf(0.f, v.y) * prm;
        v.z = fmaxf(0.f, v.z) + fminf(0.f, v.z) * prm;
        v.w = fmaxf(0.f, v.w) + fminf(0.f, v.w) * prm;
    } else if (!LK && act != TN_ACT_LINEAR) {
        v.x = tn_act_fwd(v.x, act, prm);
        v.y = tn_act_fwd(v.y, act, prm);
        v.z = tn_act_fwd(v.z, act, prm);
        v.w = tn_act_fwd(v.w, act, prm);
    }
}
// s *= act'(a), a = the layer's OUTPUT
template <bool LK = false>
__device__ __forceinline__ void tn_act_grad4(float4& s, const float4& a, int act, float prm) {
    if (act == TN_ACT_LEAKY) {
        const float tie = prm > 0.f ? 1.f + prm : 0.f;
        s.x *= a.x > 0.f ? 1.f : (a.x < 0.f ? prm : tie);
        s.y *= a.y > 0.f ? 1.f : (a.y < 0.f ? prm : tie);
        s.z *= a.z > 0.f ? 1.f : (a.z < 0.f ? prm : tie);
        s.w *= a.w > 0.f ? 1.f : (a.w < 0.f ? prm : tie);
    } else if (!LK && act != TN_ACT_LINEAR) {
        s.x *= tn_act_grad_from_out(a.x, act, prm);
        s.y *= tn_act_grad_from_out(a.y, act, prm);
        s.z *= tn_act_grad_from_out(a.z, act, prm);
        s.w *= tn_act_grad_from_out(a.w, act, prm);
    }
}

// ---------------------------------------------------------------------------
// The two expressions of the momentum-SGD update (layer.py:82-86), with their roundings spelled out:
// several kernels apply them (one step at a time / lazy slabs / delayed / two steps in flight) and the
// weight trajectories of all schedules must agree bit for bit, so the compiler must not be free to
// contract them differently from kernel to kernel.
//   v' = m*v + (1-m)*g  := fma(m, v, rn((1-m)*g))          p' = p - step*v  := fma(-step, v, p)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float tn_vel(float m, float v, float g) { return __fmaf_rn(m, v, __fmul_rn(1.f - m, g)); }
__device__ __forceinline__ float tn_stepped(float p, float step, float v) { return __fmaf_rn(-step, v, p); }

// ---------------------------------------------------------------------------
// Philox4x32-10 counter RNG: key = seed, counter = (lo(idx), hi(idx), step, stream)
// ---------------------------------------------------------------------------
struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2,
                                                     uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += W0;
        k1 += W1;
    }
    return {c0, c1, c2, c3};
}

// uniform in [0,1) with 24 bits
__host__ __device__ __forceinline__ float tn_u01(uint32_t r) { return (r >> 8) * (1.0f / 16777216.0f); }

enum { TN_STREAM_DROPOUT = 1, TN_STREAM_FLIP = 2, TN_STREAM_ELASTIC = 3, TN_STREAM_DEFORMER = 4 };
