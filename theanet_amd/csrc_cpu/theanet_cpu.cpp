// libtheanet_cpu.so -- the C++/OpenMP CPU backend behind the SAME C-ABI as libtheanet_hip.so
// (include/theanet_hip.h; SURVEY.md 7 step 2, 8b last row, 8d "CPU baseline beside it").
//
// What it is for: (1) BASELINE.json configs[0] -- the reference's own CPU-runnable case (mnist.prms,
// small batches) through the drop-in Python surface in a GPU-less container; (2) the timed CPU baseline
// of bench.py (the algorithm Theano's CPU path uses for these layers: im2col + blocked SGEMM conv
// (CorrMM), C loops for max-pool, SGEMM for the fully-connected layers); (3) exercising the host
// logic -- including the data-parallel step with world_size 2 -- where there is no GPU.
//
// What it is NOT: a fallback.  theanet_amd loads it only when THEANET_BACKEND=cpu is set explicitly;
// with the HIP library or the GPU missing the product still fails loudly.  It neither links nor calls
// anything under oracle/ (tests compare the two).
//
// Every op is synchronous ("streams", events and deferral windows are trivial here).  The fused
// MI355X-specific entry points (conv+pool blocks, LDS-resident backward, elastic+conv fusion) answer
// their capability queries with 0 -- the host then issues the unfused ops, exactly as the header
// specifies -- and return TN_E_ARG if called anyway.  Random streams use the same counter-based
// Philox4x32-10 keyed by (seed, step, GLOBAL element index) as the GPU kernels, so dropout masks,
// flip noise and elastic fields are the same bits on both backends.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <omp.h>

#include "../../include/theanet_hip.h"

struct tn_ctx {
    char err[512] = {0};
    int rank = 0, world = 1;
    int mm_f16 = 0;
    float grad_scale = 1.f;
};
static char g_err[512] = {0};

static int fail(tn_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}
#define REQUIRE(cond, ...) do { if (!(cond)) return fail(ctx, TN_E_ARG, __VA_ARGS__); } while (0)
#define NOT_HERE(name) return fail(ctx, TN_E_ARG, name ": not provided by the CPU backend (its capability query answers 0; use the unfused ops)")

// ---- activations (theanet/layer/layer.py:27-39); the backward pass keeps only the layer OUTPUT ----
static inline float act_fwd(float z, int act, float prm) {
    switch (act) {
        case TN_ACT_LEAKY: return std::fmax(0.f, z) + std::fmin(0.f, z) * prm;
        case TN_ACT_TANH: return std::tanh(z);
        case TN_ACT_SIGMOID: return 1.f / (1.f + std::exp(-z));
        case TN_ACT_SOFTPLUS: return z > 20.f ? z : std::log1p(std::exp(z));
        case TN_ACT_SCALED_TANH: return 1.7f * std::tanh(2.f * z / 3.f);
        default: return z;
    }
}
static inline float act_grad_from_out(float a, int act, float prm) {
    switch (act) {
        case TN_ACT_LEAKY:
            if (a > 0.f) return 1.f;
            if (a < 0.f) return prm;
            return prm > 0.f ? 1.f + prm : 0.f;      // Theano's Maximum/Minimum tie rule at exactly 0
        case TN_ACT_TANH: return 1.f - a * a;
        case TN_ACT_SIGMOID: return a * (1.f - a);
        case TN_ACT_SOFTPLUS: return 1.f - std::exp(-a);
        case TN_ACT_SCALED_TANH: { const float t = a * (1.f / 1.7f); return (1.7f * 2.f / 3.f) * (1.f - t * t); }
        default: return 1.f;
    }
}

// ---- Philox4x32-10: key = seed, counter = (lo(idx), hi(idx), step, stream) -- as csrc/common.h ----
struct u32x4 { uint32_t x, y, z, w; };
static inline u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    return {c0, c1, c2, c3};
}
static inline float u01(uint32_t r) { return (r >> 8) * (1.0f / 16777216.0f); }
enum { STREAM_DROPOUT = 1, STREAM_FLIP = 2, STREAM_ELASTIC = 3, STREAM_DEFORMER = 4 };
static inline uint32_t philox_word(uint64_t e, uint32_t st, uint32_t stream, uint64_t seed) {
    const uint64_t cq = e >> 2;
    const u32x4 r = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), st, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    return w[e & 3];
}

// ---- blocked SGEMM building blocks (row-major, contiguous) ---------------------------------------
// C[M,N] (+)= A[M,K] * B[K,N]; single thread (callers parallelise over images / row blocks)
static void gemm_nn_1(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                      bool accumulate) {
    const int KC = 256, NC = 1024;
    if (!accumulate)
        for (int i = 0; i < M; ++i) std::memset(C + (size_t)i * ldc, 0, sizeof(float) * N);
    for (int j0 = 0; j0 < N; j0 += NC) {
        const int jn = std::min(NC, N - j0);
        for (int k0 = 0; k0 < K; k0 += KC) {
            const int kn = std::min(KC, K - k0);
            for (int i = 0; i < M; ++i) {
                float* c = C + (size_t)i * ldc + j0;
                const float* a = A + (size_t)i * lda + k0;
                int k = 0;
                for (; k + 4 <= kn; k += 4) {          // four B rows per pass over the C row
                    const float a0 = a[k], a1 = a[k + 1], a2 = a[k + 2], a3 = a[k + 3];
                    const float* b0 = B + (size_t)(k0 + k) * ldb + j0;
                    const float *b1 = b0 + ldb, *b2 = b1 + ldb, *b3 = b2 + ldb;
#pragma omp simd
                    for (int j = 0; j < jn; ++j) c[j] += (a0 * b0[j] + a1 * b1[j]) + (a2 * b2[j] + a3 * b3[j]);
                }
                for (; k < kn; ++k) {
                    const float a0 = a[k];
                    const float* b0 = B + (size_t)(k0 + k) * ldb + j0;
#pragma omp simd
                    for (int j = 0; j < jn; ++j) c[j] += a0 * b0[j];
                }
            }
        }
    }
}
// the same, rows of C split over the OpenMP team
static void gemm_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    bool accumulate) {
    const int MB = 16;
#pragma omp parallel for schedule(dynamic)
    for (int i0 = 0; i0 < M; i0 += MB)
        gemm_nn_1(std::min(MB, M - i0), N, K, A + (size_t)i0 * lda, lda, B, ldb, C + (size_t)i0 * ldc, ldc, accumulate);
}
static void transpose(const float* A, int rows, int cols, float* T) {      // T[cols][rows]
#pragma omp parallel for
    for (int j0 = 0; j0 < cols; j0 += 32)
        for (int i0 = 0; i0 < rows; i0 += 32)
            for (int j = j0; j < std::min(cols, j0 + 32); ++j)
                for (int i = i0; i < std::min(rows, i0 + 32); ++i) T[(size_t)j * rows + i] = A[(size_t)i * cols + j];
}

extern "C" {

// ================================== lifecycle / memory ==================================
int tn_version(void) { return 100; }
int tn_device_count(int* count) { if (count) *count = 1; return TN_OK; }
int tn_ctx_create(int device, tn_ctx** out) {
    if (!out) return fail(nullptr, TN_E_ARG, "tn_ctx_create: out is NULL");
    (void)device;
    *out = new tn_ctx();
    return TN_OK;
}
int tn_ctx_destroy(tn_ctx* ctx) { delete ctx; return TN_OK; }
const char* tn_last_error(tn_ctx* ctx) { return ctx ? ctx->err : g_err; }
int tn_sync(tn_ctx*) { return TN_OK; }
int tn_stream_select(tn_ctx* ctx, int idx) { REQUIRE(idx == 0 || idx == 1, "tn_stream_select: idx %d", idx); return TN_OK; }
int tn_stream_wait(tn_ctx*, int, int) { return TN_OK; }
int tn_device_info(tn_ctx*, char* name, int name_len, int* cus, size_t* hbm_bytes) {
    if (name && name_len > 0) snprintf(name, name_len, "CPU backend (OpenMP, %d threads)", omp_get_max_threads());
    if (cus) *cus = omp_get_max_threads();
    if (hbm_bytes) *hbm_bytes = 0;
    return TN_OK;
}
int tn_set_matmul_dtype(tn_ctx* ctx, int dtype, float grad_scale) {
    REQUIRE(dtype == 0, "tn_set_matmul_dtype: the CPU backend computes in float32 only (the reference's floatX)");
    (void)grad_scale;
    return TN_OK;
}
int tn_get_matmul_dtype(tn_ctx*) { return 0; }
int tn_set_fc_matmul(tn_ctx* ctx, int mode) {
    REQUIRE(mode == 0, "tn_set_fc_matmul: the CPU backend computes the dense products in float32 only");
    return TN_OK;
}
// DTYPE 'float16' on fp16-resident tensors (conv_c8.hip): MI355X only; the capability queries answer 0 and
// tn_set_matmul_dtype refuses the mode, so the host never gets here
int tn_c8_conv_supported(int, int, int, int, int, int, int, int) { return 0; }
int tn_c8_conv_wgrad_supported(int, int, int, int, int) { return 0; }
size_t tn_c8_wt_elems(int, int, int) { return 0; }
int tn_c8_arrange_multi(tn_ctx* ctx, const tn_c8_wt_seg*, int) { NOT_HERE("tn_c8_arrange_multi"); }
int tn_c8_conv_fwd(tn_ctx* ctx, const void*, const float*, const float*, void*, uint8_t*, int, int, int, int, int, int,
                   float, int, const void*) { NOT_HERE("tn_c8_conv_fwd"); }
int tn_c8_conv_dgrad(tn_ctx* ctx, const void*, const float*, void*, int, int, int, int, int, const void*, int, float, int,
                     const uint8_t*, const void*) { NOT_HERE("tn_c8_conv_dgrad"); }
int tn_c8_conv_wgrad(tn_ctx* ctx, const void*, const void*, float*, float*, int, int, int, int, int, int,
                     const uint8_t*) { NOT_HERE("tn_c8_conv_wgrad"); }
int tn_c8_fc_supported(int, int, int, int) { return 0; }
int tn_c8_fc_fwd(tn_ctx* ctx, const void*, const float*, const float*, float*, int, int, int, int, int, float,
                 const uint8_t*) { NOT_HERE("tn_c8_fc_fwd"); }
int tn_c8_fc_fwd_dropout(tn_ctx* ctx, const void*, const float*, const float*, float*, int, int, int, int, int, float,
                         uint8_t*, float, uint64_t, uint32_t, const uint32_t*, uint64_t) { NOT_HERE("tn_c8_fc_fwd_dropout"); }
int tn_c8_fc_dgrad(tn_ctx* ctx, const float*, const float*, void*, int, int, int, int, const void*, int, float) {
    NOT_HERE("tn_c8_fc_dgrad");
}
int tn_c8_fc_wgrad(tn_ctx* ctx, const void*, const float*, float*, float*, int, int, int, int) { NOT_HERE("tn_c8_fc_wgrad"); }
int tn_c8_pack(tn_ctx* ctx, const float*, int64_t, void*, int, int, int, float) { NOT_HERE("tn_c8_pack"); }
int tn_c8_elastic_apply(tn_ctx* ctx, const float*, int64_t, const int64_t*, void*, int, int, int, int, int, int, const int32_t*,
                        const float*, const float*, float, const uint8_t*, uint64_t, uint32_t, const uint32_t*, int64_t) {
    NOT_HERE("tn_c8_elastic_apply");
}
int tn_c8_unpack(tn_ctx* ctx, const void*, float*, int, int, int, float) { NOT_HERE("tn_c8_unpack"); }

int tn_alloc(tn_ctx* ctx, size_t bytes, void** dptr) {
    void* p = nullptr;
    if (posix_memalign(&p, 256, bytes ? ((bytes + 255) & ~(size_t)255) : 256)) return fail(ctx, TN_E_NOMEM, "tn_alloc(%zu) failed", bytes);
    *dptr = p;
    return TN_OK;
}
int tn_free(tn_ctx*, void* p) { free(p); return TN_OK; }
int tn_h2d(tn_ctx*, void* d, const void* s, size_t n) { std::memcpy(d, s, n); return TN_OK; }
int tn_d2h(tn_ctx*, void* d, const void* s, size_t n) { std::memcpy(d, s, n); return TN_OK; }
int tn_host_alloc(tn_ctx* ctx, size_t bytes, void** out) {
    REQUIRE(out != nullptr && bytes > 0, "tn_host_alloc: bad arguments");
    *out = std::malloc(bytes);
    REQUIRE(*out != nullptr, "tn_host_alloc: out of memory");
    return TN_OK;
}
int tn_host_free(tn_ctx*, void* p) { std::free(p); return TN_OK; }
int tn_d2h_early(tn_ctx*, void* d, const void* s, size_t n) { std::memcpy(d, s, n); return TN_OK; }
int tn_copy_sync(tn_ctx*) { return TN_OK; }
int tn_d2h_early_ev(tn_ctx*, void* d, const void* s, size_t n, void*) { std::memcpy(d, s, n); return TN_OK; }
int tn_d2d(tn_ctx*, void* d, const void* s, size_t n) { std::memmove(d, s, n); return TN_OK; }
int tn_memset(tn_ctx*, void* d, int v, size_t n) { std::memset(d, v, n); return TN_OK; }
int tn_set_u32(tn_ctx*, uint32_t* d, uint32_t v) { *d = v; return TN_OK; }
int tn_set_i64(tn_ctx*, int64_t* d, int64_t v) { *d = v; return TN_OK; }
int tn_set_f32(tn_ctx*, float* d, float v) { *d = v; return TN_OK; }
int tn_add_u32(tn_ctx*, uint32_t* d, uint32_t inc) { *d += inc; return TN_OK; }

int tn_graph_begin(tn_ctx* ctx) { return fail(ctx, TN_E_ARG, "tn_graph_begin: no graph capture on the CPU backend"); }
int tn_graph_end(tn_ctx* ctx, void**) { return fail(ctx, TN_E_ARG, "tn_graph_end: no graph capture on the CPU backend"); }
int tn_graph_launch(tn_ctx* ctx, void*) { return fail(ctx, TN_E_ARG, "tn_graph_launch: no graph capture on the CPU backend"); }
int tn_graph_destroy(tn_ctx*, void*) { return TN_OK; }

// events: wall-clock stamps (every op is synchronous)
int tn_event_create(tn_ctx*, void** ev) { *ev = new double(0.0); return TN_OK; }
int tn_event_record(tn_ctx*, void* ev) {
    *static_cast<double*>(ev) = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return TN_OK;
}
int tn_event_wait(tn_ctx*, void*) { return TN_OK; }
int tn_event_elapsed_ms(tn_ctx*, void* a, void* b, float* ms) { *ms = (float)(*static_cast<double*>(b) - *static_cast<double*>(a)); return TN_OK; }
int tn_event_destroy(tn_ctx*, void* ev) { delete static_cast<double*>(ev); return TN_OK; }
int tn_event_sync(tn_ctx*, void*) { return TN_OK; }
int tn_event_query(tn_ctx*, void*, int* done) { *done = 1; return TN_OK; }      // calls are synchronous here

// ================================== conv (im2col + SGEMM) ==================================
int tn_conv_mfma_supported(int, int, int, int) { return 0; }

// col[(c,u,v)][i*Wo+j] = xpad[c][i*s+u][j*s+v]
static void im2col(const float* x, int C, int H, int Wd, int f, int s, int pad, int Ho, int Wo, float* col) {
    for (int c = 0; c < C; ++c)
        for (int u = 0; u < f; ++u)
            for (int v = 0; v < f; ++v) {
                float* dst = col + (size_t)((c * f + u) * f + v) * Ho * Wo;
                for (int i = 0; i < Ho; ++i) {
                    const int yy = i * s + u - pad;
                    if (yy < 0 || yy >= H) { std::memset(dst + (size_t)i * Wo, 0, sizeof(float) * Wo); continue; }
                    const float* src = x + ((size_t)c * H + yy) * Wd;
                    for (int j = 0; j < Wo; ++j) {
                        const int xx = j * s + v - pad;
                        dst[(size_t)i * Wo + j] = (xx >= 0 && xx < Wd) ? src[xx] : 0.f;
                    }
                }
            }
}
// flipped, flattened weights: Wf[k][(c,u,v)] = W[k][c][f-1-u][f-1-v]   (true convolution, convpool.py:54)
static void flip_weights(const float* W, int K, int C, int f, float* Wf) {
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < C; ++c)
            for (int u = 0; u < f; ++u)
                for (int v = 0; v < f; ++v)
                    Wf[((size_t)k * C + c) * f * f + u * f + v] = W[(((size_t)k * C + c) * f + (f - 1 - u)) * f + (f - 1 - v)];
}

int tn_conv2d_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int N, int C, int H, int Wd,
                  int K, int f, int stride, int pad_lo, int Ho, int Wo, int act, float act_param) {
    REQUIRE(N > 0 && C > 0 && K > 0 && f > 0 && stride > 0 && Ho > 0 && Wo > 0, "tn_conv2d_fwd: bad shape");
    const int Kd = C * f * f, HW = Ho * Wo;
    std::vector<float> Wf((size_t)K * Kd);
    flip_weights(W, K, C, f, Wf.data());
#pragma omp parallel
    {
        std::vector<float> col((size_t)Kd * HW);
#pragma omp for schedule(dynamic)
        for (int n = 0; n < N; ++n) {
            im2col(x + (size_t)n * C * H * Wd, C, H, Wd, f, stride, pad_lo, Ho, Wo, col.data());
            float* out = a + (size_t)n * K * HW;
            gemm_nn_1(K, HW, Kd, Wf.data(), Kd, col.data(), HW, out, HW, false);
            for (int k = 0; k < K; ++k) {
                const float bk = b[k];
                float* o = out + (size_t)k * HW;
                for (int p = 0; p < HW; ++p) o[p] = act_fwd(o[p] + bk, act, act_param);
            }
        }
    }
    return TN_OK;
}

int tn_conv2d_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N, int C, int H, int Wd,
                    int K, int f, int stride, int pad_lo, int Ho, int Wo) {
    REQUIRE(N > 0 && C > 0 && K > 0 && f > 0 && stride > 0, "tn_conv2d_wgrad: bad shape");
    const int Kd = C * f * f, HW = Ho * Wo, T = omp_get_max_threads();
    std::vector<float> part((size_t)T * K * Kd, 0.f), dbp((size_t)T * K, 0.f);
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        std::vector<float> col((size_t)Kd * HW), colT((size_t)HW * Kd);
        float* acc = part.data() + (size_t)t * K * Kd;
        float* dba = dbp.data() + (size_t)t * K;
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            im2col(x + (size_t)n * C * H * Wd, C, H, Wd, f, stride, pad_lo, Ho, Wo, col.data());
            for (int r = 0; r < Kd; ++r)                       // colT[p][r]
                for (int p = 0; p < HW; ++p) colT[(size_t)p * Kd + r] = col[(size_t)r * HW + p];
            const float* dzn = dz + (size_t)n * K * HW;
            gemm_nn_1(K, Kd, HW, dzn, HW, colT.data(), Kd, acc, Kd, true);       // dWf += dz_n . col_n^T
            for (int k = 0; k < K; ++k) {
                float s = 0.f;
                for (int p = 0; p < HW; ++p) s += dzn[(size_t)k * HW + p];
                dba[k] += s;
            }
        }
    }
    std::vector<float> dWf((size_t)K * Kd, 0.f);
    for (int t = 0; t < T; ++t)                                 // fixed order: deterministic
        for (size_t i = 0; i < dWf.size(); ++i) dWf[i] += part[(size_t)t * K * Kd + i];
    flip_weights(dWf.data(), K, C, f, dW);                      // un-flip: the same index map
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += dbp[(size_t)t * K + k];
        db[k] = s;
    }
    return TN_OK;
}

int tn_conv2d_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int N, int C, int H, int Wd, int K, int f,
                    int stride, int pad_lo, int Ho, int Wo, const float* prev_a, int prev_act, float prev_act_param) {
    REQUIRE(N > 0 && C > 0 && K > 0 && f > 0 && stride > 0, "tn_conv2d_dgrad: bad shape");
    const int Kd = C * f * f, HW = Ho * Wo;
    std::vector<float> Wf((size_t)K * Kd), WfT((size_t)Kd * K);
    flip_weights(W, K, C, f, Wf.data());
    for (int k = 0; k < K; ++k)
        for (int r = 0; r < Kd; ++r) WfT[(size_t)r * K + k] = Wf[(size_t)k * Kd + r];
#pragma omp parallel
    {
        std::vector<float> dcol((size_t)Kd * HW);
#pragma omp for schedule(dynamic)
        for (int n = 0; n < N; ++n) {
            gemm_nn_1(Kd, HW, K, WfT.data(), K, dz + (size_t)n * K * HW, HW, dcol.data(), HW, false);
            float* dxn = dx + (size_t)n * C * H * Wd;
            std::memset(dxn, 0, sizeof(float) * C * H * Wd);
            for (int c = 0; c < C; ++c)                          // col2im
                for (int u = 0; u < f; ++u)
                    for (int v = 0; v < f; ++v) {
                        const float* src = dcol.data() + (size_t)((c * f + u) * f + v) * HW;
                        for (int i = 0; i < Ho; ++i) {
                            const int yy = i * stride + u - pad_lo;
                            if (yy < 0 || yy >= H) continue;
                            float* row = dxn + ((size_t)c * H + yy) * Wd;
                            for (int j = 0; j < Wo; ++j) {
                                const int xx = j * stride + v - pad_lo;
                                if (xx >= 0 && xx < Wd) row[xx] += src[(size_t)i * Wo + j];
                            }
                        }
                    }
            if (prev_a && prev_act != TN_ACT_LINEAR) {
                const float* pa = prev_a + (size_t)n * C * H * Wd;
                for (int i = 0; i < C * H * Wd; ++i) dxn[i] *= act_grad_from_out(pa[i], prev_act, prev_act_param);
            }
        }
    }
    return TN_OK;
}

// ---- MI355X-specific fused blocks: not offered here (capability queries answer 0) ----
int tn_convpool_supported(int, int, int, int) { return 0; }
int tn_convpool_fwd(tn_ctx* ctx, const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int,
                    int, int, int, int, int, float) { NOT_HERE("tn_convpool_fwd"); }
int tn_convpool_bwd(tn_ctx* ctx, const float*, const float*, const float*, const float*, float*, float*, float*, int, int,
                    int, int, int, int, int, int, int, int, int, int, int, float) { NOT_HERE("tn_convpool_bwd"); }
int tn_convpool_fwd_mask(tn_ctx* ctx, const float*, const float*, const float*, float*, uint8_t*, int, int, int, int, int,
                         int, int, int, int, int, int, int, int, float) { NOT_HERE("tn_convpool_fwd_mask"); }
int tn_convpool_bwd_mask(tn_ctx* ctx, const float*, const float*, const float*, const uint8_t*, float*, float*, float*,
                         int, int, int, int, int, int, int, int, int, int, int, int, int, float) { NOT_HERE("tn_convpool_bwd_mask"); }
int tn_convblock_supported(int, int, int, int, int, int, int) { return 0; }
int tn_convblock_bwd(tn_ctx* ctx, const float*, const float*, const float*, const float*, float*, float*, float*, int, int,
                     int, int, int, int, int, int, int, int, int, int, int, float) { NOT_HERE("tn_convblock_bwd"); }
int tn_convpool_tile_supported(int, int, int, int, int, int, int, int, int, int, int, int, int) { return 0; }
int tn_convpool_bwd_mask_dx(tn_ctx* ctx, const float*, const float*, const float*, const float*, const uint8_t*, float*,
                            float*, float*, int, int, int, int, int, int, int, int, int, int, int, int, int, float,
                            const float*, int, float) { NOT_HERE("tn_convpool_bwd_mask_dx"); }
int tn_convblock_mask_supported(int, int, int, int, int, int, int, int, int, int, int, int) { return 0; }
int tn_convblock_bwd_mask(tn_ctx* ctx, const float*, const float*, const float*, const float*, const uint8_t*, float*,
                          float*, float*, int, int, int, int, int, int, int, int, int, int, int, int, int, float) {
    NOT_HERE("tn_convblock_bwd_mask");
}
int tn_elastic_convpool_supported(int, int, int, int, int, int, int, int, int, int) { return 0; }
int tn_elastic_convpool_fwd_mask(tn_ctx* ctx, const float*, int64_t, const int64_t*, float*, int, int, int, int, int,
                                 const int32_t*, const float*, const float*, float, const uint8_t*, uint64_t, uint32_t,
                                 const uint32_t*, int64_t, const float*, const float*, float*, uint8_t*, int, int, int, int,
                                 int, int, int, int, int, float) { NOT_HERE("tn_elastic_convpool_fwd_mask"); }

// ================================== pool / mean ==================================
int tn_pool_fwd(tn_ctx*, const float* x, float* y, int NC, int H, int Wd, int p, int Ho, int Wo) {
#pragma omp parallel for
    for (int m = 0; m < NC; ++m) {
        const float* xi = x + (size_t)m * H * Wd;
        float* yo = y + (size_t)m * Ho * Wo;
        for (int i = 0; i < Ho; ++i)
            for (int j = 0; j < Wo; ++j) {
                float best = -INFINITY;
                for (int di = 0; di < p && i * p + di < H; ++di)
                    for (int dj = 0; dj < p && j * p + dj < Wd; ++dj) best = std::fmax(best, xi[(size_t)(i * p + di) * Wd + j * p + dj]);
                yo[(size_t)i * Wo + j] = best;
            }
    }
    return TN_OK;
}
int tn_pool_bwd(tn_ctx*, const float* x, const float* y, const float* dy, float* dx, int NC, int H, int Wd, int p, int Ho,
                int Wo, int prev_act, float prev_act_param) {
#pragma omp parallel for
    for (int m = 0; m < NC; ++m) {
        const float* xi = x + (size_t)m * H * Wd;
        float* dxi = dx + (size_t)m * H * Wd;
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < Wd; ++c) {
                const int i = r / p, j = c / p;
                float g = 0.f;
                if (i < Ho && j < Wo && xi[(size_t)r * Wd + c] == y[((size_t)m * Ho + i) * Wo + j])    // every tie: MaxPoolGrad
                    g = dy[((size_t)m * Ho + i) * Wo + j];
                if (prev_act != TN_ACT_LINEAR) g *= act_grad_from_out(xi[(size_t)r * Wd + c], prev_act, prev_act_param);
                dxi[(size_t)r * Wd + c] = g;
            }
    }
    return TN_OK;
}
int tn_mean_fwd(tn_ctx*, const float* x, float* y, int NC, int HW) {
#pragma omp parallel for
    for (int m = 0; m < NC; ++m) {
        double s = 0.0;
        for (int i = 0; i < HW; ++i) s += x[(size_t)m * HW + i];
        y[m] = (float)(s / HW);
    }
    return TN_OK;
}
int tn_mean_bwd(tn_ctx*, const float* dy, float* dx, int NC, int HW, const float* prev_a, int prev_act, float prm) {
#pragma omp parallel for
    for (int m = 0; m < NC; ++m)
        for (int i = 0; i < HW; ++i) {
            float g = dy[m] / HW;
            if (prev_a && prev_act != TN_ACT_LINEAR) g *= act_grad_from_out(prev_a[(size_t)m * HW + i], prev_act, prm);
            dx[(size_t)m * HW + i] = g;
        }
    return TN_OK;
}

// ================================== fully connected ==================================
int tn_fc_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B, int n_in, int n_out, int act,
              float prm, const uint8_t* mask) {
    REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_fwd: bad shape");
    gemm_nn(B, n_out, n_in, x, n_in, W, n_out, a, n_out, false);
#pragma omp parallel for
    for (int r = 0; r < B; ++r)
        for (int c = 0; c < n_out; ++c) {
            float v = act_fwd(a[(size_t)r * n_out + c] + b[c], act, prm);
            if (mask) v *= (float)mask[(size_t)r * n_out + c];
            a[(size_t)r * n_out + c] = v;
        }
    return TN_OK;
}
size_t tn_fc_wgrad_ws_bytes(int, int, int) { return 256; }
int tn_fc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int B, int n_in, int n_out, void*) {
    REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_wgrad: bad shape");
    std::vector<float> xT((size_t)n_in * B);
    transpose(x, B, n_in, xT.data());
    gemm_nn(n_in, n_out, B, xT.data(), B, dz, n_out, dW, n_out, false);      // dW = x^T . dz
#pragma omp parallel for
    for (int c = 0; c < n_out; ++c) {
        float s = 0.f;
        for (int r = 0; r < B; ++r) s += dz[(size_t)r * n_out + c];
        db[c] = s;
    }
    return TN_OK;
}
int tn_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int B, int n_in, int n_out, const float* prev_a,
                int prev_act, float prm, const uint8_t* prev_mask) {
    REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_dgrad: bad shape");
    std::vector<float> WT((size_t)n_out * n_in);
    transpose(W, n_in, n_out, WT.data());
    gemm_nn(B, n_in, n_out, dz, n_out, WT.data(), n_in, dx, n_in, false);      // dx = dz . W^T
    if ((prev_a && prev_act != TN_ACT_LINEAR) || prev_mask) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)B * n_in; ++i) {
            float g = dx[i];
            if (prev_a && prev_act != TN_ACT_LINEAR) g *= act_grad_from_out(prev_a[i], prev_act, prm);
            if (prev_mask) g *= (float)prev_mask[i];
            dx[i] = g;
        }
    }
    return TN_OK;
}
int tn_fc_bwd(tn_ctx* ctx, const float* x, const float* dz, const float* W, float* dW, float* db, float* dx, int B,
              int n_in, int n_out, void* ws, const float* prev_a, int prev_act, float prm, const uint8_t* prev_mask) {
    int rc = tn_fc_wgrad(ctx, x, dz, dW, db, B, n_in, n_out, ws);
    if (rc) return rc;
    return tn_fc_dgrad(ctx, dz, W, dx, B, n_in, n_out, prev_a, prev_act, prm, prev_mask);
}

// ---- dropout (dropout.py:9-31): mask[i] = uniform(seed, *d_step + step, elem0 + i) >= pdrop ----
int tn_dropout_mask(tn_ctx*, uint8_t* mask, size_t n, float pdrop, uint64_t seed, uint32_t step, const uint32_t* d_step,
                    uint64_t elem0) {
    const uint32_t st = step + (d_step ? *d_step : 0u);
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; ++i) mask[i] = u01(philox_word(elem0 + (uint64_t)i, st, STREAM_DROPOUT, seed)) >= pdrop ? 1 : 0;
    return TN_OK;
}
int tn_fc_fwd_dropout(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B, int n_in, int n_out,
                      int act, float prm, uint8_t* mask_out, float pdrop, uint64_t seed, uint32_t step,
                      const uint32_t* d_step, uint64_t elem0) {
    int rc = tn_dropout_mask(ctx, mask_out, (size_t)B * n_out, pdrop, seed, step, d_step, elem0);
    if (rc) return rc;
    return tn_fc_fwd(ctx, x, W, b, a, B, n_in, n_out, act, prm, mask_out);
}
int tn_scale_mask(tn_ctx*, const float* x, const uint8_t* mask, float scale, float* y, size_t n, const float* prev_a,
                  int prev_act, float prm) {
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; ++i) {
        float v = x[i] * scale;
        if (mask) v *= (float)mask[i];
        if (prev_a && prev_act != TN_ACT_LINEAR) v *= act_grad_from_out(prev_a[i], prev_act, prm);
        y[i] = v;
    }
    return TN_OK;
}

// ================================== softmax + NLL ==================================
int tn_softmax_nll(tn_ctx* ctx, const float* z, const int32_t* y, int64_t y_row0, const int64_t* d_row0, float* logprob,
                   float* rowloss, int32_t* pred, float* rowp, float* dz, int B, int n_out, float inv_batch) {
    REQUIRE(B > 0 && n_out > 0, "tn_softmax_nll: bad shape");
    REQUIRE(y != nullptr || (rowloss == nullptr && dz == nullptr && rowp == nullptr),
            "tn_softmax_nll: labels required for loss/gradient outputs");
    const int64_t yoff = y_row0 + (d_row0 ? *d_row0 : 0);
#pragma omp parallel for
    for (int r = 0; r < B; ++r) {
        const float* zr = z + (size_t)r * n_out;
        float m = -INFINITY;
        int am = 0;
        for (int c = 0; c < n_out; ++c)
            if (zr[c] > m) { m = zr[c]; am = c; }           // first maximum (numpy argmax)
        float s = 0.f;
        for (int c = 0; c < n_out; ++c) s += std::exp(zr[c] - m);
        const float lse = std::log(s);
        const int label = y ? y[yoff + r] : -1;
        for (int c = 0; c < n_out; ++c) {
            const float lp = zr[c] - m - lse;
            if (logprob) logprob[(size_t)r * n_out + c] = lp;
            if (dz) dz[(size_t)r * n_out + c] = (std::exp(lp) - (c == label ? 1.f : 0.f)) * inv_batch;
            if (c == label) {
                if (rowloss) rowloss[r] = -lp;
                if (rowp) rowp[r] = std::exp(lp);
            }
        }
        if (pred) pred[r] = am;
    }
    return TN_OK;
}
size_t tn_softmax_cost_ws_bytes(int B) { return ((size_t)(B + 3) / 4 + 4) * sizeof(float); }
int tn_reduce_sum(tn_ctx*, const float* v, size_t n, float scale, float* out, int accumulate) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += v[i];
    out[0] = (accumulate ? out[0] : 0.f) + scale * (float)s;
    return TN_OK;
}
int tn_softmax_nll_cost(tn_ctx* ctx, const float* z, const int32_t* y, int64_t y_row0, const int64_t* d_row0,
                        float* logprob, float* rowloss, int32_t* pred, float* rowp, float* dz, int B, int n_out,
                        float inv_batch, float cost_scale, float* cost, void*) {
    REQUIRE(rowloss != nullptr, "tn_softmax_nll_cost: rowloss required");
    int rc = tn_softmax_nll(ctx, z, y, y_row0, d_row0, logprob, rowloss, pred, rowp, dz, B, n_out, inv_batch);
    if (rc) return rc;
    return tn_reduce_sum(ctx, rowloss, B, cost_scale, cost, 0);
}
int tn_fc_softmax_nll(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits, int B, int n_in,
                      int n_out, const int32_t* y, int64_t y_row0, const int64_t* d_row0, float* logprob, float* rowloss,
                      int32_t* pred, float* rowp, float* dz, float inv_batch) {
    int rc = tn_fc_fwd(ctx, x, W, b, logits, B, n_in, n_out, TN_ACT_LINEAR, 0.f, nullptr);
    if (rc) return rc;
    return tn_softmax_nll(ctx, logits, y, y_row0, d_row0, logprob, rowloss, pred, rowp, dz, B, n_out, inv_batch);
}
int tn_fc_softmax_train(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits, int B, int n_in,
                        int n_out, const int32_t* y, int64_t y_row0, const int64_t* d_row0, float* logprob,
                        float* rowloss, int32_t* pred, float* rowp, float* dz, float inv_batch, float* dW, float* db,
                        float* dx, void* ws, const float* prev_a, int prev_act, float prm, const uint8_t* prev_mask) {
    int rc = tn_fc_softmax_nll(ctx, x, W, b, logits, B, n_in, n_out, y, y_row0, d_row0, logprob, rowloss, pred, rowp, dz, inv_batch);
    if (rc) return rc;
    return tn_fc_bwd(ctx, x, dz, W, dW, db, dx, B, n_in, n_out, ws, prev_a, prev_act, prm, prev_mask);
}
// ---- the other output heads / losses (outlayers.py:38-64, 105-224): see csrc/heads.hip ----
int tn_head_rows(tn_ctx* ctx, int head, int loss, float loss_param, const float* a, const float* centers, int ncls,
                 const int32_t* y, int64_t y_row0, const int64_t* d_row0, float* feat, float* logprob, float* rowloss,
                 int32_t* pred, float* rowstat, float* da, float* dcenters, int B, int n, float inv_batch, float junk_dist,
                 int act, float act_prm) {
    REQUIRE(B > 0 && n > 0 && a && logprob && head >= 0 && head <= 4 && loss >= 0 && loss <= 4, "tn_head_rows: bad arguments");
    REQUIRE(head < 3 || (centers && ncls > 0), "tn_head_rows: centered heads need centers");
    REQUIRE(y != nullptr || (rowloss == nullptr && da == nullptr && rowstat == nullptr),
            "tn_head_rows: labels required for loss / gradient outputs");
    const int64_t yoff = y_row0 + (d_row0 ? *d_row0 : 0);
    const float eps = 0.001f;
    for (int row = 0; row < B; ++row) {                 // (dcenters accumulates: rows stay sequential)
        const float* ar = a + (size_t)row * n;
        const int label = y ? y[yoff + row] : -1;
        auto argmax = [](const float* v, int m) { int am = 0; for (int c = 1; c < m; ++c) if (v[c] > v[am]) am = c; return am; };
        if (head <= 2) {
            float* lp = logprob + (size_t)row * n;
            if (pred) pred[row] = argmax(ar, n);
            if (head == 2) {
                const float zy = label >= 0 ? ar[label] : 0.f;
                float s = 0.f, cnt = 0.f;
                for (int c = 0; c < n; ++c) {
                    lp[c] = ar[c];
                    if (label < 0) continue;
                    const float t = ar[c] + 1.f - zy;
                    s += std::fmax(0.f, t);
                    const float ac = (c != label && t >= 0.f) ? 1.f : 0.f;
                    cnt += ac;
                    if (da && c != label) da[(size_t)row * n + c] = ac * inv_batch / n;
                }
                if (label >= 0) {
                    if (rowloss) rowloss[row] = s / n;
                    if (rowstat) rowstat[row] = zy;
                    if (da) da[(size_t)row * n + label] = -cnt * inv_batch / n;
                }
                continue;
            }
            float mean = 0.f;
            if (head == 1) { for (int c = 0; c < n; ++c) mean += ar[c]; mean /= n; }
            float m = -INFINITY, s = 0.f;
            for (int c = 0; c < n; ++c) m = std::fmax(m, ar[c] - mean);
            for (int c = 0; c < n; ++c) s += std::exp(ar[c] - mean - m);
            const float lse = std::log(s);
            for (int c = 0; c < n; ++c) { lp[c] = ar[c] - mean - m - lse; if (feat) feat[(size_t)row * n + c] = ar[c] - mean; }
            if (label < 0) continue;
            const float lpy = lp[label], py = std::exp(lpy);
            if (rowstat) rowstat[row] = py;
            float rl = 0.f;
            if (head == 1) {
                const float e = std::exp(-(ar[label] - mean));
                rl = e;
                if (da) for (int c = 0; c < n; ++c) da[(size_t)row * n + c] = -e * inv_batch * ((c == label ? 1.f : 0.f) - 1.f / n);
            } else if (loss <= 2) {
                float gl;
                if (loss == 0) { rl = -lpy; gl = -1.f; }
                else if (loss == 1) { rl = lpy * lpy; gl = 2.f * lpy; }
                else { const float t = loss_param - lpy; rl = std::fmax(0.f, t); gl = t >= 0.f ? -1.f : 0.f; }
                if (da) for (int c = 0; c < n; ++c) da[(size_t)row * n + c] = gl * inv_batch * ((c == label ? 1.f : 0.f) - std::exp(lp[c]));
            } else {
                float dot = 0.f, hs = 0.f;
                auto gp = [&](int c) { return loss == 3 ? (c == label ? -(float)(n - 1) / n : 1.f / n) : (c == label ? -std::exp(-py) : 0.f); };
                for (int c = 0; c < n; ++c) { const float pc = std::exp(lp[c]); hs += std::fmax(0.f, pc + 1.f - py); dot += gp(c) * pc; }
                rl = loss == 3 ? hs / n : std::exp(-py);
                if (da) for (int c = 0; c < n; ++c) da[(size_t)row * n + c] = inv_batch * std::exp(lp[c]) * (gp(c) - dot);
            }
            if (rowloss) rowloss[row] = rl;
            continue;
        }
        const int ncol = head == 4 ? ncls + 1 : ncls;
        float* lp = logprob + (size_t)row * ncol;
        for (int k = 0; k < ncls; ++k) {
            const float* ck = centers + (size_t)k * n;
            float s = 0.f;
            for (int f = 0; f < n; ++f) {
                if (head == 3) { const float v = ar[f] * (1.f - 2.f * eps) + eps; s += std::log(ck[f] * v + (1.f - ck[f]) * (1.f - v)); }
                else { const float d = ar[f] - ck[f]; s += d * d; }
            }
            lp[k] = head == 3 ? s : -s;
        }
        if (head == 4) {
            lp[ncls] = -junk_dist;
            float m = -INFINITY, s = 0.f;
            for (int c = 0; c < ncol; ++c) m = std::fmax(m, lp[c]);
            for (int c = 0; c < ncol; ++c) s += std::exp(lp[c] - m);
            const float lse = std::log(s);
            for (int c = 0; c < ncol; ++c) lp[c] = lp[c] - m - lse;
        }
        if (pred) pred[row] = argmax(lp, ncol);
        if (label < 0) continue;
        if (rowloss) rowloss[row] = -lp[label];
        const float* cy = centers + (size_t)label * n;
        if (head == 3) {
            float wrong = 0.f;
            for (int f = 0; f < n; ++f) {
                const float v = ar[f] * (1.f - 2.f * eps) + eps, bp = cy[f] * v + (1.f - cy[f]) * (1.f - v);
                wrong += bp < .5f ? 1.f : 0.f;
                if (da) da[(size_t)row * n + f] = -inv_batch * (1.f - 2.f * eps) * (2.f * cy[f] - 1.f) / bp * act_grad_from_out(ar[f], act, act_prm);
            }
            if (rowstat) rowstat[row] = wrong / n;
        } else {
            if (rowstat) rowstat[row] = std::exp(lp[label]);
            for (int f = 0; f < n; ++f) {
                float gv = 0.f;
                for (int k = 0; k < ncls; ++k) {
                    const float w = ((k == label ? 1.f : 0.f) - std::exp(lp[k])) * inv_batch, d = ar[f] - centers[(size_t)k * n + f];
                    gv += w * 2.f * d;
                    if (dcenters) dcenters[(size_t)k * n + f] += -w * 2.f * d;
                }
                if (da) da[(size_t)row * n + f] = gv * act_grad_from_out(ar[f], act, act_prm);
            }
        }
    }
    return TN_OK;
}

int tn_wtcost(tn_ctx*, const float* p, size_t n, float L1, float L2, float* out, int accumulate) {
    double a = 0.0, s = 0.0;
    for (size_t i = 0; i < n; ++i) { a += std::fabs(p[i]); s += (double)p[i] * p[i]; }
    out[0] = (accumulate ? out[0] : 0.f) + (float)(L1 * a + L2 * s);
    return TN_OK;
}
int tn_error_stats(tn_ctx*, const int32_t* pred, const int32_t* y, int64_t y_row0, const float* rowp, int B, float* out2) {
    double e = 0.0, p = 0.0;
    for (int i = 0; i < B; ++i) { e += pred[i] != y[y_row0 + i]; p += rowp[i]; }
    out2[0] = (float)(e / B);
    out2[1] = (float)(p / B);
    return TN_OK;
}

// ================================== reductions window (trivial: every op finishes its own sums) ==========
int tn_defer_reductions(tn_ctx*, int) { return TN_OK; }
int tn_defer_flush_step(tn_ctx*, uint32_t* d_step) { if (d_step) *d_step += 1; return TN_OK; }
int tn_defer_discard(tn_ctx*) { return TN_OK; }

// ================================== momentum SGD + maxnorm (layer.py:70-107) ==================================
static void sgd_seg(float* p, float* v, const float* g, size_t n, float m, float rate, float lr, float L1, float L2,
                    float gscale) {
    const float step = rate * lr;
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; ++i) {
        const float pv = p[i], vv = v[i];
        float gg = g[i] * gscale;
        if (L1 != 0.f) gg += L1 * ((pv > 0.f) - (pv < 0.f));
        if (L2 != 0.f) gg += 2.f * L2 * pv;
        v[i] = m * vv + (1.f - m) * gg;
        p[i] = pv - step * vv;                     // the OLD velocity moves p (simultaneous Theano updates)
    }
}
int tn_sgd_update(tn_ctx*, float* p, float* v, const float* g, size_t n, float momentum, float rate, const float* d_lr,
                  float L1, float L2, float gscale) {
    sgd_seg(p, v, g, n, momentum, rate, d_lr[0], L1, L2, gscale);
    return TN_OK;
}
int tn_maxnorm(tn_ctx* ctx, float* p, int ndim, int d0, int rest, float maxnorm) {
    if (maxnorm <= 0.f) return TN_OK;
    if (ndim == 1) {
        for (int i = 0; i < d0; ++i) p[i] = std::fmin(std::fmax(p[i], -maxnorm), maxnorm);
    } else if (ndim == 2) {                         // (rows x cols): per-column L2 norm
#pragma omp parallel for
        for (int c = 0; c < rest; ++c) {
            float s = 0.f;
            for (int r = 0; r < d0; ++r) s += p[(size_t)r * rest + c] * p[(size_t)r * rest + c];
            const float nrm = std::sqrt(s), sc = (1e-7f + std::fmin(std::fmax(nrm, 0.f), maxnorm)) / (1e-7f + nrm);
            for (int r = 0; r < d0; ++r) p[(size_t)r * rest + c] *= sc;
        }
    } else if (ndim == 4) {                         // (d0 x rest): per-output-kernel norm
#pragma omp parallel for
        for (int k = 0; k < d0; ++k) {
            float s = 0.f;
            for (int i = 0; i < rest; ++i) s += p[(size_t)k * rest + i] * p[(size_t)k * rest + i];
            const float nrm = std::sqrt(s), sc = (1e-7f + std::fmin(std::fmax(nrm, 0.f), maxnorm)) / (1e-7f + nrm);
            for (int i = 0; i < rest; ++i) p[(size_t)k * rest + i] *= sc;
        }
    } else {
        return fail(ctx, TN_E_ARG, "tn_maxnorm: ndim %d", ndim);
    }
    return TN_OK;
}
static int upd_cost(tn_ctx* ctx, const tn_sgd_seg* segs, int nseg, const float* d_lr, float gscale, uint32_t* d_step_inc,
                    const float* rowloss, int nrow, float cost_scale, float* d_cost) {
    for (int s = 0; s < nseg; ++s)
        sgd_seg(segs[s].p, segs[s].v, segs[s].g, segs[s].n, segs[s].momentum, segs[s].rate, d_lr[0], segs[s].L1, segs[s].L2, gscale);
    if (d_step_inc) *d_step_inc += 1;
    if (rowloss) return tn_reduce_sum(ctx, rowloss, nrow, cost_scale, d_cost, 0);
    return TN_OK;
}
static int upd_delayed(tn_ctx* ctx, const tn_sgd_seg* segs, int nseg, const float* d_lr, float gscale, uint32_t* d_step_inc,
                       int mode) {
    REQUIRE(nseg > 0 && segs && d_lr && mode >= 1 && mode <= 3, "tn_sgd_update_net (delayed): bad arguments");
    for (int s = 0; s < nseg; ++s) {
        const tn_sgd_seg& sg = segs[s];
        const float step = sg.rate * d_lr[0], m = sg.momentum;
#pragma omp parallel for
        for (long long i = 0; i < (long long)sg.n; ++i) {
            float vv = sg.v[i];
            if (mode != 2) { vv = m * vv + (1.f - m) * (sg.g[i] * gscale); sg.v[i] = vv; }
            if (mode != 3) sg.p[i] -= step * vv;
        }
    }
    if (d_step_inc) *d_step_inc += 1;
    return TN_OK;
}
static int upd_pipe(tn_ctx* ctx, const tn_pipe_seg* segs, int nseg, const float* d_lr, uint32_t* d_step, uint32_t step_inc,
                    int update_v, const float* rowloss, int nrow, float cost_scale, float* d_cost) {
    REQUIRE(nseg > 0 && segs && d_lr, "tn_sgd_update_net (pipe): bad arguments");
    if (rowloss) tn_reduce_sum(ctx, rowloss, nrow, cost_scale, d_cost, 0);      // the previous step's cost
    for (int s = 0; s < nseg; ++s) {
        const tn_pipe_seg& sg = segs[s];
        const float step = sg.rate * d_lr[0], m = sg.momentum;
#pragma omp parallel for
        for (long long i = 0; i < (long long)sg.n; ++i) {
            float vv = sg.v[i];
            if (update_v) { vv = m * vv + (1.f - m) * sg.g[i]; sg.v[i] = vv; }
            sg.p[i] = sg.psrc[i] - step * vv;
        }
    }
    if (d_step) *d_step += step_inc;
    return TN_OK;
}
// one entry point, the schedule's form in `mode` (this library finishes every slab sum where it is produced: LAZY == PLAIN)
int tn_sgd_update_net(tn_ctx* ctx, int mode, const void* d_segs, const void*, int nseg, size_t, const float* d_lr, float gscale,
                      uint32_t* d_step, uint32_t step_inc, int flags, const float* rowloss, int nrow, float cost_scale,
                      float* d_cost) {
    REQUIRE(mode == TN_UPD_PIPE || d_step == nullptr || step_inc == 1,
            "tn_sgd_update_net: the step counter advances by one outside the pipelined schedule");
    switch (mode) {
        case TN_UPD_PLAIN:
        case TN_UPD_LAZY:
            REQUIRE((nseg <= 0 || (d_segs && d_lr)) && (!rowloss || (d_cost && nrow > 0)), "tn_sgd_update_net: bad arguments");
            return upd_cost(ctx, static_cast<const tn_sgd_seg*>(d_segs), nseg < 0 ? 0 : nseg, d_lr, gscale, d_step, rowloss, nrow,
                            cost_scale, d_cost);
        case TN_UPD_DELAYED:
            REQUIRE(rowloss == nullptr, "tn_sgd_update_net (delayed): no cost rider in this mode");
            return upd_delayed(ctx, static_cast<const tn_sgd_seg*>(d_segs), nseg, d_lr, gscale, d_step, flags);
        case TN_UPD_PIPE:
            return upd_pipe(ctx, static_cast<const tn_pipe_seg*>(d_segs), nseg, d_lr, d_step, step_inc, flags & 1, rowloss, nrow,
                            cost_scale, d_cost);
        default:
            return fail(ctx, TN_E_ARG, "tn_sgd_update_net: mode %d", mode);
    }
}

// ================================== elastic input stage (inlayers.py:63-144) ==================================
#define EL_HDR 8
size_t tn_elastic_draws_count(int h, int w) { return (size_t)EL_HDR + 2 * (size_t)h * w; }
static void elastic_draw4(int q, uint32_t st, uint64_t seed, float v[4]) {
    const u32x4 r = philox4x32((uint32_t)q, 0u, st, STREAM_ELASTIC, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t wd[4] = {r.x, r.y, r.z, r.w};
    if (4 * q < EL_HDR) {
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            const float u = u01(wd[e]);
            v[e] = (i == 2 || i == 3) ? .25f + .5f * u : -1.f + 2.f * u;
        }
        return;
    }
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((wd[2 * h] >> 8) + 1) * (1.0f / 16777216.0f);
        const float a = 6.28318530717958647692f * u01(wd[2 * h + 1]);
        const float rad = std::sqrt(-2.f * std::log(u1));
        v[2 * h] = rad * std::cos(a);
        v[2 * h + 1] = rad * std::sin(a);
    }
}
int tn_elastic_draws(tn_ctx*, float* draws, int h, int w, uint64_t seed, uint32_t step, const uint32_t* d_step) {
    const int total = (int)tn_elastic_draws_count(h, w);
    const uint32_t st = step + (d_step ? *d_step : 0u);
#pragma omp parallel for
    for (int q = 0; q < (total + 3) / 4; ++q) {
        float v[4];
        elastic_draw4(q, st, seed, v);
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < total) draws[4 * q + e] = v[e];
    }
    return TN_OK;
}
int tn_elastic_field(tn_ctx* ctx, const float* draws, int h, int w, double translation, double zoom, double magnitude,
                     int sigma, double angle, int nearest, int32_t* map_idx, float* map_fy, float* map_fx, double* target) {
    REQUIRE(h > 0 && w > 0 && zoom > 0 && sigma >= 0 && map_idx != nullptr, "tn_elastic_field: bad arguments");
    REQUIRE(nearest || (map_fy && map_fx), "tn_elastic_field: bilinear needs map_fy/map_fx");
    const int ks = 2 * sigma + 1;
    std::vector<float> filt((size_t)ks * ks, 0.f);
    if (magnitude != 0.0) {
        const double var = (double)sigma * sigma;
        const float norm = (float)(2.0 * 3.14159265358979323846 * var);
        for (int t = 0; t < ks * ks; ++t) {
            const int i = t % ks - sigma, j = t / ks - sigma;
            filt[t] = (float)std::exp(-.5 * (i * i + j * j) / var) / norm;
        }
    }
    double zy = 1.0, zx = 1.0, cs = 1.0, sn = 0.0;
    if (zoom != 1.0) { zy = std::exp(std::log(zoom) * (double)draws[4]); zx = std::exp(std::log(zoom) * (double)draws[5]); }
    if (angle != 0.0) {
        const double theta = (angle * 3.14159265358979323846 / 180.0) * (double)draws[6];
        cs = std::cos(theta); sn = std::sin(theta);
    }
    const float* n0 = draws + EL_HDR;
    const float* n1 = n0 + h * w;
    const float mag = (float)magnitude;
#pragma omp parallel for
    for (int p = 0; p < h * w; ++p) {
        const int y = p / w, x = p - y * w;
        double ty = y, tx = x;
        if (translation != 0.0) {
            ty += (double)((float)translation * draws[0]);
            tx += (double)((float)translation * draws[1]);
        }
        if (magnitude != 0.0) {
            // float32 products, float64 accumulation, rounded to float32 once (as the GPU kernel)
            double s0 = 0.0, s1 = 0.0;
            for (int u = 0; u < ks; ++u) {
                const int yy = y + u - sigma;
                if (yy < 0 || yy >= h) continue;
                for (int v = 0; v < ks; ++v) {
                    const int xx = x + v - sigma;
                    if (xx < 0 || xx >= w) continue;
                    const float fw = filt[u * ks + v];
                    s0 += (double)fw * (double)(mag * n0[yy * w + xx]);
                    s1 += (double)fw * (double)(mag * n1[yy * w + xx]);
                }
            }
            ty += (double)(float)s0;
            tx += (double)(float)s1;
        }
        if (zoom != 1.0 || angle != 0.0) {
            const double oy = (double)draws[2] * h, ox = (double)draws[3] * w;
            ty -= oy; tx -= ox;
            if (zoom != 1.0) { ty *= zy; tx *= zx; }
            if (angle != 0.0) {               // tensordot(R, target, axes=(0,0)), R=[[c,-s],[s,c]] -> R^T applied
                const double ry = cs * ty + sn * tx, rx = -sn * ty + cs * tx;
                ty = ry; tx = rx;
            }
            ty += oy; tx += ox;
        }
        if (target) { target[p] = ty; target[h * w + p] = tx; }
        const double cy = std::fmin(std::fmax(ty, 0.0), (double)h - 1 - .001);
        const double cx = std::fmin(std::fmax(tx, 0.0), (double)w - 1 - .001);
        if (nearest) {
            map_idx[p] = (int)std::rint(cy) * w + (int)std::rint(cx);
        } else {
            const int top = (int)cy, left = (int)cx;
            map_idx[p] = top * w + left;
            map_fy[p] = (float)(cy - top);
            map_fx[p] = (float)(cx - left);
        }
    }
    return TN_OK;
}
int tn_elastic_field_gen(tn_ctx* ctx, float* draws_out, uint64_t seed, uint32_t step, const uint32_t* d_step, int h, int w,
                         double translation, double zoom, double magnitude, int sigma, double angle, int nearest,
                         int32_t* map_idx, float* map_fy, float* map_fx, double* target) {
    std::vector<float> tmp;
    float* draws = draws_out;
    if (!draws) { tmp.resize(tn_elastic_draws_count(h, w)); draws = tmp.data(); }
    int rc = tn_elastic_draws(ctx, draws, h, w, seed, step, d_step);
    if (rc) return rc;
    return tn_elastic_field(ctx, draws, h, w, translation, zoom, magnitude, sigma, angle, nearest, map_idx, map_fy, map_fx, target);
}
// the rider runs at once here (it only depends on the step counter): nothing is ever pending
int tn_rider_elastic_field(tn_ctx* ctx, float* draws_out, uint64_t seed, uint32_t step, const uint32_t* d_step, int h,
                           int w, double translation, double zoom, double magnitude, int sigma, double angle, int nearest,
                           int32_t* map_idx, float* map_fy, float* map_fx, double* target) {
    return tn_elastic_field_gen(ctx, draws_out, seed, step, d_step, h, w, translation, zoom, magnitude, sigma, angle,
                                nearest, map_idx, map_fy, map_fx, target);
}
int tn_rider_pending(tn_ctx*) { return 0; }
int tn_rider_cancel(tn_ctx*) { return TN_OK; }
int tn_step_tail(tn_ctx* ctx, const tn_sgd_seg* d_segs, int nseg, size_t max_n, const float* d_lr, float gscale,
                 const float* rowloss, int nrow, float cost_scale, float* d_cost, float* draws_out, uint64_t seed,
                 const uint32_t* d_step, int h, int w, double translation, double zoom, double magnitude, int sigma,
                 double angle, int nearest, int32_t* map_idx, float* map_fy, float* map_fx, double* target) {
    int rc = tn_sgd_update_net(ctx, TN_UPD_PLAIN, d_segs, nullptr, nseg, max_n, d_lr, gscale, nullptr, 0, 0, rowloss, nrow, cost_scale, d_cost);
    if (rc) return rc;
    return tn_elastic_field_gen(ctx, draws_out, seed, 0, d_step, h, w, translation, zoom, magnitude, sigma, angle, nearest,
                                map_idx, map_fy, map_fx, target);
}
int tn_elastic_apply(tn_ctx*, const float* x, int64_t x_row0, const int64_t* d_row0, float* out, int N, int C, int h, int w,
                     int invert, int nearest, const int32_t* map_idx, const float* map_fy, const float* map_fx, float pflip,
                     const uint8_t* flipmask, uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    const int hw = h * w;
    const int64_t row_off = x_row0 + (d_row0 ? *d_row0 : 0);
    const uint32_t st = step + (d_step ? *d_step : 0u);
#pragma omp parallel for
    for (long long img = 0; img < (long long)N * C; ++img) {
        const float* xi = x + ((size_t)row_off * C + img) * hw;
        for (int p = 0; p < hw; ++p) {
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
                if (invert) { a = 1.f - a; b = 1.f - b; c = 1.f - c; d = 1.f - d; }
                v = a * (1.f - fy) * (1.f - fx) + b * (1.f - fy) * fx + c * fy * (1.f - fx) + d * fy * fx;   // inlayers.py:134-137
            }
            const size_t t = (size_t)img * hw + p;
            if (flipmask) {
                if (flipmask[t]) v = 1.f - v;
            } else if (pflip > 0.f) {
                const uint64_t e = (uint64_t)row_global0 * C * hw + (uint64_t)t;
                if (u01(philox_word(e, st, STREAM_FLIP, seed)) < pflip) v = 1.f - v;
            }
            out[t] = v;
        }
    }
    return TN_OK;
}

// ---- backward of the resampling (mid-net ElasticLayer) ----
int tn_elastic_apply_bwd(tn_ctx* ctx, const float* g, float* dx, int N, int C, int h, int w, int invert, int nearest,
                         const int32_t* map_idx, const float* map_fy, const float* map_fx, float pflip,
                         const uint8_t* flipmask, uint64_t seed, uint32_t step, const uint32_t* d_step,
                         int64_t row_global0, const float* prev_a, int prev_act, float prm) {
    REQUIRE(g && dx && N > 0 && C > 0 && h > 0 && w > 0, "tn_elastic_apply_bwd: bad arguments");
    const int hw = h * w;
    const uint32_t st = step + (d_step ? *d_step : 0u);
#pragma omp parallel for
    for (long long img = 0; img < (long long)N * C; ++img) {
        float* d = dx + (size_t)img * hw;
        std::memset(d, 0, sizeof(float) * hw);
        for (int p = 0; p < hw; ++p) {
            const size_t t = (size_t)img * hw + p;
            float gv = g[t];
            bool flip = false;
            if (flipmask) flip = flipmask[t] != 0;
            else if (pflip > 0.f) flip = u01(philox_word((uint64_t)row_global0 * C * hw + (uint64_t)t, st, STREAM_FLIP, seed)) < pflip;
            if (flip) gv = -gv;
            if (invert) gv = -gv;
            if (!map_idx) d[p] += gv;
            else if (nearest) d[map_idx[p]] += gv;
            else {
                const int i00 = map_idx[p];
                const float fy = map_fy[p], fx = map_fx[p];
                d[i00] += gv * (1.f - fy) * (1.f - fx); d[i00 + 1] += gv * (1.f - fy) * fx;
                d[i00 + w] += gv * fy * (1.f - fx); d[i00 + w + 1] += gv * fy * fx;
            }
        }
        if (prev_a && prev_act != TN_ACT_LINEAR)
            for (int i = 0; i < hw; ++i) d[i] *= act_grad_from_out(prev_a[(size_t)img * hw + i], prev_act, prm);
    }
    return TN_OK;
}

// ---- ColorLayer (color.py:9-52) ----
int tn_color_factors(tn_ctx* ctx, float* fac, int N, int C, double balance, double gamma, const float* draws,
                     uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    REQUIRE(fac && N > 0 && C > 0 && balance > 0 && gamma > 0, "tn_color_factors: bad arguments");
    const int NC = N * C;
    const double lnb = std::log(balance), lng = std::log(gamma);
    for (int i = 0; i < NC; ++i) {
        float u[3];
        if (draws) { u[0] = draws[i]; u[1] = draws[NC + i]; u[2] = draws[2 * NC + i]; }
        else {
            const uint64_t e = (uint64_t)row_global0 * C + (uint64_t)i;
            const u32x4 r = philox4x32((uint32_t)e, (uint32_t)(e >> 32), step + (d_step ? *d_step : 0u), 5u, (uint32_t)seed, (uint32_t)(seed >> 32));
            u[0] = -1.f + 2.f * u01(r.x); u[1] = -1.f + 2.f * u01(r.y); u[2] = -1.f + 2.f * u01(r.z);
        }
        fac[3 * i] = (float)std::exp(lnb * (double)u[0]);
        fac[3 * i + 1] = (float)std::exp(lng * (double)u[1]);
        fac[3 * i + 2] = (float)std::exp(lng * (double)u[2]);
    }
    return TN_OK;
}
static void color_pass(bool bwd, const float* x, const float* fac, const float* g, float* out, long long total, int hw,
                       float maxval, const float* prev_a, int prev_act, float prm) {
#pragma omp parallel for
    for (long long t = 0; t < total; ++t) {
        const long long img = t / hw;
        const float b = fac[3 * img], g1 = fac[3 * img + 1], g2 = fac[3 * img + 2];
        const float o1 = x[t] / maxval * b, o2 = std::fmin(std::fmax(o1, 0.f), 1.f), o3 = std::pow(o2, g1);
        if (!bwd) { out[t] = (1.f - std::pow(1.f - o3, g2)) * maxval; continue; }
        float d = g2 * std::pow(1.f - o3, g2 - 1.f) * g1 * std::pow(o2, g1 - 1.f) * b;
        if (!(o1 >= 0.f && o1 <= 1.f)) d = 0.f;
        float v = g[t] * d;
        if (prev_a && prev_act != TN_ACT_LINEAR) v *= act_grad_from_out(prev_a[t], prev_act, prm);
        out[t] = v;
    }
}
int tn_color_apply(tn_ctx* ctx, const float* x, int64_t x_row0, const float* fac, float* out, int N, int C, int hw, float maxval) {
    REQUIRE(x && fac && out && N > 0 && C > 0 && hw > 0 && maxval > 0, "tn_color_apply: bad arguments");
    color_pass(false, x + (size_t)x_row0 * C * hw, fac, nullptr, out, (long long)N * C * hw, hw, maxval, nullptr, 0, 0.f);
    return TN_OK;
}
int tn_color_apply_bwd(tn_ctx* ctx, const float* x, int64_t x_row0, const float* fac, const float* g, float* dx, int N, int C,
                       int hw, float maxval, const float* prev_a, int prev_act, float prm) {
    REQUIRE(x && fac && g && dx && N > 0 && C > 0 && hw > 0 && maxval > 0, "tn_color_apply_bwd: bad arguments");
    color_pass(true, x + (size_t)x_row0 * C * hw, fac, g, dx, (long long)N * C * hw, hw, maxval, prev_a, prev_act, prm);
    return TN_OK;
}

// ---- aux-input layers (auxiliary.py:14-160): input mix and column copies ----
int tn_aux_mix(tn_ctx* ctx, const float* aux, int64_t row0, float* out, int B, int d, float boost, int train,
               const float* u_inj, uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0) {
    REQUIRE(aux && out && B > 0 && d > 0, "tn_aux_mix: bad arguments");
    const float* a = aux + (size_t)row0 * 2 * d;
    const uint32_t st = step + (d_step ? *d_step : 0u);
    for (int n = 0; n < B; ++n) {
        float u = 0.f;
        if (train) {
            if (u_inj) u = u_inj[n];
            else {
                const uint64_t e = (uint64_t)row_global0 + (uint64_t)n;
                u = u01(philox4x32((uint32_t)e, (uint32_t)(e >> 32), st, 6u, (uint32_t)seed, (uint32_t)(seed >> 32)).x);
            }
        }
        for (int j = 0; j < d; ++j) {
            const float a0 = a[(size_t)n * 2 * d + j], a1 = a[(size_t)n * 2 * d + d + j];
            out[(size_t)n * d + j] = (train ? a0 * u + a1 * (1.f - u) : (a0 + a1) / 2.f) * boost;
        }
    }
    return TN_OK;
}
int tn_copy_cols(tn_ctx* ctx, const float* src, int ld_src, int col_src, float* dst, int ld_dst, int col_dst, int ncols,
                 int B, const float* prev_a, int prev_act, float prm) {
    REQUIRE(src && dst && B > 0 && ncols > 0 && col_src + ncols <= ld_src && col_dst + ncols <= ld_dst, "tn_copy_cols: bad arguments");
    for (int n = 0; n < B; ++n)
        for (int j = 0; j < ncols; ++j) {
            float v = src[(size_t)n * ld_src + col_src + j];
            const size_t o = (size_t)n * ld_dst + col_dst + j;
            if (prev_a && prev_act != TN_ACT_LINEAR) v *= act_grad_from_out(prev_a[o], prev_act, prm);
            dst[o] = v;
        }
    return TN_OK;
}

// ---- extras/deformer.py:7-18: per-image deformation, float64 like scipy ----
int tn_deformer_transform(tn_ctx* ctx, const float* imgs, float* out, int N, int h, int w, double scale, double sigma,
                          double cval, const float* noise, uint64_t seed, int64_t img_global0) {
    REQUIRE(N >= 0 && h > 0 && w > 0 && sigma > 0, "tn_deformer_transform: bad arguments");
    const int hw = h * w, r = (int)(2.0 * sigma + 0.5);
    std::vector<double> kern(2 * r + 1);
    double ksum = 0.0;
    for (int t = 0; t <= 2 * r; ++t) { const double d = t - r; kern[t] = std::exp(-0.5 / (sigma * sigma) * d * d); ksum += kern[t]; }
    for (auto& k : kern) k /= ksum;
#pragma omp parallel
    {
        std::vector<double> tr(2 * hw), tmp(2 * hw);
#pragma omp for
        for (int img = 0; img < N; ++img) {
            for (int t = 0; t < 2 * hw; ++t) {
                const int a = t / hw, p = t - a * hw;
                double u;
                if (noise) {
                    u = (double)noise[(size_t)img * 2 * hw + t];
                } else {
                    const uint64_t e = (uint64_t)(img_global0 + img) * 2 * hw + t;
                    const u32x4 q = philox4x32((uint32_t)e, (uint32_t)(e >> 32), 0u, STREAM_DEFORMER, (uint32_t)seed, (uint32_t)(seed >> 32));
                    u = -1.0 + 2.0 * (double)u01(q.x);
                }
                tr[t] = (double)(a == 0 ? p / w : p % w) + scale * u;
            }
            for (int t = 0; t < 2 * hw; ++t) {              // axis 0, edge replicated (mode='nearest')
                const int a = t / hw, p = t - a * hw, y = p / w, x = p - y * w;
                double s = 0.0;
                for (int k = -r; k <= r; ++k) s += kern[k + r] * tr[a * hw + std::min(std::max(y + k, 0), h - 1) * w + x];
                tmp[t] = s;
            }
            for (int t = 0; t < 2 * hw; ++t) {
                const int a = t / hw, p = t - a * hw, y = p / w, x = p - y * w;
                double s = 0.0;
                for (int k = -r; k <= r; ++k) s += kern[k + r] * tmp[a * hw + y * w + std::min(std::max(x + k, 0), w - 1)];
                tr[t] = s;
            }
            const float* im = imgs + (size_t)img * hw;
            auto tap = [&](int yy, int xx) -> double { return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (double)im[yy * w + xx] : cval; };
            for (int p = 0; p < hw; ++p) {
                const double cy = tr[p], cx = tr[hw + p];
                double v;
                if (cy < 0.0 || cy > h - 1 || cx < 0.0 || cx > w - 1) {
                    v = cval;
                } else {
                    const int y0 = (int)std::floor(cy), x0 = (int)std::floor(cx);
                    const double fy = cy - y0, fx = cx - x0;
                    v = tap(y0, x0) * (1 - fy) * (1 - fx) + tap(y0, x0 + 1) * (1 - fy) * fx + tap(y0 + 1, x0) * fy * (1 - fx) +
                        tap(y0 + 1, x0 + 1) * fy * fx;
                }
                out[(size_t)img * hw + p] = (float)v;
            }
        }
    }
    return TN_OK;
}

int tn_gather_rows(tn_ctx*, const void* src, const int32_t* idx, void* dst, int nrows, size_t row_bytes) {
#pragma omp parallel for
    for (int r = 0; r < nrows; ++r)
        std::memcpy(static_cast<char*>(dst) + (size_t)r * row_bytes, static_cast<const char*>(src) + (size_t)idx[r] * row_bytes, row_bytes);
    return TN_OK;
}

// ---- data-parallel exchange: ranks of a CPU job reduce HOST buffers in the Python layer
// (theanet_amd/comm.py, socket rendezvous); one rank needs nothing ----
int tn_comm_unique_id(tn_ctx*, void* id128) { std::memset(id128, 0, TN_UNIQUE_ID_BYTES); return TN_OK; }
int tn_comm_init(tn_ctx* ctx, const void*, int rank, int world) { ctx->rank = rank; ctx->world = world; return TN_OK; }
int tn_comm_destroy(tn_ctx*) { return TN_OK; }
int tn_allreduce_sum(tn_ctx* ctx, float*, size_t) {
    REQUIRE(ctx->world == 1, "tn_allreduce_sum: multi-rank reductions of the CPU backend run in theanet_amd.comm (host buffers)");
    return TN_OK;
}
int tn_allreduce_sum_async(tn_ctx* ctx, float*, size_t, void*) {     // (host code is synchronous: nothing to order)
    REQUIRE(ctx->world == 1, "tn_allreduce_sum_async: multi-rank reductions of the CPU backend run in theanet_amd.comm (host buffers)");
    return TN_OK;
}
int tn_allreduce_sum_rsag(tn_ctx* ctx, float*, size_t, int, void*) {     // one rank: the sum is the buffer itself
    REQUIRE(ctx->world == 1, "tn_allreduce_sum_rsag: multi-rank reductions of the CPU backend run in theanet_amd.comm (host buffers)");
    return TN_OK;
}
int tn_allreduce_max(tn_ctx* ctx, float*, size_t) {
    REQUIRE(ctx->world == 1, "tn_allreduce_max: multi-rank reductions of the CPU backend run in theanet_amd.comm (host buffers)");
    return TN_OK;
}
int tn_axpby(tn_ctx*, float* y, const float* x, size_t n, float a, float b) {
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; ++i) y[i] = a * x[i] + b * y[i];
    return TN_OK;
}

int tn_maxnorm_multi(tn_ctx* ctx, const tn_mn_seg* h_segs, int nseg) {
    if (nseg < 0 || nseg > 32 || (nseg && !h_segs)) return fail(ctx, TN_E_ARG, "tn_maxnorm_multi: bad arguments");
    for (int i = 0; i < nseg; ++i) {
        if (h_segs[i].maxnorm == 0.f || !h_segs[i].p) continue;
        int rc = tn_maxnorm(ctx, h_segs[i].p, h_segs[i].ndim, h_segs[i].d0, h_segs[i].rest, h_segs[i].maxnorm);
        if (rc) return rc;
    }
    return TN_OK;
}
// the update followed by the projection (layer.py:82-103 as one op; this library keeps them two passes)
int tn_sgd_update_net_maxnorm(tn_ctx* ctx, int mode, const void* d_segs, const void* h_segs, int nseg, size_t max_n,
                              const float* d_lr, float gscale, uint32_t* d_step, uint32_t step_inc, int flags,
                              const float* rowloss, int nrow, float cost_scale, float* d_cost, const tn_mn_seg* h_mn, int nmn) {
    int rc = tn_sgd_update_net(ctx, mode, d_segs, h_segs, nseg, max_n, d_lr, gscale, d_step, step_inc, flags, rowloss, nrow,
                               cost_scale, d_cost);
    return rc ? rc : tn_maxnorm_multi(ctx, h_mn, nmn);
}

}  // extern "C"

// the coarse step entry points (tn_net_plan_*, tn_net_step): the same host code as the HIP library
#define TN_REQUIRE(cond, ...) REQUIRE(cond, __VA_ARGS__)
#define tn_fail fail
#include "../csrc/net_plan.h"
