// Context lifecycle, memory, scalar stores, graph capture, events, small utilities.
#include "common.h"

#include <cmath>

char g_tn_err[512] = {0};

extern "C" {

int tn_version(void) { return 100; }

int tn_device_count(int* count) {
    tn_ctx* ctx = nullptr;
    TN_HIP(hipGetDeviceCount(count));
    return TN_OK;
}

int tn_ctx_create(int device, tn_ctx** out) {
    tn_ctx* ctx = nullptr;
    if (!out) return tn_fail(nullptr, TN_E_ARG, "tn_ctx_create: out is NULL");
    int n = 0;
    TN_HIP(hipGetDeviceCount(&n));
    if (n <= 0) return tn_fail(nullptr, TN_E_HIP, "tn_ctx_create: no HIP device visible");
    if (device < 0 || device >= n)
        return tn_fail(nullptr, TN_E_ARG, "tn_ctx_create: device %d out of range [0,%d)", device, n);
    TN_HIP(hipSetDevice(device));
    tn_ctx* c = new tn_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->streams[0], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams[1], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sync_ev[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sync_ev[1], hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->copy_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->comm_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        delete c;
        return tn_fail(nullptr, TN_E_HIP, "hipStreamCreate -> %s", hipGetErrorString(e));
    }
    c->stream = c->streams[0];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    c->scratch_bytes = 1 << 20;
    e = hipMalloc((void**)&c->scratch, c->scratch_bytes);
    if (e != hipSuccess) {
        hipStreamDestroy(c->stream);
        delete c;
        return tn_fail(nullptr, TN_E_NOMEM, "hipMalloc scratch -> %s", hipGetErrorString(e));
    }
    *out = c;
    return TN_OK;
}

int tn_ctx_destroy(tn_ctx* ctx) {
    if (!ctx) return TN_OK;
    hipSetDevice(ctx->device);
    tn_comm_destroy(ctx);
    hipStreamSynchronize(ctx->streams[0]);
    hipStreamSynchronize(ctx->streams[1]);
    if (ctx->scratch) hipFree(ctx->scratch);
    for (int i = 0; i < 2; ++i) {
        if (ctx->scratch_slot[i]) hipFree(ctx->scratch_slot[i]);
        if (ctx->tmp[i]) hipFree(ctx->tmp[i]);
    }
    hipEventDestroy(ctx->sync_ev[0]);
    hipEventDestroy(ctx->sync_ev[1]);
    hipStreamDestroy(ctx->streams[0]);
    hipStreamDestroy(ctx->streams[1]);
    if (ctx->copy_stream) {
        hipStreamSynchronize(ctx->copy_stream);
        hipStreamDestroy(ctx->copy_stream);
    }
    if (ctx->copy_ev) hipEventDestroy(ctx->copy_ev);
    if (ctx->comm_stream) {
        hipStreamSynchronize(ctx->comm_stream);
        hipStreamDestroy(ctx->comm_stream);
    }
    if (ctx->comm_ev) hipEventDestroy(ctx->comm_ev);
    delete ctx;
    return TN_OK;
}

int tn_set_matmul_dtype(tn_ctx* ctx, int dtype, float grad_scale) {
    TN_REQUIRE(dtype == 0 || dtype == 1, "tn_set_matmul_dtype: dtype %d (0 fp32, 1 fp16 operands)", dtype);
    int ex = 0;
    const float m = frexpf(grad_scale, &ex);
    TN_REQUIRE(grad_scale > 0.f && m == 0.5f, "tn_set_matmul_dtype: grad_scale %g is not a power of two", grad_scale);
    ctx->mm_f16 = dtype;
    ctx->grad_scale = dtype ? grad_scale : 1.f;
    return TN_OK;
}

int tn_get_matmul_dtype(tn_ctx* ctx) { return ctx->mm_f16; }

int tn_set_fc_matmul(tn_ctx* ctx, int mode) {
    TN_REQUIRE(mode == 0 || mode == 1, "tn_set_fc_matmul: mode %d (0 exact fp32 MFMA, 1 bf16 triplets)", mode);
    ctx->fc_b3 = mode;
    return TN_OK;
}

const char* tn_last_error(tn_ctx* ctx) { return ctx ? ctx->err : g_tn_err; }

int tn_sync(tn_ctx* ctx) {
    TN_HIP(hipStreamSynchronize(ctx->streams[1]));
    TN_HIP(hipStreamSynchronize(ctx->streams[0]));
    if (ctx->comm_stream) TN_HIP(hipStreamSynchronize(ctx->comm_stream));
    return TN_OK;
}

int tn_stream_select(tn_ctx* ctx, int idx) {
    TN_REQUIRE(idx == 0 || idx == 1, "tn_stream_select: idx %d", idx);
    const int cur = ctx->stream == ctx->streams[1] ? 1 : 0;
    if (idx != cur) {       // every stream has its own scratch (slabs of two steps in flight must not alias)
        ctx->scratch_slot[cur] = ctx->scratch;
        ctx->scratch_slot_bytes[cur] = ctx->scratch_bytes;
        ctx->scratch = ctx->scratch_slot[idx];
        ctx->scratch_bytes = ctx->scratch_slot_bytes[idx];
        ctx->scratch_slot[idx] = nullptr;
        ctx->scratch_slot_bytes[idx] = 0;
        // the deferral window travels with its stream
        ctx->defer_slot[cur] = ctx->defer; ctx->scratch_off_slot[cur] = ctx->scratch_off; ctx->npend_slot[cur] = ctx->npend;
        for (int i = 0; i < ctx->npend; ++i) ctx->pend_slot[cur][i] = ctx->pend[i];
        ctx->defer = ctx->defer_slot[idx]; ctx->scratch_off = ctx->scratch_off_slot[idx]; ctx->npend = ctx->npend_slot[idx];
        for (int i = 0; i < ctx->npend; ++i) ctx->pend[i] = ctx->pend_slot[idx][i];
        ctx->defer_slot[idx] = false; ctx->scratch_off_slot[idx] = 0; ctx->npend_slot[idx] = 0;
    }
    if (idx == 1) ctx->heavy_since_side = 0;
    ctx->stream = ctx->streams[idx];
    return TN_OK;
}

int tn_stream_wait(tn_ctx* ctx, int waiter, int signaler) {
    TN_REQUIRE((waiter == 0 || waiter == 1) && (signaler == 0 || signaler == 1) && waiter != signaler,
               "tn_stream_wait: bad stream ids");
    TN_HIP(hipEventRecord(ctx->sync_ev[signaler], ctx->streams[signaler]));
    TN_HIP(hipStreamWaitEvent(ctx->streams[waiter], ctx->sync_ev[signaler], 0));
    return TN_OK;
}

int tn_device_info(tn_ctx* ctx, char* name, int name_len, int* cus, size_t* hbm_bytes) {
    hipDeviceProp_t prop;
    TN_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return TN_OK;
}

int tn_alloc(tn_ctx* ctx, size_t bytes, void** dptr) {
    TN_REQUIRE(dptr != nullptr, "tn_alloc: dptr is NULL");
    TN_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess)
        return tn_fail(ctx, TN_E_NOMEM, "hipMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
    return TN_OK;
}

int tn_free(tn_ctx* ctx, void* dptr) {
    if (!dptr) return TN_OK;
    TN_HIP(hipStreamSynchronize(ctx->stream));
    TN_HIP(hipFree(dptr));
    return TN_OK;
}

int tn_h2d(tn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return TN_OK;
    TN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    TN_HIP(hipStreamSynchronize(ctx->stream));
    return TN_OK;
}

int tn_d2h(tn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return TN_OK;
    TN_HIP(hipStreamSynchronize(ctx->streams[1]));
    TN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->streams[0]));
    TN_HIP(hipStreamSynchronize(ctx->streams[0]));
    return TN_OK;
}

int tn_host_alloc(tn_ctx* ctx, size_t bytes, void** out) {
    TN_REQUIRE(out != nullptr && bytes > 0, "tn_host_alloc: bad arguments");
    TN_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return TN_OK;
}

int tn_host_free(tn_ctx* ctx, void* p) {
    if (p) TN_HIP(hipHostFree(p));
    return TN_OK;
}

int tn_d2h_early(tn_ctx* ctx, void* host_dst, const void* src, size_t bytes) {
    if (!bytes) return TN_OK;
    TN_HIP(hipEventRecord(ctx->copy_ev, ctx->stream));
    TN_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_ev, 0));
    TN_HIP(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    return TN_OK;
}

int tn_d2h_early_ev(tn_ctx* ctx, void* host_dst, const void* src, size_t bytes, void* done_event) {
    int rc = tn_d2h_early(ctx, host_dst, src, bytes);
    if (rc) return rc;
    if (done_event) TN_HIP(hipEventRecord((hipEvent_t)done_event, ctx->copy_stream));
    return TN_OK;
}

int tn_copy_sync(tn_ctx* ctx) {
    TN_HIP(hipStreamSynchronize(ctx->copy_stream));
    return TN_OK;
}

int tn_d2d(tn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return TN_OK;
    TN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return TN_OK;
}

int tn_memset(tn_ctx* ctx, void* dst, int byte_value, size_t bytes) {
    if (!bytes) return TN_OK;
    TN_HIP(hipMemsetAsync(dst, byte_value, bytes, ctx->stream));
    return TN_OK;
}

}  // extern "C"

template <typename T>
__global__ void set_scalar_kernel(T* dst, T v) {
    *dst = v;
}
__global__ void add_u32_kernel(uint32_t* dst, uint32_t inc) { *dst += inc; }

extern "C" {

int tn_set_u32(tn_ctx* ctx, uint32_t* d, uint32_t v) {
    set_scalar_kernel<uint32_t><<<1, 1, 0, ctx->stream>>>(d, v);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
int tn_set_i64(tn_ctx* ctx, int64_t* d, int64_t v) {
    set_scalar_kernel<int64_t><<<1, 1, 0, ctx->stream>>>(d, v);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
int tn_set_f32(tn_ctx* ctx, float* d, float v) {
    set_scalar_kernel<float><<<1, 1, 0, ctx->stream>>>(d, v);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
int tn_add_u32(tn_ctx* ctx, uint32_t* d, uint32_t inc) {
    add_u32_kernel<<<1, 1, 0, ctx->stream>>>(d, inc);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// ---- graph capture -------------------------------------------------------------
int tn_graph_begin(tn_ctx* ctx) {
    ctx->stream = ctx->streams[0];
    TN_HIP(hipStreamBeginCapture(ctx->streams[0], hipStreamCaptureModeThreadLocal));
    return TN_OK;
}

int tn_graph_end(tn_ctx* ctx, void** graph_exec) {
    hipGraph_t g = nullptr;
    ctx->stream = ctx->streams[0];
    TN_HIP(hipStreamEndCapture(ctx->streams[0], &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess)
        return tn_fail(ctx, TN_E_HIP, "hipGraphInstantiate -> %s", hipGetErrorString(e));
    *graph_exec = (void*)ge;
    return TN_OK;
}

int tn_graph_launch(tn_ctx* ctx, void* graph_exec) {
    TN_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, ctx->streams[0]));
    return TN_OK;
}

int tn_graph_destroy(tn_ctx* ctx, void* graph_exec) {
    if (graph_exec) TN_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return TN_OK;
}

// ---- events ----------------------------------------------------------------------
int tn_event_create(tn_ctx* ctx, void** ev) {
    hipEvent_t e;
    TN_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return TN_OK;
}
int tn_event_record(tn_ctx* ctx, void* ev) {
    TN_HIP(hipEventRecord((hipEvent_t)ev, ctx->stream));
    return TN_OK;
}
int tn_event_wait(tn_ctx* ctx, void* ev) {
    TN_HIP(hipStreamWaitEvent(ctx->stream, (hipEvent_t)ev, 0));
    return TN_OK;
}
int tn_event_elapsed_ms(tn_ctx* ctx, void* a, void* b, float* ms) {
    TN_HIP(hipEventSynchronize((hipEvent_t)b));
    TN_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return TN_OK;
}
int tn_event_destroy(tn_ctx* ctx, void* ev) {
    if (ev) TN_HIP(hipEventDestroy((hipEvent_t)ev));
    return TN_OK;
}
int tn_event_sync(tn_ctx* ctx, void* ev) {
    TN_HIP(hipEventSynchronize((hipEvent_t)ev));
    return TN_OK;
}
int tn_event_query(tn_ctx* ctx, void* ev, int* done) {
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e != hipSuccess && e != hipErrorNotReady) TN_HIP(e);
    *done = e == hipSuccess ? 1 : 0;
    return TN_OK;
}

}  // extern "C"

// ---- reductions / axpby / gather -------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Deterministic: single block, fixed traversal order.  n is small-ish (<= a few M).
template <int MODE>  // 0: sum v ; 1: L1*|p| + L2*p^2
__global__ __launch_bounds__(1024) void reduce_kernel(const float* __restrict__ v, size_t n,
                                                      float s0, float s1, float* out, int accumulate) {
    __shared__ float part[16];
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        float x = v[i];
        acc += (MODE == 0) ? x : (s0 * fabsf(x) + s1 * x * x);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += part[i];
        if (MODE == 0) t *= s0;
        out[0] = accumulate ? out[0] + t : t;
    }
}

__global__ void error_stats_kernel(const int32_t* pred, const int32_t* y, const float* rowp, int B,
                                   float* out2) {
    __shared__ float pe[16], pp[16];
    float e = 0.f, p = 0.f;
    for (int i = threadIdx.x; i < B; i += 1024) {
        e += (pred[i] != y[i]) ? 1.f : 0.f;
        p += rowp[i];
    }
    e = wave_sum(e);
    p = wave_sum(p);
    if ((threadIdx.x & 63) == 0) {
        pe[threadIdx.x >> 6] = e;
        pp[threadIdx.x >> 6] = p;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float te = 0.f, tp = 0.f;
        for (int i = 0; i < 16; ++i) {
            te += pe[i];
            tp += pp[i];
        }
        out2[0] = te / B;
        out2[1] = tp / B;
    }
}

__global__ void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n, float a,
                             float b) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}

__global__ void gather_rows_kernel(const uint32_t* __restrict__ src, const int32_t* __restrict__ idx,
                                   uint32_t* __restrict__ dst, int nrows, size_t row_words) {
    int r = blockIdx.y;
    const uint32_t* s = src + (size_t)idx[r] * row_words;
    uint32_t* d = dst + (size_t)r * row_words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_words;
         i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}

extern "C" {

int tn_reduce_sum(tn_ctx* ctx, const float* v, size_t n, float scale, float* out, int accumulate) {
    reduce_kernel<0><<<1, 1024, 0, ctx->stream>>>(v, n, scale, 0.f, out, accumulate);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_wtcost(tn_ctx* ctx, const float* p, size_t n, float L1, float L2, float* out, int accumulate) {
    reduce_kernel<1><<<1, 1024, 0, ctx->stream>>>(p, n, L1, L2, out, accumulate);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_error_stats(tn_ctx* ctx, const int32_t* pred, const int32_t* y, int64_t y_row0,
                   const float* rowp, int B, float* out2) {
    error_stats_kernel<<<1, 1024, 0, ctx->stream>>>(pred, y + y_row0, rowp, B, out2);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_axpby(tn_ctx* ctx, float* y, const float* x, size_t n, float a, float b) {
    if (!n) return TN_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    axpby_kernel<<<blocks, 256, 0, ctx->stream>>>(y, x, n, a, b);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_gather_rows(tn_ctx* ctx, const void* src, const int32_t* d_index, void* dst, int nrows,
                   size_t row_bytes) {
    TN_REQUIRE(row_bytes % 4 == 0, "tn_gather_rows: row_bytes must be a multiple of 4");
    if (!nrows) return TN_OK;
    size_t words = row_bytes / 4;
    int bx = (int)((words + 255) / 256);
    if (bx > 64) bx = 64;
    gather_rows_kernel<<<dim3(bx, nrows), 256, 0, ctx->stream>>>((const uint32_t*)src, d_index,
                                                               (uint32_t*)dst, nrows, words);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"

#include "net_plan.h"
