// Finishing reductions of the weight-gradient kernels, and the scratch they reduce from.
//
// Every wgrad kernel of the library leaves S partial slabs (one per block / K-split) that are
// summed in a fixed order: out[i] = sum_{s<S} src[s*stride + i].  Launched one by one these
// sums cost a kernel boundary each (~4-5 us for a few KB of work: four per training step).  With
// tn_defer_reductions(ctx, 1) the ops only RECORD their reduction and keep their slabs (the
// context scratch turns into a bump allocator); tn_defer_reductions(ctx, 0) runs all of them as
// ONE launch -- blockIdx.y selects the record.  Without deferral every op flushes its own records
// at once, so the C-ABI ops stay self-contained.
#include "common.h"

struct RedBatch {
    int nrec;
    tn_red_rec r[TN_RED_MAX];
};

// wide (S <= 32): thread = 4 consecutive outputs (16-byte accesses) or 1; loop over the slabs.
// tall (S  > 32): block = 16 outputs x 16 slab lanes, lane l sums slabs l, l+16, ...; the 16 lane
//                 sums are added in lane order.
// flip = f*f > 0: conv slabs are in correlation layout, out[kc*ff + uv] = sum src[kc*ff + ff-1-uv].
__global__ __launch_bounds__(256) void slab_sum_multi_kernel(RedBatch b, uint32_t* inc) {
    if (inc && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *inc += 1;     // rider: step counter
    const tn_red_rec rec = b.r[blockIdx.y];
    const uint32_t n = rec.n, S = rec.S, stride = rec.stride;
    const float* __restrict__ src = rec.src;
    float* __restrict__ out = rec.out;
    if (S <= 32) {
        const bool vec = rec.flip == 0 && (n & 3) == 0 && (stride & 3) == 0 &&
                         (((uintptr_t)src | (uintptr_t)out) & 15) == 0;
        if (vec) {
            const uint32_t i4 = (blockIdx.x * 256u + threadIdx.x) * 4u;
            if (i4 >= n) return;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (uint32_t z = 0; z < S; ++z) {
                const float4 v = *reinterpret_cast<const float4*>(src + (size_t)z * stride + i4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4*>(out + i4) = s;
        } else {
            const uint32_t i = blockIdx.x * 256u + threadIdx.x;
            if (i >= n) return;
            uint32_t j = i;
            if (rec.flip) {
                const uint32_t kc = i / rec.flip, uv = i - kc * rec.flip;
                j = kc * rec.flip + (rec.flip - 1 - uv);
            }
            float s = 0.f;
#pragma unroll 8
            for (uint32_t z = 0; z < S; ++z) s += src[(size_t)z * stride + j];
            out[i] = s;
        }
        return;
    }
    __shared__ float red[16][17];
    const uint32_t ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const uint32_t i = blockIdx.x * 16u + ol;
    if (blockIdx.x * 16u >= n) return;
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
        out[i] = t;
    }
}

static uint32_t red_blocks(const tn_red_rec& r) {
    if (r.S > 32) return (r.n + 15) / 16;
    const bool vec = r.flip == 0 && (r.n & 3) == 0 && (r.stride & 3) == 0 &&
                     (((uintptr_t)r.src | (uintptr_t)r.out) & 15) == 0;
    return vec ? (r.n / 4 + 255) / 256 : (r.n + 255) / 256;
}

__global__ void red_inc_kernel(uint32_t* inc) { *inc += 1; }

int tn_red_flush(tn_ctx* ctx) { return tn_red_flush_inc(ctx, nullptr); }

int tn_red_flush_inc(tn_ctx* ctx, uint32_t* inc) {
    if (ctx->npend == 0) {
        if (inc) {
            red_inc_kernel<<<1, 1, 0, ctx->stream>>>(inc);
            TN_LAUNCH_CHECK();
        }
        return TN_OK;
    }
    RedBatch b;
    b.nrec = ctx->npend;
    uint32_t gx = 1;
    for (int i = 0; i < ctx->npend; ++i) {
        b.r[i] = ctx->pend[i];
        const uint32_t nb = red_blocks(b.r[i]);
        if (nb > gx) gx = nb;
    }
    ctx->npend = 0;
    ctx->scratch_off = 0;          // the slabs are consumed in stream order
    slab_sum_multi_kernel<<<dim3(gx, b.nrec), 256, 0, ctx->stream>>>(b, inc);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_red_push(tn_ctx* ctx, const float* src, float* out, uint32_t n, uint32_t S, uint32_t stride,
                uint32_t flip) {
    if (n == 0 || out == nullptr) return TN_OK;
    if (ctx->npend == TN_RED_MAX) {
        int rc = tn_red_flush(ctx);
        if (rc) return rc;
    }
    tn_red_rec& r = ctx->pend[ctx->npend++];
    r.src = src; r.out = out; r.n = n; r.S = S; r.stride = stride; r.flip = flip;
    return TN_OK;
}

int tn_red_commit(tn_ctx* ctx) { return ctx->defer ? TN_OK : tn_red_flush(ctx); }

// bytes of context scratch: the whole buffer outside a deferral window, a fresh 256-byte aligned
// piece of it inside one (earlier pieces hold slabs that are still to be reduced)
int tn_scratch_get(tn_ctx* ctx, size_t bytes, float** out) {
    ++ctx->scratch_gen;
    bytes = (bytes + 255) & ~(size_t)255;
    size_t off = ctx->defer ? ctx->scratch_off : 0;
    if (off + bytes > ctx->scratch_bytes) {
        int rc = tn_red_flush(ctx);            // pending slabs live in the buffer about to go
        if (rc) return rc;
        off = 0;
        // inside a deferral window an overflow means the step's slabs do not fit: the forced flush above costs a
        // reduction launch and a round trip of those gradients through HBM EVERY step (wide6: 231 MB, 46 us), so the
        // buffer doubles until a whole step fits (a few synchronisations during the first steps, none afterwards)
        if (bytes > ctx->scratch_bytes || ctx->defer) {
            TN_HIP(hipStreamSynchronize(ctx->streams[0]));
            TN_HIP(hipStreamSynchronize(ctx->streams[1]));
            const size_t had = ctx->scratch_bytes;
            if (ctx->scratch) TN_HIP(hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
            // a deferral window needs room for every op of the step: grow generously
            size_t nb = ctx->defer ? 4 * bytes + (8u << 20) : bytes + (bytes >> 2);
            if (ctx->defer && nb < 2 * had) nb = 2 * had;
            hipError_t e = hipMalloc((void**)&ctx->scratch, nb);
            if (e != hipSuccess) return tn_fail(ctx, TN_E_NOMEM, "scratch hipMalloc(%zu) failed", nb);
            ctx->scratch_bytes = nb;
        }
    }
    *out = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->scratch) + off);
    if (ctx->defer) ctx->scratch_off = off + bytes;
    return TN_OK;
}

int tn_tmp_get(tn_ctx* ctx, size_t bytes, float** out) {
    const int k = ctx->stream == ctx->streams[1] ? 1 : 0;
    if (bytes > ctx->tmp_bytes[k]) {
        TN_HIP(hipStreamSynchronize(ctx->streams[k]));       // launches may still read the old buffer
        if (ctx->tmp[k]) TN_HIP(hipFree(ctx->tmp[k]));
        ctx->tmp[k] = nullptr; ctx->tmp_bytes[k] = 0;
        const size_t nb = bytes + (bytes >> 3);
        hipError_t e = hipMalloc(&ctx->tmp[k], nb);
        if (e != hipSuccess) return tn_fail(ctx, TN_E_NOMEM, "tmp hipMalloc(%zu) failed", nb);
        ctx->tmp_bytes[k] = nb;
    }
    *out = reinterpret_cast<float*>(ctx->tmp[k]);
    return TN_OK;
}

extern "C" int tn_defer_flush_step(tn_ctx* ctx, uint32_t* d_step) {
    ctx->defer = false;
    return tn_red_flush_inc(ctx, d_step);
}

// Forget every recorded-but-unfinished reduction of BOTH streams (their outputs belong to a net that no
// longer exists: finishing them would write through dangling pointers).
extern "C" int tn_defer_discard(tn_ctx* ctx) {
    ctx->npend = 0; ctx->scratch_off = 0; ctx->defer = false;
    for (int k = 0; k < 2; ++k) { ctx->npend_slot[k] = 0; ctx->scratch_off_slot[k] = 0; ctx->defer_slot[k] = false; }
    return TN_OK;
}

extern "C" int tn_defer_reductions(tn_ctx* ctx, int on) {
    if (on) {
        ctx->defer = true;
        return TN_OK;
    }
    ctx->defer = false;
    return tn_red_flush(ctx);
}
