// Data-parallel gradient exchange: one flat fp32 all-reduce per step over RCCL (xGMI).
// New relative to the reference (which is single-process); SURVEY.md section 8(e).
// RCCL is dlopen'ed on first use so single-GPU runs never load it.
#include <dlfcn.h>

#include "common.h"

// Minimal slice of the RCCL (NCCL-compatible) ABI, declared here so that the library has no
// link-time dependency on librccl.so.
typedef struct { char internal[128]; } tn_ncclUniqueId;
typedef void* tn_ncclComm_t;
enum { TN_NCCL_FLOAT32 = 7 };
enum { TN_NCCL_SUM = 0, TN_NCCL_MAX = 2 };

struct RcclApi {
    int (*GetUniqueId)(tn_ncclUniqueId*);
    int (*CommInitRank)(tn_ncclComm_t*, int, tn_ncclUniqueId, int);
    int (*CommDestroy)(tn_ncclComm_t);
    int (*AllReduce)(const void*, void*, size_t, int, int, tn_ncclComm_t, hipStream_t);
    int (*ReduceScatter)(const void*, void*, size_t, int, int, tn_ncclComm_t, hipStream_t);
    int (*AllGather)(const void*, void*, size_t, int, tn_ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(int);
};
static RcclApi g_rccl;

static int load_rccl(tn_ctx* ctx) {
    if (ctx->rccl_lib) return TN_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return tn_fail(ctx, TN_E_COMM, "cannot dlopen librccl: %s", dlerror());
#define SYM(field, name)                                                              \
    *(void**)(&g_rccl.field) = dlsym(lib, name);                                       \
    if (!g_rccl.field) return tn_fail(ctx, TN_E_COMM, "librccl lacks symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(ReduceScatter, "ncclReduceScatter");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    ctx->rccl_lib = lib;
    return TN_OK;
}

#define TN_NCCL(call)                                                                   \
    do {                                                                                \
        int r_ = (call);                                                                \
        if (r_ != 0)                                                                    \
            return tn_fail(ctx, TN_E_COMM, "%s -> %s", #call, g_rccl.GetErrorString(r_)); \
    } while (0)

extern "C" {

int tn_comm_unique_id(tn_ctx* ctx, void* id128) {
    int rc = load_rccl(ctx);
    if (rc) return rc;
    tn_ncclUniqueId id;
    TN_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return TN_OK;
}

int tn_comm_init(tn_ctx* ctx, const void* id128, int rank, int world) {
    TN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tn_comm_init: rank %d / world %d", rank, world);
    int rc = load_rccl(ctx);
    if (rc) return rc;
    TN_HIP(hipSetDevice(ctx->device));
    tn_ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    tn_ncclComm_t comm = nullptr;
    TN_NCCL(g_rccl.CommInitRank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    return TN_OK;
}

int tn_comm_destroy(tn_ctx* ctx) {
    if (ctx && ctx->comm) {
        hipStreamSynchronize(ctx->streams[0]);
        hipStreamSynchronize(ctx->streams[1]);
        if (ctx->comm_stream) hipStreamSynchronize(ctx->comm_stream);
        g_rccl.CommDestroy((tn_ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    return TN_OK;
}

static int allreduce(tn_ctx* ctx, float* buf, size_t n, int op) {
    if (!ctx->comm) return tn_fail(ctx, TN_E_COMM, "all-reduce without tn_comm_init");
    if (!n) return TN_OK;
    TN_NCCL(g_rccl.AllReduce(buf, buf, n, TN_NCCL_FLOAT32, op, (tn_ncclComm_t)ctx->comm, ctx->stream));
    return TN_OK;
}

int tn_allreduce_sum(tn_ctx* ctx, float* buf, size_t n) { return allreduce(ctx, buf, n, TN_NCCL_SUM); }

int tn_allreduce_sum_async(tn_ctx* ctx, float* buf, size_t n, void* done_event) {
    if (!ctx->comm) return tn_fail(ctx, TN_E_COMM, "all-reduce without tn_comm_init");
    // behind the producer (the compute stream as enqueued so far) ...
    TN_HIP(hipEventRecord(ctx->comm_ev, ctx->stream));
    TN_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ev, 0));
    // ... and, being on ONE stream, behind every earlier collective of this entry point
    if (n) TN_NCCL(g_rccl.AllReduce(buf, buf, n, TN_NCCL_FLOAT32, TN_NCCL_SUM, (tn_ncclComm_t)ctx->comm, ctx->comm_stream));
    if (done_event) TN_HIP(hipEventRecord((hipEvent_t)done_event, ctx->comm_stream));
    return TN_OK;
}
int tn_allreduce_max(tn_ctx* ctx, float* buf, size_t n) { return allreduce(ctx, buf, n, TN_NCCL_MAX); }

int tn_allreduce_sum_rsag(tn_ctx* ctx, float* buf, size_t n, int on_comm_stream, void* done_event) {
    if (!ctx->comm) return tn_fail(ctx, TN_E_COMM, "all-reduce without tn_comm_init");
    hipStream_t st = ctx->stream;
    if (on_comm_stream) {
        TN_HIP(hipEventRecord(ctx->comm_ev, ctx->stream));
        TN_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ev, 0));
        st = ctx->comm_stream;
    }
    const size_t W = (size_t)ctx->world, q = n / W, body = q * W;
    tn_ncclComm_t comm = (tn_ncclComm_t)ctx->comm;
    if (q) {
        // in place: rank r's sums land in its own slice, the gather fills in everybody else's
        float* mine = buf + (size_t)ctx->rank * q;
        TN_NCCL(g_rccl.ReduceScatter(buf, mine, q, TN_NCCL_FLOAT32, TN_NCCL_SUM, comm, st));
        TN_NCCL(g_rccl.AllGather(mine, buf, q, TN_NCCL_FLOAT32, comm, st));
    }
    if (n > body) TN_NCCL(g_rccl.AllReduce(buf + body, buf + body, n - body, TN_NCCL_FLOAT32, TN_NCCL_SUM, comm, st));
    if (on_comm_stream && done_event) TN_HIP(hipEventRecord((hipEvent_t)done_event, ctx->comm_stream));
    return TN_OK;
}

}  // extern "C"
