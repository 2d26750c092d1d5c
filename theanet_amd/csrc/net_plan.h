// tn_net_plan_* / tn_net_step: a training (or test) step as ONE C call ("optionally a coarse
// tn_net_build(spec) / tn_net_step(i) for the launch-bound fast path").  Included by both backends (ctx.hip,
// theanet_cpu.cpp): plain host code.
//
// A plan is the flat list of C-ABI calls one step makes -- entry point + argument block -- recorded by the host
// (theanet_amd/plan.py watches the calls of a few ordinary steps, checks that they repeat and which arguments follow
// the minibatch index) and replayed here without the interpreter: the host cost of a mnist.prms step is ~65 us of
// Python + ctypes for ~14 calls against ~1.4 us per hipLaunchKernel (tools/probe/launchrate.hip), which at 512 images
// per GPU (one rank of the 8-GPU strong-scaling run) is more than the GPU needs for the step.
//
// Generic invocation without a foreign-function library: every entry point of include/theanet_hip.h takes the
// context followed by pointers / integers (INTEGER class of the x86-64 SysV ABI) and floats / doubles (SSE class);
// the two classes are assigned to registers independently (6 + 8) and only INTEGER arguments ever overflow to the stack
// (no entry point has more than 8 floating-point arguments: checked when a call is added).  So one prototype
//   int f(long x6, double x8, long x42 on the stack)
// reaches all of them: unused registers / stack slots are ignored by the callee, a float travels as the low half of
// its xmm register.
#pragma once
#if !defined(__x86_64__)
#error "net_plan.h: the generic call trampoline relies on the argument classes of the x86-64 SysV ABI (the host of an MI355X node); port tn_plan_invoke before building for another target"
#endif
#include <dlfcn.h>

#include <cstdint>
#include <cstring>
#include <vector>

#define TN_PLAN_MAX_INT 48        // 5 in registers after the context + 43 on the stack
struct tn_plan_call {
    void* fn;
    int nint, nflt;
    int64_t iv[TN_PLAN_MAX_INT], istride[TN_PLAN_MAX_INT];
    double fv[8];
};
struct tn_net_plan {
    std::vector<tn_plan_call> calls;
    void* self = nullptr;         // dlopen handle of this library
};

typedef int (*tn_plan_fn)(void*, long, long, long, long, long, double, double, double, double, double, double, double, double,
                          long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long);

static int tn_plan_invoke(tn_ctx* ctx, const tn_plan_call& c, int64_t i) {
    long a[TN_PLAN_MAX_INT];
    for (int k = 0; k < c.nint; ++k) a[k] = (long)(c.iv[k] + i * c.istride[k]);
    for (int k = c.nint; k < TN_PLAN_MAX_INT; ++k) a[k] = 0;
    const double* f = c.fv;
    return reinterpret_cast<tn_plan_fn>(c.fn)(
        ctx, a[0], a[1], a[2], a[3], a[4], f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7],
        a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18], a[19], a[20],
        a[21], a[22], a[23], a[24], a[25], a[26], a[27], a[28], a[29], a[30], a[31], a[32], a[33], a[34], a[35], a[36],
        a[37], a[38], a[39], a[40], a[41], a[42], a[43], a[44], a[45], a[46], a[47]);
}

extern "C" {

int tn_net_plan_create(tn_ctx* ctx, void** plan) {
    TN_REQUIRE(plan != nullptr, "tn_net_plan_create: NULL output");
    tn_net_plan* p = new tn_net_plan();
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&tn_version), &info) && info.dli_fname)
        p->self = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
    if (!p->self) {
        delete p;
        return tn_fail(ctx, TN_E_ARG, "tn_net_plan_create: cannot open the library's own handle");
    }
    *plan = p;
    return TN_OK;
}

// Append one call.  name: an entry point of this header whose first parameter is the context (which is not part of
// the argument list here).  kinds[k]: 0 pointer / integer of any width (vals[k] = the value, sign-extended to 64 bits;
// strides[k] is added per unit of tn_net_step's index), 1 float (vals[k] = its 32 bits), 2 double (its 64 bits).
int tn_net_plan_add(tn_ctx* ctx, void* plan, const char* name, int nargs, const uint8_t* kinds, const uint64_t* vals,
                    const int64_t* strides) {
    tn_net_plan* p = static_cast<tn_net_plan*>(plan);
    TN_REQUIRE(p && name && nargs >= 0 && (nargs == 0 || (kinds && vals)), "tn_net_plan_add: bad arguments");
    tn_plan_call c{};
    c.fn = dlsym(p->self, name);
    TN_REQUIRE(c.fn != nullptr && strncmp(name, "tn_", 3) == 0, "tn_net_plan_add: no entry point '%s'", name);
    for (int k = 0; k < nargs; ++k) {
        if (kinds[k] == 0) {
            TN_REQUIRE(c.nint < TN_PLAN_MAX_INT, "tn_net_plan_add: %s: too many integer arguments", name);
            c.iv[c.nint] = (int64_t)vals[k];
            c.istride[c.nint] = strides ? strides[k] : 0;
            ++c.nint;
        } else {
            TN_REQUIRE(c.nflt < 8 && kinds[k] <= 2, "tn_net_plan_add: %s: more than 8 floating-point arguments", name);
            uint64_t bits = kinds[k] == 1 ? (vals[k] & 0xffffffffull) : vals[k];
            memcpy(&c.fv[c.nflt++], &bits, 8);
        }
    }
    p->calls.push_back(c);
    return TN_OK;
}

// Issue every call of the plan in order, integer arguments advanced by index * stride; stops at the first error.
int tn_net_step(tn_ctx* ctx, void* plan, int64_t index) {
    tn_net_plan* p = static_cast<tn_net_plan*>(plan);
    TN_REQUIRE(p != nullptr, "tn_net_step: NULL plan");
    for (const tn_plan_call& c : p->calls) {
        const int rc = tn_plan_invoke(ctx, c, index);
        if (rc) return rc;
    }
    return TN_OK;
}

int tn_net_plan_size(tn_ctx* ctx, void* plan) {
    (void)ctx;
    return plan ? (int)static_cast<tn_net_plan*>(plan)->calls.size() : 0;
}

int tn_net_plan_destroy(tn_ctx* ctx, void* plan) {
    (void)ctx;
    tn_net_plan* p = static_cast<tn_net_plan*>(plan);
    if (p) {
        if (p->self) dlclose(p->self);
        delete p;
    }
    return TN_OK;
}

}  // extern "C"
