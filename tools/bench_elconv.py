#!/usr/bin/env python
"""Where the first launch of the mnist step goes (elastic_convpool_fwd_kernel, 4096 images of 28 x 28): the op through the
C-ABI with pieces switched off by its own arguments -- no flip noise (pflip 0: no Philox), no gather (map NULL: identity),
no pooling mask -- HIP events, us per launch.   python tools/bench_elconv.py [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from theanet_amd import _lib
from theanet_amd.device import get_context
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = get_context(); lib = ctx.lib
rng = np.random.default_rng(0)
h = w = 28; K = 4; Ho = Wo = 26; Hp = Wp = 13
x = ctx.array(rng.random((4 * N, 1, h, w), dtype=np.float32))
xd = ctx.empty((N, 1, h, w)); y = ctx.empty((N, K, Hp, Wp)); mask = ctx.empty((N, K, Hp, Wp), np.uint8)
W = ctx.array((rng.standard_normal((K, 1, 3, 3)) / 3).astype(np.float32)); b = ctx.zeros((K,))
idx = (np.arange(h * w) + rng.integers(-2, 3, h * w)).clip(0, h * w - 1).astype(np.int32)
mi = ctx.array(idx)


def timeit(fn, iters=200):
    for _ in range(5):
        fn()
    ctx.sync()
    a, e = ctypes.c_void_p(), ctypes.c_void_p()
    lib.tn_event_create(ctx.h, ctypes.byref(a)); lib.tn_event_create(ctx.h, ctypes.byref(e))
    lib.tn_event_record(ctx.h, a)
    for _ in range(iters):
        fn()
    lib.tn_event_record(ctx.h, e)
    ms = ctypes.c_float()
    ctx.call("tn_event_elapsed_ms", a, e, ctypes.byref(ms))
    return ms.value * 1e3 / iters


def run(name, map_ptr, pflip, mask_ptr):
    f = lambda: ctx.call("tn_elastic_convpool_fwd_mask", x.ptr, 0, None, xd.ptr, N, h, w, 1, 1, map_ptr, None, None, pflip, None,
                         12345, 7, None, 0, W.ptr, b.ptr, y.ptr, mask_ptr, K, 3, 0, Ho, Wo, 2, Hp, Wp, _lib.TN_ACT_LEAKY, .1)
    print("%-52s %6.1f us" % (name, timeit(f)))


run("as in the step (gather, flip noise, mask)", mi.ptr, .03, mask.ptr)
run("no flip noise (no Philox)", mi.ptr, 0.0, mask.ptr)
run("no gather (identity map), flip noise", None, .03, mask.ptr)
run("no gather, no flip noise", None, 0.0, mask.ptr)
run("no pooling mask", mi.ptr, .03, None)

# ---- the conv2 block forward of mnist.prms (convpool_fwd_kernel): with and without its pooling mask
C2, K2, H2, Ho2, Hp2 = 4, 20, 13, 11, 6
x2 = ctx.array(rng.standard_normal((N, C2, H2, H2)).astype(np.float32)); W2 = ctx.array((rng.standard_normal((K2, C2, 3, 3)) / 6).astype(np.float32))
b2 = ctx.zeros((K2,)); y2 = ctx.empty((N, K2, Hp2, Hp2)); m2 = ctx.empty((N, K2, Hp2, Hp2), np.uint8)
for name, mp in (("conv2 block forward, with the pooling mask", m2.ptr), ("conv2 block forward, no mask", None)):
    f = lambda: ctx.call("tn_convpool_fwd_mask", x2.ptr, W2.ptr, b2.ptr, y2.ptr, mp, N, C2, H2, H2, K2, 3, 0, Ho2, Ho2, 2, Hp2, Hp2,
                         _lib.TN_ACT_LEAKY, .05)
    print("%-52s %6.1f us" % (name, timeit(f)))
